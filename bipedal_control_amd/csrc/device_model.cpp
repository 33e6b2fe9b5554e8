#include "device_model.h"

#include <cstring>
#include <stdexcept>

namespace bpmpc {

DeviceModel make_device_model(const RobotModel& m) {
  DeviceModel d;
  std::memset(&d, 0, sizeof(d));
  if (m.nj < 1 || m.nj > kMaxJoints) throw std::runtime_error("make_device_model: unsupported joint count");
  d.nj = m.nj;
  for (int b = 0; b <= m.nj; ++b) {
    d.parent[b] = m.parent[b];
    // path base -> b
    int chain[kMaxJoints], n = 0;
    for (int c = b; c > 0; c = m.parent[c]) {
      if (n >= kMaxJoints) throw std::runtime_error("make_device_model: kinematic chain too deep");
      chain[n++] = c;
    }
    d.depth[b] = n;
    if (n > d.max_depth) d.max_depth = n;
    for (int k = 0; k < n; ++k) d.path[b][k] = chain[n - 1 - k];
    for (int i = 0; i < 9; ++i) d.Rfix[b][i] = m.Rfix[b][i];
    for (int i = 0; i < 3; ++i) { d.pfix[b][i] = m.pfix[b][i]; d.axis[b][i] = m.axis[b][i]; d.com[b][i] = m.com[b][i]; }
    d.mass[b] = m.mass[b];
    const double* I = m.inertia[b];
    d.inertia[b][0] = I[0]; d.inertia[b][1] = 0.5 * (I[1] + I[3]); d.inertia[b][2] = 0.5 * (I[2] + I[6]);
    d.inertia[b][3] = I[4]; d.inertia[b][4] = 0.5 * (I[5] + I[7]); d.inertia[b][5] = I[8];
  }
  for (int b = 0; b <= m.nj; ++b)
    for (int c = 0; c <= m.nj; ++c) {
      bool inside = (b == 0);
      for (int k = c; k > 0 && !inside; k = m.parent[k]) inside = (k == b);
      if (inside) d.subtree[b] |= (1u << c);
    }
  for (int i = 0; i < kNumContacts; ++i) {
    d.contact_body[i] = m.contact_body[i];
    for (int k = m.contact_body[i]; k > 0; k = m.parent[k]) d.contact_path[i] |= (1u << k);
    for (int a = 0; a < 3; ++a) d.contact_off[i][a] = m.contact_off[i][a];
  }
  for (int i = 0; i < m.nx * m.nx; ++i) d.Q[i] = m.Q[i];
  for (int i = 0; i < m.nu * m.nu; ++i) d.R[i] = m.R[i];
  d.friction = m.friction_coefficient;
  d.cone_reg = m.cone_regularization;
  d.cone_grip = m.cone_gripper_force;
  d.cone_shift = m.cone_hessian_shift;
  d.barrier_mu = m.barrier_mu;
  d.barrier_delta = m.barrier_delta;
  d.cone_gauss_newton = 0;
  if (m.hard_friction_cone) {     // see device_model.h; [OCS2-upstream, recalled] multiple_shooting::setupIntermediateNode, inequality constraints as penalty
    d.barrier_mu = m.sqp_inequality_mu;
    d.barrier_delta = m.sqp_inequality_delta;
    d.cone_shift = 0.0;
    d.cone_gauss_newton = 1;
  }
  d.serial_legs = (m.nj % 2 == 0) ? 1 : 0;
  for (int b = 1; b <= m.nj && d.serial_legs; ++b) {
    const bool head = (b == 1 || b == m.nj / 2 + 1);
    if (m.parent[b] != (head ? 0 : b - 1)) d.serial_legs = 0;
  }
  d.pos_gain = m.position_error_gain;
  d.robot_mass = m.robot_mass;
  return d;
}

}  // namespace bpmpc
