// Plain-old-data model constants shared by host code and the HIP kernels (uploaded once per solver).
// Everything here is wave-uniform when read inside a kernel, so the compiler turns the accesses into scalar loads.
#pragma once
#include "robot_model.h"

namespace bpmpc {

struct DeviceModel {
  int nj;
  int parent[kMaxBodies];              // parent body of body b (b >= 1)
  int depth[kMaxBodies];               // number of joints between the base and body b
  int max_depth;                       // longest chain
  int path[kMaxBodies][kMaxJoints];    // path[b][d] = d-th body on the way base -> b (path[b][depth[b]-1] == b)
  unsigned subtree[kMaxBodies];        // bit c set: body c belongs to the subtree rooted at body b
  unsigned contact_path[kNumContacts]; // bit j set: joint j (1..nj) moves contact point i
  int contact_body[kNumContacts];
  double Rfix[kMaxBodies][9], pfix[kMaxBodies][3], axis[kMaxBodies][3];
  double mass[kMaxBodies], com[kMaxBodies][3];
  double inertia[kMaxBodies][6];       // xx, xy, xz, yy, yz, zz about the com, body frame
  double contact_off[kNumContacts][3];
  double Q[kMaxState * kMaxState];     // nx x nx, row major with stride nx
  double R[kMaxState * kMaxState];     // nu x nu, row major with stride nu
  double friction, cone_reg, cone_grip, cone_shift, barrier_mu, barrier_delta, pos_gain, robot_mass;
  // hard friction cone (inequality constraint, penalised by the SQP solver through its LINEAR approximation): barrier_mu / barrier_delta hold
  // sqp.inequalityConstraintMu / Delta, cone_shift is 0 and the second derivative of the cone does not enter the Hessian (Gauss-Newton)
  int cone_gauss_newton;
  // 1: the joints form two serial legs of nj / 2 joints each in body order (body b hangs on b - 1, the leg heads 1 and nj / 2 + 1 on the base):
  // the lane-per-coordinate kernels then walk the tree by DPP shifts between neighbouring lanes (linearize_fast.h, CHAIN)
  int serial_legs;
};

DeviceModel make_device_model(const RobotModel& m);

}  // namespace bpmpc
