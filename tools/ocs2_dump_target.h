// ocs2_dump_target.h - the target-trajectory arithmetic of the external parity hook in PLAIN DOUBLES (no OCS2, no Eigen).
//
// tools/ocs2_dump_primal.cpp (compiled where the reference runs) builds the TargetTrajectories it hands to the reference's own SqpMpc
// with these functions; tools/compare_ocs2_dump.py rebuilds the same problem through libbpmpc (bpmpc_cmd_vel_to_targets).  A CPU test
// compiles this header HERE and requires it to equal the library bit for bit (tests/test_ocs2_dump.py::test_dump_target_header_*), so
// the tool and the comparer cannot drift apart: round 2's hand-written target in the dump program forgot the momentum reference
// (stateTrajectory[*].head(3) = cmdVelRot) and the z / pitch / roll of the first point.
//
// What is restated (reference: bipedal_controllers/src/TargetTrajectoriesPublisher.cpp):
//   target_pose_to_targets   :40-58   two points {t_now, t_reach}; first = current pose with z = comHeight, pitch = roll = 0;
//                                     second = target pose; momentum entries 0; joints = defaultJointState on both
//   cmd_vel_to_targets       :76-99   v = R_zyx(current yaw, pitch, roll) * cmd[0:3]; target x, y advanced by v * T, z = comHeight,
//                                     yaw advanced by cmd[3] * T, pitch = roll = 0; reach time t_now + T; entries [0:3] of BOTH points = v
//   goal_to_targets          :60-74   target = (goal x, goal y, comHeight, goal yaw, 0, 0); reach time = t_now +
//                                     max(|dyaw| / targetRotationVelocity, |dxy| / targetDisplacementVelocity) (:30-38)
// State layout: x = [h_lin/m (3), h_ang/m (3), base x y z (3), yaw pitch roll (3), joints (nj)], nx = 12 + nj.
// The rotation is R = Rz(yaw) Ry(pitch) Rx(roll) [OCS2-upstream getRotationMatrixFromZyxEulerAngles]; the products are written in the
// operation order of the library (bipedal_control_amd/csrc/reference_gen.cpp cmd_vel_to_targets) - upstream may associate
// c1 * (s2 * s3) where this writes (c1 * s2) * s3: one ulp, far below the comparison tolerance, but the drift test is bit for bit.
#pragma once
#include <cmath>

namespace bpmpc_dump {

struct TargetSettings {
  int nj;                        // actuated joints (CentroidalModelInfo::actuatedDofNum)
  double com_height;             // reference.info comHeight
  const double* default_joints;  // reference.info defaultJointState [nj]
  double target_rotation_velocity, target_displacement_velocity;   // reference.info targetRotationVelocity / targetDisplacementVelocity
};

// times[2], states[2 * (12 + nj)]
inline void target_pose_to_targets(const TargetSettings& s, const double target_pose[6], double t_now, const double* x_now, double t_reach,
                                   double* times, double* states) {
  const int nx = 12 + s.nj;
  for (int i = 0; i < 2 * nx; ++i) states[i] = 0.0;
  times[0] = t_now;
  times[1] = t_reach;
  for (int i = 0; i < 6; ++i) {
    states[6 + i] = x_now[6 + i];
    states[nx + 6 + i] = target_pose[i];
  }
  states[6 + 2] = s.com_height;   // z
  states[6 + 4] = 0.0;            // pitch
  states[6 + 5] = 0.0;            // roll
  for (int j = 0; j < s.nj; ++j) states[12 + j] = states[nx + 12 + j] = s.default_joints[j];
}

inline void cmd_vel_to_targets(const TargetSettings& s, const double cmd[4], double t_now, const double* x_now, double time_to_target,
                               double* times, double* states) {
  const int nx = 12 + s.nj;
  const double z = x_now[9], y = x_now[10], r = x_now[11];
  const double cz = std::cos(z), sz = std::sin(z), cy = std::cos(y), sy = std::sin(y), cx = std::cos(r), sx = std::sin(r);
  const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                       sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                       -sy,     cy * sx,                cy * cx};
  double v[3];
  for (int i = 0; i < 3; ++i) v[i] = R[3 * i] * cmd[0] + R[3 * i + 1] * cmd[1] + R[3 * i + 2] * cmd[2];
  const double pose[6] = {x_now[6] + v[0] * time_to_target, x_now[7] + v[1] * time_to_target, s.com_height, x_now[9] + cmd[3] * time_to_target, 0.0, 0.0};
  target_pose_to_targets(s, pose, t_now, x_now, t_now + time_to_target, times, states);
  for (int i = 0; i < 3; ++i) states[i] = states[nx + i] = v[i];
}

inline void goal_to_targets(const TargetSettings& s, const double goal[4], double t_now, const double* x_now, double* times, double* states) {
  const double pose[6] = {goal[0], goal[1], s.com_height, goal[3], 0.0, 0.0};
  const double dx = pose[0] - x_now[6], dy = pose[1] - x_now[7], dyaw = pose[3] - x_now[9];
  const double rotation_time = std::fabs(dyaw) / s.target_rotation_velocity;
  const double displacement_time = std::sqrt(dx * dx + dy * dy) / s.target_displacement_velocity;
  target_pose_to_targets(s, pose, t_now, x_now, t_now + (rotation_time > displacement_time ? rotation_time : displacement_time), times, states);
}

}  // namespace bpmpc_dump
