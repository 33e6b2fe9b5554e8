// Minimal URDF reader: just enough XML to recover links (inertial) and joints (origin/parent/child/axis/type).
// The kinematic conventions applied on top of it are those of the reference's model construction,
// centroidal_model::createPinocchioInterface(urdf, jointNames) (call site
// ocs2_bipedal_robot/src/BipedalRobotInterface.cpp:117) -- see robot_model.cpp.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace bpmpc {

struct UrdfLink {
  std::string name;
  double mass = 0.0;
  double com[3] = {0, 0, 0};
  double inertia[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // about the com, expressed in the link frame
};

struct UrdfJoint {
  std::string name, type, parent, child;
  double xyz[3] = {0, 0, 0};
  double rpy[3] = {0, 0, 0};
  double axis[3] = {1, 0, 0};
};

struct UrdfRobot {
  std::vector<UrdfLink> links;
  std::vector<UrdfJoint> joints;
};

// Throws std::runtime_error on malformed input.
UrdfRobot read_urdf_file(const std::string& path);

// rotation from fixed-axis roll/pitch/yaw (URDF convention): R = Rz(yaw) Ry(pitch) Rx(roll), row major
void rpy_to_matrix(const double rpy[3], double R[9]);

}  // namespace bpmpc
