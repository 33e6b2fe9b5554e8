// ocs2_dump_primal - EXTERNAL PARITY HOOK, to be compiled and run on a box that has the reference (zitongbai/bipedal_control) and
// its dependencies (OCS2, Pinocchio, CppAD, HPIPM) built: it runs ONE solve of the reference's own SqpMpc on a stated problem and
// writes the PrimalSolution as CSV.  tools/compare_ocs2_dump.py diffs that file against this repository's oracle (and the HIP path
// when a GPU is present).  It cannot be built in this repository's container (none of the dependencies exist here) - it is shipped
// so that the oracle's "parity unpinned" status can be lifted by anyone who has the reference running.
//
// Add to ocs2_bipedal_robot_ros/CMakeLists.txt next to bipedal_robot_sqp_mpc:
//     add_executable(ocs2_dump_primal <path>/ocs2_dump_primal.cpp)
//     target_link_libraries(ocs2_dump_primal ${catkin_LIBRARIES})
// Usage:
//     ocs2_dump_primal <task.info> <robot.urdf> <reference.info> <out.csv> [intervals = 20] [gait.info gaitName]
//   default = BASELINE.json configs[0] / SURVEY.md section 8(d) Config 1: t0 = 0, x0 = initialState (task.info), schedule all STANCE
//   (initialModeSchedule of reference.info), target = two identical points [0_6, 0, 0, comHeight, 0, 0, 0, defaultJointState] at
//   t = 0 and t = horizon, cold start, sqp.sqpIteration iterations (1), horizon = intervals * sqp.dt.
//   With a gait: that template is inserted at t = -1.225 s (scenarios.GAIT_START) before the solve, the problem of configs[1] with the
//   unperturbed initial state and the velocity command (0.3, 0, 0, 0) (TargetTrajectoriesPublisher.cpp:40-62, TIME_TO_TARGET = horizon).
// Interfaces used: BipedalRobotInterface (ocs2_bipedal_robot/include/ocs2_bipedal_robot/BipedalRobotInterface.h:56-127), SqpMpc as
// constructed at ocs2_bipedal_robot_ros/src/BipedalRobotSqpMpcNode.cpp:70-72, GaitSchedule::insertModeSequenceTemplate
// (src/gait/GaitSchedule.cpp:46-72), loadModeSequenceTemplate (src/gait/ModeSequenceTemplate.cpp:50-71).
//
// CSV format ("bpmpc-ocs2-dump v1"):
//     # bpmpc-ocs2-dump v1,nx,nu,nodes,intervals,gait
//     k,t_k,x_k[0..nx),u_k[0..nu)        one row per node k = 0..nodes-1 (the terminal node repeats the last input, as PrimalSolution does)
#include <fstream>
#include <iomanip>
#include <iostream>

#include <ocs2_bipedal_robot/BipedalRobotInterface.h>
#include <ocs2_bipedal_robot/gait/ModeSequenceTemplate.h>
#include <ocs2_core/misc/LoadData.h>
#include <ocs2_sqp/SqpMpc.h>

using namespace ocs2;
using namespace bipedal_robot;

int main(int argc, char** argv) {
  if (argc < 5) {
    std::cerr << "usage: ocs2_dump_primal <task.info> <robot.urdf> <reference.info> <out.csv> [intervals] [gait.info gaitName]\n";
    return 2;
  }
  const std::string taskFile = argv[1], urdfFile = argv[2], referenceFile = argv[3], outFile = argv[4];
  const int intervals = argc > 5 ? std::atoi(argv[5]) : 20;
  const bool withGait = argc > 7;

  BipedalRobotInterface interface(taskFile, urdfFile, referenceFile);
  const auto& info = interface.getCentroidalModelInfo();
  sqp::Settings sqpSettings = interface.sqpSettings();
  mpc::Settings mpcSettings = interface.mpcSettings();
  const scalar_t horizon = intervals * sqpSettings.dt;
  mpcSettings.timeHorizon_ = horizon;
  mpcSettings.coldStart_ = true;

  scalar_t comHeight = 0.0;
  vector_t defaultJointState(info.actuatedDofNum);
  loadData::loadCppDataType(referenceFile, "comHeight", comHeight);
  loadData::loadEigenMatrix(referenceFile, "defaultJointState", defaultJointState);

  const vector_t x0 = interface.getInitialState();
  vector_t xTarget0 = vector_t::Zero(info.stateDim), xTarget1;
  TargetTrajectories targets;
  if (!withGait) {
    xTarget0(8) = comHeight;
    xTarget0.tail(info.actuatedDofNum) = defaultJointState;
    xTarget1 = xTarget0;
  } else {
    // cmdVelToTargetTrajectories((0.3, 0, 0, 0)) from the current state, reach time = horizon (TargetTrajectoriesPublisher.cpp:40-62)
    const vector_t pose0 = x0.segment<6>(6);
    const scalar_t yaw = pose0(3);
    vector_t pose1 = pose0;
    pose1(0) += 0.3 * std::cos(yaw) * horizon;
    pose1(1) += 0.3 * std::sin(yaw) * horizon;
    pose1(2) = comHeight;
    pose1(4) = 0.0;
    pose1(5) = 0.0;
    xTarget0.segment<6>(6) = pose0;
    xTarget0.tail(info.actuatedDofNum) = defaultJointState;
    xTarget1 = xTarget0;
    xTarget1.segment<6>(6) = pose1;
    interface.getSwitchedModelReferenceManagerPtr()->getGaitSchedule()->insertModeSequenceTemplate(
        loadModeSequenceTemplate(argv[6], argv[7], false), -1.225, 2.0 * horizon);
  }
  targets = TargetTrajectories({0.0, horizon}, {xTarget0, xTarget1}, {vector_t::Zero(info.inputDim), vector_t::Zero(info.inputDim)});

  SqpMpc mpc(mpcSettings, sqpSettings, interface.getOptimalControlProblem(), interface.getInitializer());
  mpc.getSolverPtr()->setReferenceManager(interface.getReferenceManagerPtr());
  interface.getReferenceManagerPtr()->setTargetTrajectories(targets);
  mpc.run(0.0, x0);
  const PrimalSolution primal = mpc.getSolverPtr()->primalSolution(horizon);

  std::ofstream out(outFile);
  out << std::setprecision(17);
  out << "# bpmpc-ocs2-dump v1," << info.stateDim << "," << info.inputDim << "," << primal.timeTrajectory_.size() << "," << intervals << ","
      << (withGait ? argv[7] : "stance") << "\n";
  for (size_t k = 0; k < primal.timeTrajectory_.size(); ++k) {
    out << k << "," << primal.timeTrajectory_[k];
    for (int i = 0; i < primal.stateTrajectory_[k].size(); ++i) out << "," << primal.stateTrajectory_[k](i);
    for (int i = 0; i < primal.inputTrajectory_[k].size(); ++i) out << "," << primal.inputTrajectory_[k](i);
    out << "\n";
  }
  std::cerr << "wrote " << primal.timeTrajectory_.size() << " nodes to " << outFile << std::endl;
  return 0;
}
