// ocs2_dump_primal - EXTERNAL PARITY HOOK, to be compiled and run on a box that has the reference (zitongbai/bipedal_control) and
// its dependencies (OCS2, Pinocchio, CppAD, HPIPM) built: it runs one or two solves of the reference's own SqpMpc on a stated problem
// and writes each PrimalSolution as CSV.  tools/compare_ocs2_dump.py diffs those files against this repository's oracle (and the HIP
// path when a GPU is present).  It cannot be built in this repository's container (none of the dependencies exist here) - it is shipped
// so that the oracle's "parity unpinned" status can be lifted by anyone who has the reference running.
//
// Add to ocs2_bipedal_robot_ros/CMakeLists.txt next to bipedal_robot_sqp_mpc (ocs2_dump_target.h beside the .cpp):
//     add_executable(ocs2_dump_primal <path>/ocs2_dump_primal.cpp)
//     target_link_libraries(ocs2_dump_primal ${catkin_LIBRARIES})
// Usage:
//     ocs2_dump_primal <task.info> <robot.urdf> <reference.info> <out.csv> [intervals = 20] [gait.info gaitName [secondSolvePeriod]]
//   default = BASELINE.json configs[0] / SURVEY.md section 8(d) Config 1: t0 = 0, x0 = initialState (task.info), schedule all STANCE
//   (initialModeSchedule of reference.info), target = two identical points [0_6, 0, 0, comHeight, 0, 0, 0, defaultJointState] at
//   t = 0 and t = horizon, cold start, sqp.sqpIteration iterations (1), horizon = intervals * sqp.dt.
//   With a gait: that template is inserted at t = -1.225 s (scenarios.GAIT_START) before the solve, the problem of configs[1] with the
//   unperturbed initial state and the velocity command (0.3, 0, 0, 0): cmdVelToTargetTrajectories restated in plain doubles in
//   ocs2_dump_target.h (TargetTrajectoriesPublisher.cpp:76-99, TIME_TO_TARGET = horizon) - the momentum reference head(3) = cmdVelRot on
//   both points, z = comHeight and pitch = roll = 0 on the first point included.
//   With secondSolvePeriod (e.g. 0.02): mpc.coldStart false and a SECOND run at t = period from the same measured state with the target
//   re-issued at that time - the warm start of SqpSolver::initializeStateInputTrajectories from the first solution (LinearController,
//   useFeedbackPolicy) and the second call of GaitSchedule::getModeSchedule on the same object; written to <out.csv>.2.
//   The cases a maintainer is asked for (README.md / INTEGRATION.md "pinning the oracle"): H1 stance 20, H1 trot 100, H1 trot 67 with a
//   second solve, Hunter trot 67 (model_settings.positionErrorGain 20: the position term of the zero-velocity / normal-velocity rows).
// Interfaces used: BipedalRobotInterface (ocs2_bipedal_robot/include/ocs2_bipedal_robot/BipedalRobotInterface.h:56-127), SqpMpc as
// constructed at ocs2_bipedal_robot_ros/src/BipedalRobotSqpMpcNode.cpp:70-72, GaitSchedule::insertModeSequenceTemplate
// (src/gait/GaitSchedule.cpp:46-72), loadModeSequenceTemplate (src/gait/ModeSequenceTemplate.cpp:50-71).
//
// CSV format ("bpmpc-ocs2-dump v2"; v1 files - no t0 / solve fields - are still read):
//     # bpmpc-ocs2-dump v2,nx,nu,nodes,intervals,gait,t0,solve
//     k,t_k,x_k[0..nx),u_k[0..nu)        one row per node k = 0..nodes-1 (the terminal node and pre-event nodes repeat the previous input,
//                                        as multiple_shooting::toPrimalSolution does)
#include <fstream>
#include <iomanip>
#include <iostream>

#include <ocs2_bipedal_robot/BipedalRobotInterface.h>
#include <ocs2_bipedal_robot/gait/ModeSequenceTemplate.h>
#include <ocs2_core/misc/LoadData.h>
#include <ocs2_sqp/SqpMpc.h>

#include "ocs2_dump_target.h"

using namespace ocs2;
using namespace bipedal_robot;

namespace {
void writeDump(const std::string& path, const PrimalSolution& primal, int nx, int nu, int intervals, const std::string& gait, scalar_t t0, int solve) {
  std::ofstream out(path);
  out << std::setprecision(17);
  out << "# bpmpc-ocs2-dump v2," << nx << "," << nu << "," << primal.timeTrajectory_.size() << "," << intervals << "," << gait << "," << t0 << "," << solve << "\n";
  for (size_t k = 0; k < primal.timeTrajectory_.size(); ++k) {
    out << k << "," << primal.timeTrajectory_[k];
    for (int i = 0; i < primal.stateTrajectory_[k].size(); ++i) out << "," << primal.stateTrajectory_[k](i);
    for (int i = 0; i < primal.inputTrajectory_[k].size(); ++i) out << "," << primal.inputTrajectory_[k](i);
    out << "\n";
  }
  std::cerr << "wrote " << primal.timeTrajectory_.size() << " nodes to " << path << std::endl;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) {
    std::cerr << "usage: ocs2_dump_primal <task.info> <robot.urdf> <reference.info> <out.csv> [intervals] [gait.info gaitName [secondSolvePeriod]]\n";
    return 2;
  }
  const std::string taskFile = argv[1], urdfFile = argv[2], referenceFile = argv[3], outFile = argv[4];
  const int intervals = argc > 5 ? std::atoi(argv[5]) : 20;
  const bool withGait = argc > 7;
  const scalar_t secondPeriod = argc > 8 ? std::atof(argv[8]) : 0.0;

  BipedalRobotInterface interface(taskFile, urdfFile, referenceFile);
  const auto& info = interface.getCentroidalModelInfo();
  sqp::Settings sqpSettings = interface.sqpSettings();
  mpc::Settings mpcSettings = interface.mpcSettings();
  const scalar_t horizon = intervals * sqpSettings.dt;
  mpcSettings.timeHorizon_ = horizon;
  mpcSettings.coldStart_ = !(secondPeriod > 0.0);

  scalar_t comHeight = 0.0, targetRotationVelocity = 0.0, targetDisplacementVelocity = 0.0;
  vector_t defaultJointState(info.actuatedDofNum);
  loadData::loadCppDataType(referenceFile, "comHeight", comHeight);
  loadData::loadCppDataType(referenceFile, "targetRotationVelocity", targetRotationVelocity);
  loadData::loadCppDataType(referenceFile, "targetDisplacementVelocity", targetDisplacementVelocity);
  loadData::loadEigenMatrix(referenceFile, "defaultJointState", defaultJointState);
  const bpmpc_dump::TargetSettings ts{static_cast<int>(info.actuatedDofNum), comHeight, defaultJointState.data(), targetRotationVelocity,
                                      targetDisplacementVelocity};

  const vector_t x0 = interface.getInitialState();
  const int nx = static_cast<int>(info.stateDim), nu = static_cast<int>(info.inputDim);
  // the target handed to the reference manager before a run at time t from the measured state x0
  auto targetsAt = [&](scalar_t t) {
    double times[2];
    std::vector<double> xs(2 * nx);
    if (!withGait) {     // Config 1: two identical points [0_6, 0, 0, comHeight, 0, 0, 0, defaultJointState] at t and t + horizon
      const double pose[6] = {0.0, 0.0, comHeight, 0.0, 0.0, 0.0};
      vector_t origin = vector_t::Zero(nx);
      bpmpc_dump::target_pose_to_targets(ts, pose, t, origin.data(), t + horizon, times, xs.data());
    } else {
      const double cmd[4] = {0.3, 0.0, 0.0, 0.0};
      bpmpc_dump::cmd_vel_to_targets(ts, cmd, t, x0.data(), horizon, times, xs.data());
    }
    const vector_t p0 = Eigen::Map<const vector_t>(xs.data(), nx), p1 = Eigen::Map<const vector_t>(xs.data() + nx, nx);
    return TargetTrajectories({times[0], times[1]}, {p0, p1}, {vector_t::Zero(nu), vector_t::Zero(nu)});
  };
  if (withGait) {
    interface.getSwitchedModelReferenceManagerPtr()->getGaitSchedule()->insertModeSequenceTemplate(
        loadModeSequenceTemplate(argv[6], argv[7], false), -1.225, 2.0 * horizon);
  }
  const std::string gaitName = withGait ? argv[7] : "stance";

  SqpMpc mpc(mpcSettings, sqpSettings, interface.getOptimalControlProblem(), interface.getInitializer());
  mpc.getSolverPtr()->setReferenceManager(interface.getReferenceManagerPtr());
  interface.getReferenceManagerPtr()->setTargetTrajectories(targetsAt(0.0));
  mpc.run(0.0, x0);
  writeDump(outFile, mpc.getSolverPtr()->primalSolution(horizon), nx, nu, intervals, gaitName, 0.0, 0);
  if (secondPeriod > 0.0) {
    interface.getReferenceManagerPtr()->setTargetTrajectories(targetsAt(secondPeriod));
    mpc.run(secondPeriod, x0);
    writeDump(outFile + ".2", mpc.getSolverPtr()->primalSolution(secondPeriod + horizon), nx, nu, intervals, gaitName, secondPeriod, 1);
  }
  return 0;
}
