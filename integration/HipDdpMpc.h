// HipDdpMpc - the MPC_BASE that owns the engine's DDP solver: counterpart of ocs2::GaussNewtonDDP_MPC at
//   ocs2_bipedal_robot_ros/src/BipedalRobotDdpMpcNode.cpp:70-71
//
//   // was: GaussNewtonDDP_MPC mpc(interface.mpcSettings(), interface.ddpSettings(), interface.getRollout(),
//   //                             interface.getOptimalControlProblem(), interface.getInitializer());
//   ocs2::bipedal_robot::HipDdpMpc mpc(interface.mpcSettings(), taskFile, urdfFile, referenceFile, interface.getOptimalControlProblem());
//   mpc.getSolverPtr()->setReferenceManager(rosReferenceManagerPtr);        // unchanged (:72)
//   mpc.getSolverPtr()->addSynchronizedModule(gaitReceiverPtr);             // unchanged (:73)
//
// The solver behind it is the same C ABI handle as HipSqpSolver with bpmpc_settings.solver = BPMPC_SOLVER_DDP: ONE ILQR iteration per run on
// the ddp block of task.info:115-156 (algorithm ILQR, maxNumIterations 1, LINE_SEARCH, DIAGONAL_SHIFT - any other value is refused when the
// solver is created), the solution is the accepted roll-out on its own time points with a FeedforwardController (ddp.useFeedbackPolicy
// false).  What of GaussNewtonDDP is NOT behind it: SLQ and the continuous-time backward pass, later iterations on the roll-out's grid,
// the constraint penalty schedule, getValueFunction().
// Syntax-checked in this repository against integration/mock_ocs2 only (see HipSqpSolver.h).
#pragma once

#include <memory>
#include <string>

#include <ocs2_mpc/MPC_BASE.h>

#include "HipSqpSolver.h"

namespace ocs2 {
namespace bipedal_robot {

class HipDdpMpc final : public MPC_BASE {
 public:
  HipDdpMpc(mpc::Settings mpcSettings, const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile,
            const OptimalControlProblem& optimalControlProblem, HipSqpSolver::Settings solverSettings = HipSqpSolver::Settings())
      : MPC_BASE(std::move(mpcSettings)) {
    solverSettings.solver = BPMPC_SOLVER_DDP;
    solverPtr_.reset(new HipSqpSolver(taskFile, urdfFile, referenceFile, optimalControlProblem, solverSettings));
  }
  ~HipDdpMpc() override = default;

  HipSqpSolver* getSolverPtr() override { return solverPtr_.get(); }
  const HipSqpSolver* getSolverPtr() const override { return solverPtr_.get(); }

 protected:
  // GaussNewtonDDP_MPC::calculateController
  void calculateController(scalar_t initTime, const vector_t& initState, scalar_t finalTime) override {
    if (settings().coldStart_) solverPtr_->reset();
    solverPtr_->run(initTime, initState, finalTime);
  }

 private:
  std::unique_ptr<HipSqpSolver> solverPtr_;
};

}  // namespace bipedal_robot
}  // namespace ocs2
