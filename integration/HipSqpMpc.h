// HipSqpMpc - the MPC_BASE that owns a HipSqpSolver: counterpart of ocs2::SqpMpc at
//   bipedal_controllers/src/BipedalController.cpp:303-306   and   ocs2_bipedal_robot_ros/src/BipedalRobotSqpMpcNode.cpp:70.
//
//   // was: mpc_ = std::make_shared<SqpMpc>(bipedalInterface_->mpcSettings(), bipedalInterface_->sqpSettings(),
//   //                                      bipedalInterface_->getOptimalControlProblem(), bipedalInterface_->getInitializer());
//   mpc_ = std::make_shared<ocs2::bipedal_robot::HipSqpMpc>(bipedalInterface_->mpcSettings(), taskFile, urdfFile, referenceFile,
//                                                           bipedalInterface_->getOptimalControlProblem());
//   mpc_->getSolverPtr()->setReferenceManager(rosReferenceManagerPtr);      // unchanged (:307)
//   mpc_->getSolverPtr()->addSynchronizedModule(gaitReceiverPtr);           // unchanged (:308)
//
// Syntax-checked in this repository against integration/mock_ocs2 only (see HipSqpSolver.h).
#pragma once

#include <memory>
#include <string>

#include <ocs2_mpc/MPC_BASE.h>

#include "HipSqpSolver.h"

namespace ocs2 {
namespace bipedal_robot {

class HipSqpMpc final : public MPC_BASE {
 public:
  HipSqpMpc(mpc::Settings mpcSettings, const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile,
            const OptimalControlProblem& optimalControlProblem, HipSqpSolver::Settings solverSettings = HipSqpSolver::Settings())
      : MPC_BASE(std::move(mpcSettings)) {
    solverPtr_.reset(new HipSqpSolver(taskFile, urdfFile, referenceFile, optimalControlProblem, solverSettings));
  }
  ~HipSqpMpc() override = default;

  HipSqpSolver* getSolverPtr() override { return solverPtr_.get(); }
  const HipSqpSolver* getSolverPtr() const override { return solverPtr_.get(); }

 protected:
  // SqpMpc::calculateController
  void calculateController(scalar_t initTime, const vector_t& initState, scalar_t finalTime) override {
    if (settings().coldStart_) solverPtr_->reset();
    solverPtr_->run(initTime, initState, finalTime);
  }

 private:
  std::unique_ptr<HipSqpSolver> solverPtr_;
};

}  // namespace bipedal_robot
}  // namespace ocs2
