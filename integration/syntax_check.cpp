// Compile-only translation unit (tests/test_integration_headers.py): instantiates the adaptor against integration/mock_ocs2 so that
// a missing override of a SolverBase / MPC_BASE pure virtual, a typo or a signature mismatch with include/bpmpc.h fails the build.
// Syntax check only - it pins nothing about OCS2 and is never linked or run.
#include "HipDdpMpc.h"
#include "HipSqpMpc.h"

namespace {
struct FixedReferences final : ocs2::ReferenceManagerInterface {
  ocs2::ModeSchedule ms;
  ocs2::TargetTrajectories tt;
  const ocs2::ModeSchedule& getModeSchedule() const override { return ms; }
  const ocs2::TargetTrajectories& getTargetTrajectories() const override { return tt; }
};
}  // namespace

int syntax_check_only() {
  ocs2::OptimalControlProblem ocp;
  ocs2::bipedal_robot::HipSqpMpc mpc(ocs2::mpc::Settings(), "task.info", "robot.urdf", "reference.info", ocp);   // concrete: every pure virtual is overridden
  mpc.getSolverPtr()->setReferenceManager(std::make_shared<FixedReferences>());
  mpc.getSolverPtr()->addSynchronizedModule(std::make_shared<ocs2::SolverSynchronizedModule>());
  ocs2::vector_t x(24);
  mpc.run(0.0, x);
  ocs2::PrimalSolution primal;
  mpc.getSolverPtr()->getPrimalSolution(1.0, &primal);
  ocs2::bipedal_robot::HipDdpMpc ddp(ocs2::mpc::Settings(), "task.info", "robot.urdf", "reference.info", ocp);       // the DDP counterpart (BipedalRobotDdpMpcNode.cpp:70-71)
  ddp.getSolverPtr()->setReferenceManager(std::make_shared<FixedReferences>());
  ddp.run(0.0, x);
  return static_cast<int>(mpc.getSolverPtr()->getNumIterations()) + static_cast<int>(mpc.getSolverPtr()->getPerformanceIndeces().merit) +
         static_cast<int>(ddp.getSolverPtr()->getNumIterations());
}
