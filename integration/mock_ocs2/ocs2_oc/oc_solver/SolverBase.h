// SYNTAX-CHECK STAND-IN (see ../../README.md) for ocs2_oc/oc_solver/SolverBase.h: the virtual interface as recalled in SURVEY.md
// section 8(b).  Declarations only; not OCS2.
#pragma once
#include <ocs2_core/control/LinearController.h>
namespace ocs2 {
class ReferenceManagerInterface {
 public:
  virtual ~ReferenceManagerInterface() = default;
  virtual const ModeSchedule& getModeSchedule() const = 0;
  virtual const TargetTrajectories& getTargetTrajectories() const = 0;
};
class SolverSynchronizedModule {};
class SolverObserver {};
class SolverBase {
 public:
  virtual ~SolverBase() = default;
  virtual void reset() = 0;
  void run(scalar_t initTime, const vector_t& initState, scalar_t finalTime) { runImpl(initTime, initState, finalTime); }
  void setReferenceManager(std::shared_ptr<ReferenceManagerInterface> p) { ref_ = std::move(p); }
  const ReferenceManagerInterface& getReferenceManager() const { return *ref_; }
  void addSynchronizedModule(std::shared_ptr<SolverSynchronizedModule>) {}
  void addSolverObserver(std::unique_ptr<SolverObserver>) {}
  virtual const PerformanceIndex& getPerformanceIndeces() const = 0;
  virtual size_t getNumIterations() const = 0;
  virtual const std::vector<PerformanceIndex>& getIterationsLog() const = 0;
  virtual scalar_t getFinalTime() const = 0;
  virtual void getPrimalSolution(scalar_t finalTime, PrimalSolution* primalSolutionPtr) const = 0;
  virtual const ProblemMetrics& getSolutionMetrics() const = 0;
  virtual const OptimalControlProblem& getOptimalControlProblem() const = 0;
  virtual ScalarFunctionQuadraticApproximation getValueFunction(scalar_t time, const vector_t& state) const = 0;
  virtual ScalarFunctionQuadraticApproximation getHamiltonian(scalar_t time, const vector_t& state, const vector_t& input) = 0;
  virtual vector_t getStateInputEqualityConstraintLagrangian(scalar_t time, const vector_t& state) const = 0;
  virtual MultiplierCollection getIntermediateDualSolution(scalar_t time) const = 0;
 private:
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* externalControllerPtr) = 0;
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const PrimalSolution& primalSolution) = 0;
  std::shared_ptr<ReferenceManagerInterface> ref_;
};
}  // namespace ocs2
