// SYNTAX-CHECK STAND-IN (see ../README.md) for ocs2_mpc/MPC_BASE.h and mpc::Settings.  Declarations only; not OCS2.
#pragma once
#include <ocs2_oc/oc_solver/SolverBase.h>
namespace ocs2 {
namespace mpc { struct Settings { scalar_t timeHorizon_ = 1.0; bool coldStart_ = false; }; }
class MPC_BASE {
 public:
  explicit MPC_BASE(mpc::Settings s) : settings_(std::move(s)) {}
  virtual ~MPC_BASE() = default;
  virtual bool run(scalar_t currentTime, const vector_t& currentState) { calculateController(currentTime, currentState, currentTime + settings_.timeHorizon_); return true; }
  virtual SolverBase* getSolverPtr() = 0;
  virtual const SolverBase* getSolverPtr() const = 0;
  const mpc::Settings& settings() const { return settings_; }
 protected:
  virtual void calculateController(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
 private:
  mpc::Settings settings_;
};
}  // namespace ocs2
