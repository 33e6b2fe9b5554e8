// STAND-IN (see ../../README.md) for ocs2_core: Types.h, ControllerBase / LinearController / FeedforwardController,
// ModeSchedule, TargetTrajectories, PerformanceIndex.  Written from SURVEY.md section 8(b); not OCS2.  The controllers keep what they are
// constructed from, so that integration/mock_run.cpp can print what the adaptor handed over.
#pragma once
#include <Eigen/Core>
#include <memory>
#include <string>
#include <vector>
namespace ocs2 {
using scalar_t = double;
using scalar_array_t = std::vector<scalar_t>;
using size_array_t = std::vector<size_t>;
using vector_t = Eigen::Matrix<scalar_t, Eigen::Dynamic, 1>;
using matrix_t = Eigen::Matrix<scalar_t, Eigen::Dynamic, Eigen::Dynamic>;
using vector_array_t = std::vector<vector_t>;
using matrix_array_t = std::vector<matrix_t>;
class ControllerBase { public: virtual ~ControllerBase() = default; virtual ControllerBase* clone() const = 0; };
class LinearController final : public ControllerBase {
 public:
  LinearController(scalar_array_t t, vector_array_t bias, matrix_array_t gain) : timeStamp_(std::move(t)), biasArray_(std::move(bias)), gainArray_(std::move(gain)) {}
  LinearController* clone() const override { return new LinearController(*this); }
  scalar_array_t timeStamp_;
  vector_array_t biasArray_;
  matrix_array_t gainArray_;
};
class FeedforwardController final : public ControllerBase {
 public:
  FeedforwardController(scalar_array_t t, vector_array_t u) : timeStamp_(std::move(t)), uffArray_(std::move(u)) {}
  FeedforwardController* clone() const override { return new FeedforwardController(*this); }
  scalar_array_t timeStamp_;
  vector_array_t uffArray_;
};
struct ModeSchedule { scalar_array_t eventTimes; size_array_t modeSequence; };
struct TargetTrajectories { scalar_array_t timeTrajectory; vector_array_t stateTrajectory; vector_array_t inputTrajectory; };
struct PerformanceIndex { scalar_t merit = 0, cost = 0, dualFeasibilitiesSSE = 0, dynamicsViolationSSE = 0, equalityConstraintsSSE = 0, inequalityConstraintsSSE = 0, equalityLagrangian = 0, inequalityLagrangian = 0; };
struct ScalarFunctionQuadraticApproximation {};
struct MultiplierCollection {};
// [OCS2-upstream, recalled] ocs2_oc/oc_data/ProblemMetrics.h: one Metrics per node; the equality-constraint members hold one vector per TERM of
// the constraint collection, in registration order (empty for a term that is not active at that time).  Only what the adaptor fills.
struct Metrics { scalar_t cost = 0; vector_t dynamicsViolation; vector_array_t stateEqConstraint, stateInputEqConstraint; };
struct ProblemMetrics {
  Metrics final;
  std::vector<Metrics> preJumps, intermediates;
  void clear() { final = Metrics(); preJumps.clear(); intermediates.clear(); }
};
struct OptimalControlProblem {};
struct PrimalSolution {
  scalar_array_t timeTrajectory_;
  vector_array_t stateTrajectory_, inputTrajectory_;
  scalar_array_t postEventIndices_;
  ModeSchedule modeSchedule_;
  std::unique_ptr<ControllerBase> controllerPtr_;
  PrimalSolution() = default;
  PrimalSolution(const PrimalSolution& o) : timeTrajectory_(o.timeTrajectory_), stateTrajectory_(o.stateTrajectory_), inputTrajectory_(o.inputTrajectory_), modeSchedule_(o.modeSchedule_), controllerPtr_(o.controllerPtr_ ? o.controllerPtr_->clone() : nullptr) {}
  PrimalSolution& operator=(const PrimalSolution& o) { timeTrajectory_ = o.timeTrajectory_; stateTrajectory_ = o.stateTrajectory_; inputTrajectory_ = o.inputTrajectory_; modeSchedule_ = o.modeSchedule_; controllerPtr_.reset(o.controllerPtr_ ? o.controllerPtr_->clone() : nullptr); return *this; }
  PrimalSolution(PrimalSolution&&) = default;
  PrimalSolution& operator=(PrimalSolution&&) = default;
};
}  // namespace ocs2
