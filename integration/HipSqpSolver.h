// HipSqpSolver - OCS2 SolverBase adaptor over the C ABI of libbpmpc.so (include/bpmpc.h).
//
// Drop-in for ocs2::SqpSolver at the reference's two construction sites of the MPC
//   bipedal_controllers/src/BipedalController.cpp:303-308          (SqpMpc + setReferenceManager + addSynchronizedModule)
//   ocs2_bipedal_robot_ros/src/BipedalRobotSqpMpcNode.cpp:70-72,85  (SqpMpc + MPC_ROS_Interface)
// through HipSqpMpc.h.  Lives in the reference tree (e.g. ocs2_bipedal_robot/include/ocs2_bipedal_robot/solver/) and is compiled
// there, against the real OCS2 / Eigen headers.  In THIS repository it is syntax-checked against integration/mock_ocs2 (stand-ins for
// the handful of OCS2 / Eigen declarations it touches; tests/test_integration_headers.py) and executed against the library with them
// (integration/mock_run.cpp, tests/test_gpu_adaptor_mock_run.py) - neither pins anything about OCS2's behaviour.  The list of SolverBase virtuals is the one recalled in SURVEY.md section 8(b): verify it against the OCS2
// checkout in use (a missing override is a compile error there, never silent).
//
// What SolverBase::run does before it reaches runImpl - preRun(): ReferenceManager::preSolverRun (SwitchedModelReferenceManager::
// modifyReferences, src/reference_manager/SwitchedModelReferenceManager.cpp:62-69: gait tiling + swing planner) and the synchronized
// modules (GaitReceiver) - is inherited unchanged; runImpl hands the resulting mode schedule and target trajectories to the engine.
#pragma once

#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <bpmpc.h>

#include <ocs2_core/control/LinearController.h>
#include <ocs2_oc/oc_solver/SolverBase.h>

namespace ocs2 {
namespace bipedal_robot {

struct HipSqpSolverSettings {
  int device = 0;
  int maxNodes = 160;        // shooting intervals incl. event nodes: timeHorizon / sqp.dt + 2 per gait event inside the horizon
  int sqpIterations = 0;     // <= 0: sqp.sqpIteration of task.info
  int solver = BPMPC_SOLVER_SQP;   // BPMPC_SOLVER_DDP: the engine's GaussNewtonDDP slice (HipDdpMpc.h; one ILQR iteration per run, ddp block of task.info)
  int useFeedbackPolicy = -1;      // -1: sqp.useFeedbackPolicy (ddp.useFeedbackPolicy for the DDP solver) of task.info, as the single settings object the
                                   // reference hands to its solver (BipedalController.cpp:303-306); 1 / 0 override it - for the engine's warm start and
                                   // policy rollout AND for the controller getPrimalSolution() returns alike (bpmpc_settings.feedback_policy)
  bool computeSolutionMetrics = false;   // fill getSolutionMetrics() after every run (one more kernel and a read-back: for solver observers)
  bool useHardFrictionConeConstraint = false;   // the interface's fourth constructor argument (BipedalRobotInterface.h:66-69): cones as inequality constraints
};

class HipSqpSolver final : public SolverBase {
 public:
  using Settings = HipSqpSolverSettings;

  /** The engine ingests the same three files BipedalRobotInterface is constructed from (BipedalRobotInterface.cpp:67-110). The
   *  OptimalControlProblem is only kept to answer getOptimalControlProblem(); its cost / constraint objects are not evaluated. */
  HipSqpSolver(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile, const OptimalControlProblem& ocp,
               Settings settings = Settings())
      : settings_(settings), ocp_(ocp) {
    if (settings_.solver == BPMPC_SOLVER_DDP && settings_.computeSolutionMetrics)
      throw std::runtime_error("[HipSqpSolver] solution metrics are evaluated on the shooting grid: not available for the DDP solver");
    check(bpmpc_model_create_ex(urdfFile.c_str(), taskFile.c_str(), referenceFile.c_str(), settings_.useHardFrictionConeConstraint ? 1 : 0, &model_));
    bpmpc_settings s{};
    s.device = settings_.device;
    s.max_batch = 1;
    s.max_nodes = settings_.maxNodes;
    s.sqp_iterations = settings_.sqpIterations;
    s.return_gains = 1;
    s.solver = settings_.solver;
    s.feedback_policy = settings_.useFeedbackPolicy < 0 ? 0 : (settings_.useFeedbackPolicy ? 1 : 2);
    const int rc = bpmpc_solver_create(model_, &s, &solver_);
    if (rc != BPMPC_OK) {
      const std::string why = bpmpc_last_error();
      bpmpc_model_destroy(model_);
      model_ = nullptr;
      throw std::runtime_error("[HipSqpSolver] " + why);
    }
    check(bpmpc_model_dims(model_, &nx_, &nu_, nullptr, nullptr));
    if (settings_.useFeedbackPolicy < 0) {       // the file's sqp block decides (integratorType / projectStateInputEqualityConstraints the engine
      double sqp[8] = {0}, ddp[20] = {0};        // does not implement were refused by bpmpc_model_create above: BPMPC_ERR_UNSUPPORTED)
      check(bpmpc_model_get(model_, "sqp", sqp, 8) >= 6 ? BPMPC_OK : BPMPC_ERR_IO);
      check(bpmpc_model_get(model_, "ddp", ddp, 20) >= 13 ? BPMPC_OK : BPMPC_ERR_IO);
      settings_.useFeedbackPolicy = (settings_.solver == BPMPC_SOLVER_DDP ? ddp[12] : sqp[5]) != 0.0 ? 1 : 0;
    }
  }
  ~HipSqpSolver() override {
    bpmpc_solver_destroy(solver_);
    bpmpc_model_destroy(model_);
  }
  HipSqpSolver(const HipSqpSolver&) = delete;
  HipSqpSolver& operator=(const HipSqpSolver&) = delete;

  // ---- SolverBase interface (as SqpSolver implements it)
  void reset() override {
    primalSolution_ = PrimalSolution();
    performanceIndeces_.clear();
    totalNumIterations_ = 0;
    haveSolution_ = false;       // the next run cold-starts from BipedalRobotInitializer::compute (bpmpc_solver_setup)
  }
  scalar_t getFinalTime() const override { return primalSolution_.timeTrajectory_.empty() ? 0.0 : primalSolution_.timeTrajectory_.back(); }
  void getPrimalSolution(scalar_t /*finalTime*/, PrimalSolution* primalSolutionPtr) const override { *primalSolutionPtr = primalSolution_; }
  /** Values of the state-input equality constraint terms at the solution, per node and per term in registration order
   *  ("<foot>_zeroForce", "<foot>_zeroVelocity", "<foot>_normalVelocity" for each contact, BipedalRobotInterface.cpp:187-191): what the
   *  reference's ConstraintTermObserver on "<foot>_zeroVelocity" reads (BipedalRobotSqpMpcNode.cpp:74-86).  Evaluated on request only
   *  (Settings::computeSolutionMetrics), as the reference says of its observers: debugging, slows the solver down. */
  const ProblemMetrics& getSolutionMetrics() const override { return problemMetrics_; }
  size_t getNumIterations() const override { return totalNumIterations_; }
  const OptimalControlProblem& getOptimalControlProblem() const override { return ocp_; }
  const PerformanceIndex& getPerformanceIndeces() const override { return getIterationsLog().back(); }
  const std::vector<PerformanceIndex>& getIterationsLog() const override {
    if (performanceIndeces_.empty()) throw std::runtime_error("[HipSqpSolver]: No performance log yet, no problem solved yet?");
    return performanceIndeces_;
  }
  // not provided by a multiple-shooting solver; SqpSolver throws from the same entry points
  ScalarFunctionQuadraticApproximation getValueFunction(scalar_t, const vector_t&) const override {
    throw std::runtime_error("[HipSqpSolver] getValueFunction() not available yet.");
  }
  ScalarFunctionQuadraticApproximation getHamiltonian(scalar_t, const vector_t&, const vector_t&) override {
    throw std::runtime_error("[HipSqpSolver] getHamiltonian() not available yet.");
  }
  vector_t getStateInputEqualityConstraintLagrangian(scalar_t, const vector_t&) const override {
    throw std::runtime_error("[HipSqpSolver] getStateInputEqualityConstraintLagrangian() not available yet.");
  }
  MultiplierCollection getIntermediateDualSolution(scalar_t) const override {
    throw std::runtime_error("[HipSqpSolver] getIntermediateDualSolution() not available yet.");
  }

  /** Statistics of the last solve (PerformanceIndex of the reference + step size). */
  const bpmpc_stats& lastStats() const { return stats_; }

 private:
  void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime) override {
    if (static_cast<int>(initState.size()) != nx_) throw std::runtime_error("[HipSqpSolver] state dimension does not match the model");
    // preRun() has updated the reference manager: hand its mode schedule and target trajectories over verbatim
    const ModeSchedule& ms = getReferenceManager().getModeSchedule();
    const TargetTrajectories& tt = getReferenceManager().getTargetTrajectories();
    const std::vector<int> modes(ms.modeSequence.begin(), ms.modeSequence.end());
    const bpmpc_mode_schedule sched{static_cast<int>(ms.eventTimes.size()), ms.eventTimes.data(), modes.data()};
    std::vector<double> states;
    states.reserve(tt.stateTrajectory.size() * nx_);
    for (const vector_t& x : tt.stateTrajectory) {
      if (static_cast<int>(x.size()) != nx_) throw std::runtime_error("[HipSqpSolver] target state dimension does not match the model");
      states.insert(states.end(), x.data(), x.data() + nx_);
    }
    const bpmpc_target target{static_cast<int>(tt.timeTrajectory.size()), tt.timeTrajectory.data(), states.data()};

    const int N = settings_.maxNodes;
    const double horizon = finalTime - initTime;
    t_.assign(N + 1, 0.0);
    x_.assign(static_cast<size_t>(N + 1) * nx_, 0.0);
    u_.assign(static_cast<size_t>(N) * nu_, 0.0);
    K_.assign(static_cast<size_t>(N) * nu_ * nx_, 0.0);
    if (!haveSolution_) {   // first call / after reset(): cold start (BipedalRobotInitializer, src/initialization/BipedalRobotInitializer.cpp:56-63)
      check(bpmpc_solve_batch(solver_, 1, horizon, &initTime, initState.data(), &sched, 1, &target, nullptr, nullptr, t_.data(), x_.data(), u_.data(),
                              K_.data(), &stats_));
    } else {                // MPC loop, mpc.coldStart false (task.info:173): previous solution shifted on the device
      check(bpmpc_solver_setup_from_previous(solver_, 1, horizon, &initTime, initState.data(), &sched, 1, &target));
      check(bpmpc_solver_run(solver_));
      check(bpmpc_solver_fetch(solver_, t_.data(), x_.data(), u_.data(), K_.data(), &stats_));
    }
    if (stats_.status == 2) {   // reaches the catch block of the MPC thread, BipedalController.cpp:344-348
      haveSolution_ = false;
      throw std::runtime_error("[HipSqpSolver] numerical failure in the Riccati sweep (non positive-definite stage Hessian)");
    }
    if (stats_.status == 3) {   // DDP solver: no baseline roll-out to compare the step lengths with
      haveSolution_ = false;
      throw std::runtime_error("[HipSqpSolver] DDP: the roll-out of the nominal policy failed (integrator out of steps, or more time points than max_nodes + 1)");
    }
    haveSolution_ = true;
    fillPrimalSolution(ms);
    if (settings_.computeSolutionMetrics) fillSolutionMetrics();
    PerformanceIndex before, after;
    before.merit = before.cost = stats_.merit_before;
    before.dynamicsViolationSSE = stats_.dynamics_sse_before;
    before.equalityConstraintsSSE = stats_.equality_sse_before;
    after.merit = after.cost = stats_.merit_after;
    after.dynamicsViolationSSE = stats_.dynamics_sse_after;
    after.equalityConstraintsSSE = stats_.equality_sse_after;
    performanceIndeces_.push_back(before);
    performanceIndeces_.push_back(after);
    totalNumIterations_ += static_cast<size_t>(stats_.iterations);
  }
  // The engine warm-starts from its own previous solution (kept on the device); an external controller / primal solution is only
  // used as the signal that one exists - as SqpSolver does when it is handed a non-LinearController
  void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* /*externalControllerPtr*/) override {
    runImpl(initTime, initState, finalTime);
  }
  void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const PrimalSolution& /*primalSolution*/) override {
    runImpl(initTime, initState, finalTime);
  }

  // multiple_shooting::toPrimalSolution [OCS2-upstream, recalled; restated in oracle/reference_py.py primal_solution_arrays]: one
  // entry per node time; the input and gain of the terminal node AND of every pre-event node repeat the previous entry.  The engine
  // stores u = 0, K = 0 at event nodes (they have no input), so a pre-event node - the first of two nodes at the same time - must not
  // be copied verbatim: the LinearController would interpolate forces, joint velocities and gains towards zero before every gait event.
  void fillPrimalSolution(const ModeSchedule& ms) {
    const int n = stats_.n_nodes;
    primalSolution_ = PrimalSolution();
    primalSolution_.modeSchedule_ = ms;
    vector_array_t uff;
    matrix_array_t gains;
    primalSolution_.timeTrajectory_.reserve(n + 1);
    using RowMajorMap = Eigen::Map<const Eigen::Matrix<scalar_t, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>;
    for (int k = 0; k <= n; ++k) {
      const bool preEvent = k < n && t_[k + 1] == t_[k];          // event nodes have zero duration: the grid repeats the event time
      const bool repeat = k > 0 && (k == n || preEvent);
      primalSolution_.timeTrajectory_.push_back(t_[k]);
      primalSolution_.stateTrajectory_.emplace_back(Eigen::Map<const vector_t>(&x_[static_cast<size_t>(k) * nx_], nx_));
      if (repeat) {
        primalSolution_.inputTrajectory_.push_back(primalSolution_.inputTrajectory_.back());
      } else {
        const int ku = std::min(k, n - 1);                        // k == n only with n == 0 ... never (n >= 1); k == 0 is never repeated
        primalSolution_.inputTrajectory_.emplace_back(Eigen::Map<const vector_t>(&u_[static_cast<size_t>(ku) * nu_], nu_));
      }
      if (settings_.useFeedbackPolicy) {
        matrix_t Kk;
        if (repeat) Kk = gains.back();
        else Kk = RowMajorMap(&K_[static_cast<size_t>(std::min(k, n - 1)) * nu_ * nx_], nu_, nx_);
        uff.push_back(primalSolution_.inputTrajectory_.back() - Kk * primalSolution_.stateTrajectory_.back());
        gains.push_back(Kk);
      }
    }
    if (settings_.useFeedbackPolicy) {
      primalSolution_.controllerPtr_.reset(new LinearController(primalSolution_.timeTrajectory_, std::move(uff), std::move(gains)));
    } else {
      primalSolution_.controllerPtr_.reset(new FeedforwardController(primalSolution_.timeTrajectory_, primalSolution_.inputTrajectory_));
    }
  }

  // ProblemMetrics [OCS2-upstream, recalled]: intermediates[k].stateInputEqConstraint = one vector per constraint term in registration order
  // (3 per contact: zeroForce_i, zeroVelocity_i, normalVelocity_i), empty where the term is inactive; event nodes go to preJumps (no terms).
  void fillSolutionMetrics() {
    const int N = settings_.maxNodes, n = stats_.n_nodes;
    std::vector<double> values(static_cast<size_t>(N) * 16);
    std::vector<int> rows(N), modes(N);
    check(bpmpc_solver_constraint_values(solver_, values.data(), rows.data(), modes.data()));
    problemMetrics_.clear();
    for (int k = 0; k < n; ++k) {
      Metrics m;
      if (modes[k] < 0) { problemMetrics_.preJumps.push_back(m); continue; }
      m.stateInputEqConstraint.assign(12, vector_t());
      int row = 0;
      for (int i = 0; i < 4; ++i) {
        const bool stance = i < 2 ? (modes[k] & 1) != 0 : (modes[k] & 2) != 0;
        auto take = [&](int term, int count) {
          m.stateInputEqConstraint[3 * i + term] = Eigen::Map<const vector_t>(&values[static_cast<size_t>(k) * 16 + row], count);
          row += count;
        };
        if (stance) take(1, 3); else { take(0, 3); take(2, 1); }
      }
      problemMetrics_.intermediates.push_back(m);
    }
  }

  static void check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("[HipSqpSolver] bpmpc status ") + std::to_string(rc) + ": " + bpmpc_last_error());
  }

  Settings settings_;
  OptimalControlProblem ocp_;
  bpmpc_model* model_ = nullptr;
  bpmpc_solver* solver_ = nullptr;
  int nx_ = 0, nu_ = 0;
  bool haveSolution_ = false;
  size_t totalNumIterations_ = 0;
  bpmpc_stats stats_{};
  std::vector<double> t_, x_, u_, K_;
  PrimalSolution primalSolution_;
  ProblemMetrics problemMetrics_;
  std::vector<PerformanceIndex> performanceIndeces_;
};

}  // namespace bipedal_robot
}  // namespace ocs2
