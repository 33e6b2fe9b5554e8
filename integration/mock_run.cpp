// Executes the OCS2 adaptor (HipSqpMpc / HipSqpSolver) against libbpmpc.so with the stand-ins of integration/mock_ocs2 in place of OCS2
// (see mock_ocs2/README.md: this checks the ADAPTOR'S logic, it pins nothing about OCS2).  The reference manager is a fixed one: mode
// schedule from the gait API of the library, target trajectory from bpmpc_cmd_vel_to_targets - what SwitchedModelReferenceManager and
// TargetTrajectoriesPublisher would have left there.  Three MPC runs (cold start, then two receding-horizon runs); per run one line of
// checksums of what getPrimalSolution returns.  tests/test_gpu_adaptor_mock_run.py compares with the same solves through the Python mirror.
//   build: g++ -std=c++17 -I include -I integration -I integration/mock_ocs2 integration/mock_run.cpp -L bipedal_control_amd -lbpmpc -Wl,-rpath,... -o mock_run
//   run:   ./mock_run assets/h1 h1_mpc.urdf [task file instead of assets/h1/task.info | -] [ddp]
// With a task file whose sqp.useFeedbackPolicy is false the primal solution carries a FeedforwardController: the line then reports
// "feedforward 1", sb = checksum of its uffArray_ (the input trajectory) and sk = 0.
#include <cstdio>
#include <string>

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

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: mock_run <asset dir> <urdf file name>\n"); return 2; }
  const std::string dir = argv[1];
  const bool ddp = argc > 4 && std::string(argv[4]) == "ddp";      // the DDP counterpart (HipDdpMpc); "-" as the task file = the asset's own
  const std::string urdf = dir + "/" + argv[2], task = (argc > 3 && std::string(argv[3]) != "-") ? std::string(argv[3]) : dir + "/task.info", reference = dir + "/reference.info", gaitfile = dir + "/gait.info";
  try {
    bpmpc_model* model = nullptr;
    if (bpmpc_model_create(urdf.c_str(), task.c_str(), reference.c_str(), &model) != 0) throw std::runtime_error(bpmpc_last_error());
    int nx = 0, nu = 0;
    bpmpc_model_dims(model, &nx, &nu, nullptr, nullptr);
    ocs2::vector_t x0(nx);
    bpmpc_model_get(model, "initial_state", x0.data(), nx);
    const double horizon = 1.005, period = 0.02;             // 67 intervals of 0.015 s
    // reference manager contents: trot from gait.info inserted at -1.225 s, schedule over [-T, 3 T]; command 0.3 m/s forward
    auto refs = std::make_shared<FixedReferences>();
    {
      bpmpc_gait* gait = nullptr;
      bpmpc_gait_create(model, &gait);
      double sw[16]; int modes[16], n_modes = 0;
      if (bpmpc_gait_load_template(gaitfile.c_str(), "trot", sw, modes, 16, &n_modes) != 0) throw std::runtime_error(bpmpc_last_error());
      bpmpc_gait_insert_template(gait, sw, modes, n_modes, -1.225, 3 * horizon);
      double ev[512]; int md[513], n_ev = 0;
      if (bpmpc_gait_mode_schedule(gait, -horizon, 3 * horizon, ev, md, 512, &n_ev) != 0) throw std::runtime_error(bpmpc_last_error());
      refs->ms.eventTimes.assign(ev, ev + n_ev);
      refs->ms.modeSequence.assign(md, md + n_ev + 1);
      bpmpc_gait_destroy(gait);
    }
    ocs2::OptimalControlProblem ocp;
    ocs2::mpc::Settings mpcSettings;
    mpcSettings.timeHorizon_ = horizon;
    mpcSettings.coldStart_ = false;
    ocs2::bipedal_robot::HipSqpSolver::Settings ss;
    ss.maxNodes = 96;
    ss.computeSolutionMetrics = !ddp;         // (the observers read the constraint rows on the shooting grid: not defined on a DDP roll-out)
    std::unique_ptr<ocs2::MPC_BASE> mpc_owner;
    if (ddp) mpc_owner.reset(new ocs2::bipedal_robot::HipDdpMpc(mpcSettings, task, urdf, reference, ocp, ss));      // BipedalRobotDdpMpcNode.cpp:70-71
    else mpc_owner.reset(new ocs2::bipedal_robot::HipSqpMpc(mpcSettings, task, urdf, reference, ocp, ss));
    ocs2::MPC_BASE& mpc = *mpc_owner;
    mpc.getSolverPtr()->setReferenceManager(refs);
    mpc.getSolverPtr()->addSynchronizedModule(std::make_shared<ocs2::SolverSynchronizedModule>());
    const double cmd[4] = {0.3, 0.0, 0.0, 0.0};
    for (int k = 0; k < 3; ++k) {
      const double t = k * period;
      double tt[2];
      std::vector<double> xs(2 * nx);
      if (bpmpc_cmd_vel_to_targets(model, cmd, t, x0.data(), horizon, tt, xs.data()) != 0) throw std::runtime_error(bpmpc_last_error());
      refs->tt.timeTrajectory.assign(tt, tt + 2);
      refs->tt.stateTrajectory.clear();
      for (int p = 0; p < 2; ++p) refs->tt.stateTrajectory.emplace_back(Eigen::Map<const ocs2::vector_t>(&xs[p * nx], nx));
      mpc.run(t, x0);                                          // MPC_BASE::run -> calculateController -> SolverBase::run -> runImpl
      ocs2::PrimalSolution primal;
      mpc.getSolverPtr()->getPrimalSolution(t + horizon, &primal);
      const auto* ctrl = dynamic_cast<const ocs2::LinearController*>(primal.controllerPtr_.get());
      const auto* ffwd = dynamic_cast<const ocs2::FeedforwardController*>(primal.controllerPtr_.get());
      if (!ctrl && !ffwd) throw std::runtime_error("neither a LinearController nor a FeedforwardController in the primal solution");
      const size_t n = primal.timeTrajectory_.size();
      if (primal.stateTrajectory_.size() != n || primal.inputTrajectory_.size() != n ||
          (ctrl && (ctrl->biasArray_.size() != n || ctrl->gainArray_.size() != n)) || (ffwd && (ffwd->uffArray_.size() != n || ffwd->timeStamp_.size() != n)))
        throw std::runtime_error("trajectory lengths differ");
      double st = 0, sx = 0, su = 0, sb = 0, sk = 0;
      for (size_t i = 0; i < n; ++i) {
        st += primal.timeTrajectory_[i] * (1 + i % 3);
        for (int j = 0; j < nx; ++j) sx += primal.stateTrajectory_[i].data()[j] * (1 + (i + j) % 7);
        for (int j = 0; j < nu; ++j) su += primal.inputTrajectory_[i].data()[j] * (1 + (i + j) % 5);
        for (int j = 0; j < nu; ++j) sb += (ctrl ? ctrl->biasArray_[i] : ffwd->uffArray_[i]).data()[j] * (1 + (i + j) % 4);
        if (ctrl)
          for (int a = 0; a < nu; ++a)
            for (int b = 0; b < nx; ++b) sk += ctrl->gainArray_[i](a, b) * (1 + (a + 2 * b) % 3);      // by (row, column): independent of the storage order
      }
      // what a ConstraintTermObserver on "<foot>_zeroVelocity" would see (BipedalRobotSqpMpcNode.cpp:74-86): term 3 i + 1 of every intermediate node
      const ocs2::ProblemMetrics& metrics = mpc.getSolverPtr()->getSolutionMetrics();
      double szv = 0;
      size_t nzv = 0;
      for (size_t i = 0; !ddp && i < metrics.intermediates.size(); ++i)
        for (int c = 0; c < 4; ++c) {
          const ocs2::vector_t& v = metrics.intermediates[i].stateInputEqConstraint[3 * c + 1];
          for (long j = 0; j < v.size(); ++j) { szv += v.data()[j] * (1 + (i + c + j) % 3); ++nzv; }
        }
      const ocs2::PerformanceIndex& perf = mpc.getSolverPtr()->getPerformanceIndeces();
      std::printf("run %d points %zu iterations %zu merit %.17g dyn %.17g st %.17g sx %.17g su %.17g sb %.17g sk %.17g final %.17g nzv %zu szv %.17g prejumps %zu feedforward %d\n", k, n,
                  mpc.getSolverPtr()->getNumIterations(), perf.merit, perf.dynamicsViolationSSE, st, sx, su, sb, sk, mpc.getSolverPtr()->getFinalTime(), nzv, szv,
                  metrics.preJumps.size(), ffwd ? 1 : 0);
    }
    bpmpc_model_destroy(model);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "mock_run: %s\n", e.what());
    return 1;
  }
  return 0;
}
