// Batched SQP solver on one MI355X: device buffers, kernel launches and the solver part of the C ABI (include/bpmpc.h).
// Kernel bodies live in kernels/*.h; this file only maps (problem, node) -> workgroup and owns HBM.
//
// HBM layout (FP64; B = problems, N = max_nodes, slot s = b*N + k, nx = nu = 12 + nj):
//   iterate      x[B][(N+1)][nx], u[B][N][nu]                      coalesced 176-B rows per node
//   LQ model     A,Q [s][nx*nx]  B [s][nx*nu]  R [s][nu*nu]  P [s][nu*nx]  b,q [s][nx]  r [s][nu]  c [s]
//                C [s][16*nx]  D [s][16*nu]  e [s][16]  nc [s]     one contiguous block per node and quantity: every
//                                                                  wavefront streams its node with 512-B bursts
//   projection   Px,Pu,Pe,nut and the projected LQ model At..rt with the same per-node blocking
//   grids        per distinct (t0, schedule): kind, mode, dt, start, zref, zdref  [n_grids][N]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <map>
#include <tuple>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bpmpc.h"
#include "capi_internal.h"
#include "device_model.h"
#include "launch.h"
#include "kernel_launchers.h"
#include "reference_gen.h"
#include "kernels/reference_device.h"
#include "kernels/rollout.h"

namespace bpmpc {

#define HIP_CHECK(expr)                                                                                     \
  do {                                                                                                      \
    hipError_t err_ = (expr);                                                                               \
    if (err_ != hipSuccess) throw DeviceError(std::string(#expr) + ": " + hipGetErrorString(err_));         \
  } while (0)

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };      // -> BPMPC_ERR_UNSUPPORTED

// ------------------------------------------------------------------------------------------------ solver object
struct KernelTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0.0;
  int launches = 0;
};

}  // namespace bpmpc

using namespace bpmpc;

struct HostStaging {   // host-side images of the tables bpmpc_solver_setup uploads
  std::vector<int> kind, mode, nodes, pgrid, tgt_n;
  std::vector<double> gdt, gstart, zref, zdref, tgt_t, tgt_x;
};

// Page-locked host memory for the small transfers of the MPC loop (setup_commands, fetch): a copy from / to pageable memory is
// staged by the runtime and blocks the calling thread for ~10 us each; from / to pinned memory it is only enqueued.  Slices stay
// valid until the next reset(), which the callers issue when everything in flight has been waited for.
struct PinnedArena {
  char* base = nullptr;
  size_t cap = 0, used = 0, demand = 0;
  // start of a new cycle: nothing of the previous one is in flight any more.  Grows to what the previous cycle asked for.
  void reset() {
    if (demand > cap) {
      if (base) (void)hipHostFree(base);
      base = nullptr; cap = 0;
      const size_t want = std::max<size_t>(2 * demand, size_t(1) << 20);
      if (hipHostMalloc(reinterpret_cast<void**>(&base), want) == hipSuccess) cap = want; else base = nullptr;
    }
    used = 0; demand = 0;
  }
  void* take(size_t bytes) {                              // nullptr: no room in this cycle, the caller uses the pageable path
    const size_t at = (used + 63) & ~size_t(63);
    demand = ((demand + 63) & ~size_t(63)) + bytes;
    if (!base || at + bytes > cap) return nullptr;
    used = at + bytes;
    return base + at;
  }
  void release() { if (base) (void)hipHostFree(base); base = nullptr; cap = used = demand = 0; }
};

struct bpmpc_solver {
  HostStaging staging;
  PinnedArena pin_up, pin_down;
  char* xfer = nullptr;                                     // device staging block of the batched small transfers (upload_batch / Downloads)
  size_t xfer_cap = 0;
  RobotModel rm;
  DeviceModel dm;
  DeviceModel* d_model = nullptr;
  bpmpc_settings settings{};
  int nx = 0, nu = 0;
  int batch = 0, n_grids = 0, n_nodes_max = 0;
  int num_cus = 256;                                        // compute units of the device
  // change of variables reading the packed joint rows of the structured elimination: needs an input weight without force / joint-velocity
  // cross terms (checked when the solver is created; BipedalRobotInterface.cpp:239-271 builds it so).  BPMPC_DENSE_PROJECT=1 switches it off.
  bool structured_project = false;
  // Riccati sweep by regime: eight waves with fixed roles while every problem has a CU to itself; four-wave workgroups up to two problems per CU;
  // beyond that one wavefront per problem (riccati_wave.h at one wave per SIMD up to four problems per CU, riccati_wave2.h at two beyond).
  // BPMPC_RICCATI_WAVE: 0 never a wave per problem; 1 (default) as described; 2 riccati_wave.h at every batch size, 4 riccati_wave2.h at every
  // batch size (tests); 3 riccati_wave2.h whenever a wave per problem is used
  int riccati_wave = 1;
  bool lin_compact = true;                                  // BPMPC_LIN_COMPACT=0: the lineariser keeps the event nodes in line (A/B, tests)
  bool force_tables = false;                                // BPMPC_LIN_TABLES=1: the table walks also on a robot of two serial legs (tests)
  bool wt_joint_rows = false;                               // BPMPC_WT_JOINT_ROWS=1: the change of variables always writes the joint rows of Wt (A/B of the byte cut below)
  // which sweep runs the current batch (see launch_riccati)
  bool sweep_wave_regime() const { return riccati_wave == 2 || riccati_wave == 4 || ((riccati_wave == 1 || riccati_wave == 3) && batch > 2 * num_cus); }
  bool sweep_two_per_simd() const { return sweep_wave_regime() && (riccati_wave >= 3 || (riccati_wave == 1 && batch > 4 * num_cus)); }
  // the wave-per-problem sweeps (riccati_wave.h, riccati_wave2.h) and the loaders of the eight-wave sweep (riccati_mfma8.h, PackedStageLoader JR) complete the joint rows of Wt = [At | bt | Bt] from Vt ([I | b | 0] + dt Vt): the change of variables then
  // neither computes nor writes them (3.8 KB per node less each way at the batch sizes where both kernels stream)
  bool sweep_completes_joint_rows() const { return !wt_joint_rows && structured_project && !settings.reference_kernels; }     // every fast sweep does
  // The eight-wave sweep holds a CU per problem; a larger batch runs it in ROUNDS (the dispatcher starts a workgroup as a CU becomes free, so the roll-outs
  // behind the sweeps no longer stream at the same time).  Since its roll-out goes through the ring (round 6) two and three rounds of it beat what those batch
  // sizes ran before on the 22-state robots - batch 512: 0.5447 against 0.5837 ms on the four-wave workgroups, 768: 0.807 against 0.878 on a wave per problem;
  // 1024 in four rounds: 1.075 against 0.917, so from there on the wave sweeps - and lose on nx = 24 (G1 / 512: 0.8545 against 0.7682 on the four-wave
  // workgroups), whose eight-wave stage is 55 % longer (`experiments/LOG.md`).  BPMPC_R8_ROUNDS overrides (1: the regimes of rounds 3 to 5; tests).
  int r8_rounds = 0;                                        // 0: by the robot, as measured
  int eight_wave_rounds() const { return r8_rounds > 0 ? r8_rounds : (rm.nj == 10 ? 3 : 1); }
  bool riccati_double_buffered() const { return riccati_wave != 2 && riccati_wave != 4 && batch <= eight_wave_rounds() * num_cus; }
  bool has_solution = false;                               // a solve has completed on the current setup
  bool has_rollout = false;                                // roll_x holds the end states of a rollout
  bool rollout_unchecked = false;                          // ... whose status flags have not been read back yet
  std::vector<int> grid_kind;                               // host copy of the node kinds of the current setup [n_grids][N]
  int max_rows = kMaxEqRows;                                // largest number of equality rows over the nodes of the current setup
  int max_vel_rows = 12;                                    // ... of rows that constrain a contact velocity (12 double stance, 8 single support, 4 flight)
  bool cold = true;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t producer_stream = nullptr;                   // linearisation + projection of the pipelined horizon chunks
  hipEvent_t ev_go = nullptr;
  std::vector<hipEvent_t> ev_chunk;
  Buffers buf{};
  std::vector<void*> allocations;
  std::map<std::string, std::pair<void*, size_t>> named;   // name -> (device ptr, element count)  (doubles unless in int_named)
  std::map<std::string, bool> is_int;
  std::vector<double> node_times;                          // host copy: [n_grids][N+1]
  std::vector<int> grid_nodes, grid_of_problem;
  std::map<std::string, KernelTimer> timers;
  int* h_remaining = nullptr;                              // pinned
  LineSearchSettings ls{};

  template <typename T>
  T* alloc(const char* name, size_t count, bool integer = false) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, count * sizeof(T)));
    HIP_CHECK(hipMemsetAsync(p, 0, count * sizeof(T), stream));
    allocations.push_back(p);
    if (name) { named[name] = {p, count}; is_int[name] = integer; }
    return static_cast<T*>(p);
  }

  Launch launch_params() const {
    Launch L;
    L.model = d_model;
    L.buf = buf;
    L.batch = batch;
    L.N = settings.max_nodes;
    L.k0 = 0;
    // node range of a launch: the longest grid of the current setup, not the solver's capacity - the per-node kernels map their
    // workgroups onto batch x klen node slots and every slot beyond a problem's grid is a lane group that idles
    L.klen = n_nodes_max > 0 ? n_nodes_max : settings.max_nodes;
    L.cold = cold ? 1 : 0;
    L.serial_legs = (dm.serial_legs && !force_tables) ? 1 : 0;
    L.feedback = feedback();
    L.ls = ls;
    L.reg_prim = settings.reg_prim;
    L.lin_ev_n = -1; L.lin_inter = 0;
    for (int i = 0; i < kLinMaxEvents; ++i) L.lin_ev[i] = 0;
    if (n_grids == 1 && lin_compact && !grid_kind.empty() && (int)grid_nodes.size() == 1) {
      const int n = grid_nodes[0];
      int ne = 0;
      for (int k = 0; k < n; ++k)
        if (grid_kind[k] == 1) { if (ne < kLinMaxEvents) L.lin_ev[ne] = k; ++ne; }
      if (ne <= kLinMaxEvents && n == L.klen) { L.lin_ev_n = ne; L.lin_inter = n - ne; }
    }
    L.ilqr = is_ddp() ? 1 : 0;                                  // the DDP solver: every kernel of the backward pass works on the Euler-discretised model
    L.ilqr_shift = is_ddp() ? rm.ddp.ls_hessian_correction_multiple : 0.0;
    return L;
  }

  // Events that only measure time: no system-scope fence when they complete (hipEventDisableSystemFence: "avoiding the cost of cache writeback and
  // invalidation, and the performance impact of those actions on the execution of following work") - with the default flags the step that carries
  // the roofline kernel's events ran 3 % slower than the steps without them (the kernel behind the lineariser found its inputs flushed from L2)
  static constexpr unsigned kTimingEventFlags = hipEventDisableSystemFence;
  // settings.profile: 0 off, 1 every kernel class, 2 the linearisation kernel only (the roofline measurement of bench.py: every
  // event pair costs one to two microseconds of stream time, ten pairs per solve are 2 % of a step)
  bool timed(const char* cls) const { return settings.profile == 1 || (settings.profile == 2 && std::strcmp(cls, "linearize") == 0); }
  void time_begin(const char* cls, hipEvent_t* a, hipEvent_t* b, hipStream_t on = nullptr) {
    if (!timed(cls)) return;
    HIP_CHECK(hipEventCreateWithFlags(a, kTimingEventFlags));
    if (hipEventCreateWithFlags(b, kTimingEventFlags) != hipSuccess) { (void)hipEventDestroy(*a); throw DeviceError("hipEventCreate failed"); }
    if (hipEventRecord(*a, on ? on : stream) != hipSuccess) { (void)hipEventDestroy(*a); (void)hipEventDestroy(*b); throw DeviceError("hipEventRecord failed"); }
  }
  void time_end(const char* cls, hipEvent_t a, hipEvent_t b, hipStream_t on = nullptr) {
    if (!timed(cls)) return;
    HIP_CHECK(hipEventRecord(b, on ? on : stream));
    KernelTimer& t = timers[cls];
    t.pending.emplace_back(a, b);
    if (t.pending.size() > 4096) collect_timers();   // a profiled loop that never asks for the times must not grow without bound
  }
  // The lineariser's events are attached to its dispatch (kl::linearize_fast): their elapsed time is the kernel's duration, without the barrier
  // packets and the dispatch latency a pair of hipEventRecord calls brackets as well
  void launch_linearize_fast(hipStream_t on, const Launch& L, int nodes) {
    if (!timed("linearize")) { kl::linearize_fast(nj(), settings.materialize_lq != 0, nodes, on, L); return; }
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreateWithFlags(&a, kTimingEventFlags));
    if (hipEventCreateWithFlags(&b, kTimingEventFlags) != hipSuccess) { (void)hipEventDestroy(a); throw DeviceError("hipEventCreate failed"); }
    kl::linearize_fast(nj(), settings.materialize_lq != 0, nodes, on, L, a, b);
    KernelTimer& t = timers["linearize"];
    t.pending.emplace_back(a, b);
    if (t.pending.size() > 4096) collect_timers();
  }
  void collect_timers() {
    for (auto& kv : timers) {
      hipError_t first_error = hipSuccess;                 // the events are destroyed whatever happens; the first failure is reported afterwards
      for (auto& pr : kv.second.pending) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(pr.second);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.first, pr.second);
        if (e == hipSuccess) { kv.second.total_ms += ms; kv.second.launches += 1; }
        else if (first_error == hipSuccess) first_error = e;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
      }
      kv.second.pending.clear();
      if (first_error != hipSuccess) throw DeviceError(std::string("kernel timers: ") + hipGetErrorString(first_error));
    }
  }

  void stage_prepare();
  void stage_linearize();
  void stage_project();
  void stage_riccati();
  void launch_project(hipStream_t on, const Launch& L, int nodes);
  void launch_riccati(const Launch& L);
  void stage_linesearch();
  void pipelined_backward();
  void run_iterations();
  void run_ddp();
  void ddp_nominal_rollout();
  DdpBuffers ddp{};                                       // the DDP slice (settings.solver = BPMPC_SOLVER_DDP)
  bool is_ddp() const { return settings.solver == BPMPC_SOLVER_DDP; }
  // one value for the warm start, the policy rollout and the controller a caller builds (sqp.useFeedbackPolicy / ddp.useFeedbackPolicy of task.info, or the override)
  int feedback() const { return settings.feedback_policy == 1 ? 1 : (settings.feedback_policy == 2 ? 0 : (is_ddp() ? rm.ddp.use_feedback_policy : rm.sqp.use_feedback_policy)); }
  int nj() const { return rm.nj; }
};

// One launch (or a short sequence of launches) bracketed by the events of its kernel class
#define TIMED_ON(on, cls, ...)                                                      \
  do {                                                                              \
    hipEvent_t ev_a_, ev_b_;                                                        \
    time_begin(cls, &ev_a_, &ev_b_, on);                                            \
    __VA_ARGS__;                                                                    \
    time_end(cls, ev_a_, ev_b_, on);                                                \
    HIP_CHECK(hipGetLastError());                                                   \
  } while (0)
#define TIMED(cls, ...) TIMED_ON(stream, cls, __VA_ARGS__)

void bpmpc_solver::stage_prepare() {
  const Launch L = launch_params();
  TIMED("prepare", kl::prepare(nj(), batch * settings.max_nodes, stream, L));
}
void bpmpc_solver::stage_linearize() {
  const Launch L = launch_params();
  if (settings.reference_kernels) TIMED("linearize", kl::linearize_reference(nj(), batch * settings.max_nodes, stream, L));
  else { launch_linearize_fast(stream, L, batch * L.klen); HIP_CHECK(hipGetLastError()); }
}
// constraint elimination + change of variables of `nodes` node slots (the fast kernels)
void bpmpc_solver::launch_project(hipStream_t on, const Launch& L, int nodes) {
  // packed joint rows of [Px | Pe | Pu] when the input weight allows the change of variables to generate the force rows; Px, Pu, Pe for the
  // general one (which packs the joint rows for the sweeps' loaders itself)
  TIMED_ON(on, "project_lu", kl::project_lu_s(nj(), max_vel_rows, structured_project, nodes, on, L));
  TIMED_ON(on, "project", kl::project_fast(nj(), structured_project, !sweep_completes_joint_rows(), nodes, on, L));
}
void bpmpc_solver::stage_project() {
  const Launch L = launch_params();
  if (settings.reference_kernels) TIMED("project", kl::project_reference(nj(), batch * settings.max_nodes, stream, L));
  else launch_project(stream, L, batch * L.klen);
}
void bpmpc_solver::launch_riccati(const Launch& L) {
  if (riccati_double_buffered()) { TIMED("riccati", kl::riccati_fast8(nj(), !sweep_completes_joint_rows(), batch, stream, L)); return; }
  // up to two problems per CU the four-wave workgroups finish in one round (0.61 against 0.91 ms at batch 512); beyond that a wave per
  // problem, four per CU, wins (G1 / 1024: 1.30 against 1.54 ms; 4096: 4.07 against 4.53 ms)
  if (!sweep_wave_regime()) {
    TIMED("riccati", kl::riccati_fast(nj(), false, !sweep_completes_joint_rows(), batch, stream, L));
    return;
  }
  // two waves per SIMD (riccati_wave2.h) need eight problems per CU to fill the chip - the dispatcher packs a CU before it opens the next
  // one, 1024 problems would occupy half of the CUs - and win from the first batch that takes riccati_wave.h a second round: beyond four
  // problems per CU (batch 1024: 0.98 against 1.27 ms; 4096: 3.75 against 3.22 ms)
  TIMED("riccati", { kl::riccati_wave(nj(), sweep_two_per_simd(), !sweep_completes_joint_rows(), batch, stream, L); if (L.k0 == 0) kl::riccati_rollout(nj(), batch, stream, L); });
}
void bpmpc_solver::stage_riccati() {
  const Launch L = launch_params();
  if (settings.reference_kernels) TIMED("riccati", kl::riccati_reference(nj(), batch, stream, L));
  else launch_riccati(L);
}
void bpmpc_solver::stage_linesearch() {
  const Launch L = launch_params();
  if (settings.reference_kernels) HIP_CHECK(hipMemsetAsync(buf.remaining, 0, sizeof(int), stream));   // only the host loop below reads the counter
  hipEvent_t ev_a, ev_b;
  time_begin("linesearch", &ev_a, &ev_b);
  if (settings.reference_kernels) kl::ls_begin(nj(), batch, stream, L);   // fast path: done inside the Riccati kernel
  // alpha = 1, 1/2, ... >= alpha_min  ([OCS2-upstream] SqpSolver::takeStep do-while)
  int max_trials = 0;
  for (double a = 1.0; a >= ls.alpha_min; a *= ls.alpha_decay) ++max_trials;
  if (!settings.reference_kernels) {
    // first round for everybody.  Second round again as a launch over all nodes: a problem that back-tracks once (config 2 tiled from t = 0: one
    // of 256, six of 4096) has its ~107 trial nodes evaluated by seven workgroups at once instead of four passes of ONE workgroup in k_ls_tail
    // (95 -> ~35 us for the whole batch); workgroups of finished problems leave after reading two flags.  Later rounds per problem on the
    // device (k_ls_tail).  No read-back inside a solve.  BPMPC_LS_WIDE_ROUNDS (default 2) = rounds that run as launches over all nodes.
    const int nodes = batch * L.klen;
    static const int wide_rounds = [] { const char* e = std::getenv("BPMPC_LS_WIDE_ROUNDS"); const int v = e ? std::atoi(e) : 2; return v < 1 ? 1 : v; }();
    int round = 0;
    bool pending = false;                                  // the last of these rounds is judged by k_ls_tail (the workgroups of the few problems still open)
    for (; round < wide_rounds && round < max_trials; ++round) {
      kl::trial_fast(nj(), nodes, stream, L);
      pending = round > 0 && (round + 1 == wide_rounds || round + 1 == max_trials);
      if (!pending) kl::ls_decide(nj(), batch, stream, L, round > 0);
    }
    kl::ls_tail(nj(), batch, stream, L, round, max_trials, pending);
    HIP_CHECK(hipGetLastError());
    time_end("linesearch", ev_a, ev_b);
    return;
  }
  for (int t = 0; t < max_trials; ++t) {
    kl::trial_reference(nj(), batch * settings.max_nodes, stream, L);
    kl::ls_decide(nj(), batch, stream, L);
    HIP_CHECK(hipGetLastError());
    // one 4-byte read-back per trial round: stop as soon as every problem has accepted (or given up)
    HIP_CHECK(hipMemcpyAsync(h_remaining, buf.remaining, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (*h_remaining <= 0) break;
  }
  time_end("linesearch", ev_a, ev_b);
}
// Linearisation, projection and Riccati sweep of one SQP iteration with the horizon cut into chunks: the producers
// (linearise, LU, change of variables) of chunk c+1 run on their own stream while the latency-bound Riccati sweep of
// chunk c occupies one workgroup per problem.  Only stream/event ordering is used, no device-side waiting.
void bpmpc_solver::pipelined_backward() {
  const int chunks = settings.pipeline_chunks;
  const int n = n_nodes_max;
  HIP_CHECK(hipEventRecord(ev_go, stream));
  HIP_CHECK(hipStreamWaitEvent(producer_stream, ev_go, 0));
  for (int c = 0; c < chunks; ++c) {
    // chunk c covers [lo, hi), latest stages first
    const int hi = n - (int)((long long)n * c / chunks), lo = n - (int)((long long)n * (c + 1) / chunks);
    if (hi <= lo) continue;
    Launch L = launch_params();
    L.k0 = lo;
    L.klen = hi - lo;
    L.lin_ev_n = -1;                 // a chunk of the horizon: node slot = launch index
    const int nodes = batch * L.klen;
    launch_linearize_fast(producer_stream, L, nodes);
    HIP_CHECK(hipGetLastError());
    launch_project(producer_stream, L, nodes);
    HIP_CHECK(hipEventRecord(ev_chunk[c], producer_stream));
    HIP_CHECK(hipStreamWaitEvent(stream, ev_chunk[c], 0));
    if (c == 0) { L.klen = settings.max_nodes - lo; }   // problems on longer grids than n_nodes_max do not exist; keep k_hi >= N
    launch_riccati(L);
  }
}

void bpmpc_solver::run_iterations() {
  if (rm.nj != 10 && rm.nj != 12) throw std::runtime_error("unsupported joint count");
  if (is_ddp()) {
    // a DDP solution lives on the adaptive time points of its roll-out (k_ddp_finish rewrites x / u there): a second iteration on the same setup
    // would linearise a trajectory that is not on the grid.  setup / setup_from_previous / reset put a nominal trajectory back on the grid.
    if (has_solution) throw Unsupported("DDP: one ILQR iteration per setup (the solution lives on the roll-out's own time points); call setup, setup_from_previous or reset before the next run");
    run_ddp(); has_solution = true; return;
  }
  const int iters = ls.max_iterations;
  for (int it = 0; it < iters; ++it) {
    if (!settings.reference_kernels && settings.pipeline_chunks > 1) {
      pipelined_backward();
    } else {
      stage_linearize();
      stage_project();
      stage_riccati();
    }
    stage_linesearch();
  }
  has_solution = true;
}

// One ILQR iteration of the DDP solver (k_ddp.hip; oracle/ddp_py.py is its restatement, step for step).  Everything is enqueued on the solver's
// stream, nothing is read back or waited for.  Round 6: the backward pass runs on the kernels of the SQP path - the fast lineariser in its ILQR
// form, structured elimination, change of variables on the matrix cores, the Riccati sweep of the batch's regime (launch_params() carries the
// ILQR switch) - and the line search rolls ALL step lengths out in one launch (batch x nv virtual problems), as upstream evaluates them concurrently.
void bpmpc_solver::run_ddp() {
  stage_linearize();            // Euler-discretised LQ model, DIAGONAL_SHIFT on R
  stage_project();              // constraint elimination
  stage_riccati();              // Riccati recursion from S_N = 0; its linear roll-out gives du = lff + K dx
  const Launch L = launch_params();
  kl::ddp_policy(nj(), batch, stream, L, ddp);
  HIP_CHECK(hipGetLastError());
  // line search: the baseline (step length 0: the new gains, no feedforward increment) and maxStepLength, x contractionRate, .. >= minStepLength
  constexpr double kArmijoCoefficient = 1e-4;                                 // [OCS2-upstream] line_search::Settings default (not in task.info)
  const int N = settings.max_nodes;
  RolloutArgs a{};
  a.batch = batch * ddp.nv; a.n_problems = batch; a.lff = ddp.lff;
  for (int v = 0; v < ddp.nv; ++v) a.alpha[v] = ddp.alpha_v[v];
  a.N = N; a.p_grid = buf.p_grid; a.g_nodes = buf.g_nodes; a.g_kind = buf.g_kind; a.g_time = buf.g_time;
  a.x = buf.x; a.u = buf.u; a.K = buf.K; a.x_start = buf.p_x0;
  a.t_start = nullptr; a.duration = -1.0;                                      // every problem over its own horizon [t_0, t_N]
  a.abs_tol = rm.rollout.abs_tol; a.rel_tol = rm.rollout.rel_tol; a.time_step = rm.rollout.time_step;
  a.feedback = 1;                                                            // the search rolls out the FEEDBACK policy whatever controller is handed out
  a.x_end = ddp.end_x; a.u_end = ddp.end_u; a.steps = ddp.roll_steps; a.status = ddp.roll_status;
  a.rec_t = ddp.rec_t; a.rec_x = ddp.rec_x; a.rec_u = ddp.rec_u; a.rec_n = ddp.rec_n; a.rec_cap = ddp.cap;
  double longest = 0.0;
  for (size_t g = 0; g < grid_nodes.size(); ++g) longest = std::max(longest, node_times[g * (N + 1) + grid_nodes[g]] - node_times[g * (N + 1)]);
  a.max_steps = (int)(rm.rollout.max_steps_per_second * std::max(1.0, longest));
  TIMED("ddp_rollout", kl::rollout(rm.nj, dm.serial_legs && !force_tables, batch * ddp.nv, stream, d_model, a));
  TIMED("ddp_search", { kl::ddp_cost(nj(), batch, !settings.reference_kernels, stream, L, ddp); kl::ddp_select(nj(), batch, stream, L, ddp, kArmijoCoefficient); kl::ddp_finish(nj(), batch, stream, L, ddp); });
}

// GaussNewtonDDP::rolloutInitialTrajectory of a warm tick (k_ddp.hip k_ddp_nominal): the shifted previous FeedforwardController integrated from the
// measured state over the new horizon (TimeTriggeredRollout, ODE45), its states interpolated onto the shooting grid as the nominal trajectory.
void bpmpc_solver::ddp_nominal_rollout() {
  const int N = settings.max_nodes;
  RolloutArgs a{};
  a.batch = batch; a.N = N; a.p_grid = buf.p_grid; a.g_nodes = buf.g_nodes; a.g_kind = buf.g_kind; a.g_time = buf.g_time;
  a.x = buf.x; a.u = buf.u; a.K = buf.K; a.x_start = buf.p_x0;
  a.t_start = nullptr; a.duration = -1.0;
  a.abs_tol = rm.rollout.abs_tol; a.rel_tol = rm.rollout.rel_tol; a.time_step = rm.rollout.time_step;
  a.feedback = 0;                                      // ddp.useFeedbackPolicy false: the previous controller is its input trajectory
  a.x_end = ddp.end_x; a.u_end = ddp.end_u; a.steps = ddp.roll_steps; a.status = ddp.roll_status;
  a.rec_t = ddp.rec_t; a.rec_x = ddp.rec_x; a.rec_u = ddp.rec_u; a.rec_n = ddp.rec_n; a.rec_cap = ddp.cap;
  double longest = 0.0;
  for (size_t g = 0; g < grid_nodes.size(); ++g) longest = std::max(longest, node_times[g * (N + 1) + grid_nodes[g]] - node_times[g * (N + 1)]);
  a.max_steps = (int)(rm.rollout.max_steps_per_second * std::max(1.0, longest));
  kl::rollout(rm.nj, dm.serial_legs && !force_tables, batch, stream, d_model, a);
  kl::ddp_nominal(nj(), batch, stream, launch_params(), ddp);
  HIP_CHECK(hipGetLastError());
}

namespace {

int translate(const std::exception& e) {
  set_last_error(e.what());
  if (dynamic_cast<const DeviceError*>(&e)) return BPMPC_ERR_DEVICE;
  if (dynamic_cast<const Unsupported*>(&e)) return BPMPC_ERR_UNSUPPORTED;
  if (dynamic_cast<const std::invalid_argument*>(&e)) return BPMPC_ERR_INVALID_ARGUMENT;
  if (dynamic_cast<const std::length_error*>(&e)) return BPMPC_ERR_CAPACITY;
  return BPMPC_ERR_IO;
}

void allocate(bpmpc_solver* s) {
  const size_t B = s->settings.max_batch, N = s->settings.max_nodes, NX = s->nx, NU = s->nu, S = B * N;
  Buffers& b = s->buf;
  b.qrd = s->alloc<double>("qrd", S * kQrdStride);
  b.lin_park = s->alloc<double>("lin_park", S * kLinParkDoublesPerLane * (s->rm.nj == 10 ? LinFastCfg<10, true>::LPN : LinFastCfg<12, true>::LPN));
  b.lin_dump = s->alloc<double>(nullptr, kLinDumpDoubles);
  b.x_prev = s->alloc<double>("x_prev", B * (N + 1) * NX); b.u_prev = s->alloc<double>("u_prev", S * NU);
  b.K_prev = s->alloc<double>("K_prev", S * NU * NX);
  b.tp_time = s->alloc<double>("tp_time", B * (N + 1)); b.tp_kind = s->alloc<int>("tp_kind", S, true);
  b.tp_nodes = s->alloc<int>("tp_nodes", B, true); b.tp_grid = s->alloc<int>("tp_grid", B, true);
  b.ric_carry = s->alloc<double>("ric_carry", B * (NX * NX + NX + 2));   // S, s, status, scratch word
  b.zero_page = s->alloc<double>(nullptr, 4 + 32);     // {0.0, 1.0, 0.0, 0.0}: zeros and ones of the generated force rows, a 16-byte pair of zeros; then a whole row of zeros (alloc zero-fills)
  { const double one = 1.0; HIP_CHECK(hipMemcpyAsync(b.zero_page + 1, &one, sizeof(double), hipMemcpyHostToDevice, s->stream)); HIP_CHECK(hipStreamSynchronize(s->stream)); }
  b.roll_t = s->alloc<double>(nullptr, B); b.roll_x0 = s->alloc<double>(nullptr, B * NX); b.roll_x = s->alloc<double>("roll_x", B * NX);
  b.roll_u = s->alloc<double>("roll_u", B * NU); b.roll_steps = s->alloc<int>(nullptr, B * 2); b.roll_status = s->alloc<int>(nullptr, B);
  b.g_time = s->alloc<double>("g_time", B * (N + 1)); b.rg_t0 = s->alloc<double>(nullptr, B); b.rg_start = s->alloc<double>(nullptr, B);
  b.p_t0 = s->alloc<double>(nullptr, B); b.p_cmd = s->alloc<double>(nullptr, B * 4); b.lib_d = s->alloc<double>(nullptr, kRefLibCapacity);
  b.rg_gait = s->alloc<int>(nullptr, B); b.rg_status = s->alloc<int>(nullptr, B); b.rg_rows = s->alloc<int>(nullptr, B); b.lib_i = s->alloc<int>(nullptr, kRefLibCapacity);
  b.g_kind = s->alloc<int>("g_kind", S, true); b.g_mode = s->alloc<int>("g_mode", S, true); b.g_nodes = s->alloc<int>("g_nodes", B, true);
  b.g_dt = s->alloc<double>("g_dt", S); b.g_start = s->alloc<double>("g_start", S);
  b.g_zref = s->alloc<double>("g_zref", S * 4); b.g_zdref = s->alloc<double>("g_zdref", S * 4);
  b.p_grid = s->alloc<int>("p_grid", B, true);
  b.p_x0 = s->alloc<double>("x0", B * NX);
  b.p_tgt_t = s->alloc<double>(nullptr, B * kMaxTargetPoints); b.p_tgt_x = s->alloc<double>(nullptr, B * kMaxTargetPoints * NX);
  b.p_tgt_n = s->alloc<int>(nullptr, B);
  b.x = s->alloc<double>("x", B * (N + 1) * NX); b.u = s->alloc<double>("u", S * NU);
  b.x_init = s->alloc<double>("x_init", B * (N + 1) * NX); b.u_init = s->alloc<double>("u_init", S * NU);
  b.xref = s->alloc<double>("xref", S * NX);
  b.A = s->alloc<double>("A", S * NX * NX); b.B = s->alloc<double>("B", S * NX * NU); b.b = s->alloc<double>("b", S * NX);
  b.Q = s->alloc<double>("Q", S * NX * NX); b.R = s->alloc<double>("R", S * NU * NU); b.P = s->alloc<double>("P", S * NU * NX);
  b.q = s->alloc<double>("q", S * NX); b.r = s->alloc<double>("r", S * NU); b.c = s->alloc<double>("c", S);
  b.C = s->alloc<double>("C", S * kMaxEqRows * NX); b.D = s->alloc<double>("D", S * kMaxEqRows * NU); b.e = s->alloc<double>("e", S * kMaxEqRows);
  b.perf = s->alloc<double>("perf", S * 3); b.nc = s->alloc<int>("nc", S, true);
  b.Px = s->alloc<double>("Px", S * NU * NX); b.Pu = s->alloc<double>("Pu", S * NU * NU); b.Pe = s->alloc<double>("Pe", S * NU);
  b.nut = s->alloc<int>("nut", S, true);
  b.n_info = s->alloc<int>("n_info", S, true);
  b.n_aux = s->alloc<double>("n_aux", S * kNodeAux);
  b.At = b.Bt = b.bt = b.Qt = b.Rt = b.Pt = b.qt = b.rt = b.Kt = b.kt = b.Wt = b.Qp = b.Mt = b.Vt = nullptr;
  if (s->settings.reference_kernels) {      // the projected model as plain matrices and the gain scratch of the reference sweep (24 KB per node)
    b.At = s->alloc<double>("At", S * NX * NX); b.Bt = s->alloc<double>("Bt", S * NX * NU); b.bt = s->alloc<double>("bt", S * NX);
    b.Qt = s->alloc<double>("Qt", S * NX * NX); b.Rt = s->alloc<double>("Rt", S * NU * NU); b.Pt = s->alloc<double>("Pt", S * NU * NX);
    b.qt = s->alloc<double>("qt", S * NX); b.rt = s->alloc<double>("rt", S * NU);
    b.Kt = s->alloc<double>("Kt", S * NU * NX); b.kt = s->alloc<double>("kt", S * NU);
  } else {                                  // ... packed, for the fast kernels (22.5 KB per node at nx = 22)
    const size_t wp = ((NX + 1 + NU + 15) / 16) * 16;      // PackedLq<NJ>::WP, QP
    b.Wt = s->alloc<double>("Wt", S * NX * wp); b.Qp = s->alloc<double>("Qp", S * NX * 32); b.Mt = s->alloc<double>("Mt", S * NU * wp);
    b.Vt = s->alloc<double>("Vt", S * (NX - 12) * wp);
  }
  b.dx = s->alloc<double>("dx", B * (N + 1) * NX); b.du = s->alloc<double>("du", S * NU);
  b.K = s->alloc<double>("K", S * NU * NX);   // feedback gains (also the forward roll-out operator of the fast Riccati kernel)
  b.Acl = s->alloc<double>("Acl", S * NX * NX); b.bcl = s->alloc<double>(nullptr, S * NX); b.kff = s->alloc<double>(nullptr, S * NU);
  b.mvec = s->alloc<double>(nullptr, S * NX); b.mscal = s->alloc<double>(nullptr, S);
  b.rprof = s->alloc<double>("rprof", std::max<size_t>(B * 8, 32768));      // debug slots of the profiling builds (phase cycles per problem; workgroup timeline of -DBPMPC_LIN_TIMELINE)
  b.summary = s->alloc<double>("summary", B * 4); b.dx0 = s->alloc<double>(nullptr, B * NX);
  b.trial_perf = s->alloc<double>("trial_perf", S * 3); b.base = s->alloc<double>("base", B * 3); b.alpha = s->alloc<double>("alpha", B);
  b.stats = s->alloc<double>("stats", B * kStatsStride);
  b.done = s->alloc<int>("done", B, true); b.active = s->alloc<int>("active", B, true); b.iterations = s->alloc<int>("iterations", B, true);
  b.remaining = s->alloc<int>(nullptr, 1);
  {   // what one setup_commands moves each way, or one fetch of a small batch (x, u, K, stats: the batch = 1 loop of the reference)
    const size_t tick = B * (64 + NX * 8 + N * 4 + (N + 1) * 8 + 64) + (size_t)kRefLibCapacity * 16 + 4096;
    const size_t solution = (B * ((N + 1) * NX + N * NU + N * NU * NX + kStatsStride)) * sizeof(double) + 4096;
    s->xfer_cap = std::max(tick, std::min<size_t>(size_t(1) << 20, solution));
    s->xfer = s->alloc<char>(nullptr, s->xfer_cap);
  }
  if (s->is_ddp()) {
    DdpBuffers& d = s->ddp;
    d.cap = (int)N + 1;
    {   // step lengths of the line search: the baseline, then maxStepLength, x contractionRate, .. >= minStepLength (task.info:147-155: eight roll-outs)
      constexpr double kContractionRate = 0.5;                                  // [OCS2-upstream] line_search::Settings default (not in task.info)
      d.nv = 0; d.alpha_v[d.nv++] = 0.0;
      for (double al = s->rm.ddp.ls_max_step_length; al >= s->rm.ddp.ls_min_step_length && d.nv < kMaxDdpSteps; al *= kContractionRate) d.alpha_v[d.nv++] = al;
    }
    const size_t P = B * (size_t)d.cap, V = (size_t)d.nv;
    d.lff = s->alloc<double>("ddp_lff", S * NU);
    d.rec_t = s->alloc<double>(nullptr, V * P); d.rec_x = s->alloc<double>(nullptr, V * P * NX); d.rec_u = s->alloc<double>(nullptr, V * P * NU); d.rec_n = s->alloc<int>("ddp_rec_n", V * B, true);
    d.cost = s->alloc<double>(nullptr, V * P * 3);
    d.end_x = s->alloc<double>(nullptr, V * B * NX); d.end_u = s->alloc<double>(nullptr, V * B * NU); d.roll_steps = s->alloc<int>("ddp_roll_steps", V * B * 2, true); d.roll_status = s->alloc<int>("ddp_roll_status", V * B, true);
    d.sol_t = s->alloc<double>("ddp_t", P); d.sol_x = s->alloc<double>(nullptr, P * NX); d.sol_u = s->alloc<double>("ddp_u", P * NU); d.sol_n = s->alloc<int>(nullptr, B);
    d.merit0 = s->alloc<double>(nullptr, B); d.merit = s->alloc<double>(nullptr, B); d.alpha = s->alloc<double>(nullptr, B);
    d.update_is = s->alloc<double>("ddp_update_is", B);
    d.accepted = s->alloc<int>(nullptr, B); d.failed = s->alloc<int>(nullptr, B); d.n_points = s->alloc<int>("ddp_points", B, true);
  }
}

template <typename T>
void upload(bpmpc_solver* s, T* dst, const std::vector<T>& src) {
  if (!src.empty()) HIP_CHECK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s->stream));
}
// the same through the pinned arena (pin_up.reset() by the caller, once nothing of the previous call is in flight)
inline void upload_pinned(bpmpc_solver* s, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return;
  void* stage = s->pin_up.take(bytes);
  if (stage) std::memcpy(stage, src, bytes);
  HIP_CHECK(hipMemcpyAsync(dst, stage ? stage : src, bytes, hipMemcpyHostToDevice, s->stream));
}
template <typename T>
void upload_pinned(bpmpc_solver* s, T* dst, const std::vector<T>& src) { upload_pinned(s, dst, src.data(), src.size() * sizeof(T)); }

// Small transfers in ONE copy.  Every hipMemcpyAsync is a DMA operation of its own on the stream (5 .. 8 us each whatever its size) and a runtime call on the
// host: the nine uploads and five read-backs of a setup_commands and the four results of a fetch were most of what a batch = 1 MPC tick spent outside its
// solve.  Here the pieces travel as one block through `xfer`; a kernel scatters the block to (gathers it from) the arrays the other kernels use.
struct CopyTable {
  static constexpr int kMax = 16;
  void* dst[kMax]; const void* src[kMax]; unsigned words[kMax]; int n;
};
__global__ __launch_bounds__(256) void k_copy_table(CopyTable t) {
  const int e = blockIdx.y;
  const unsigned* src = static_cast<const unsigned*>(t.src[e]);
  unsigned* dst = static_cast<unsigned*>(t.dst[e]);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < t.words[e]; i += gridDim.x * 256) dst[i] = src[i];
}
void launch_copy_table(bpmpc_solver* s, const CopyTable& t) {
  if (t.n == 0) return;
  unsigned most = 0;
  for (int i = 0; i < t.n; ++i) most = std::max(most, t.words[i]);
  hipLaunchKernelGGL(k_copy_table, dim3(std::min(64u, (most + 255) / 256), t.n), dim3(256), 0, s->stream, t);
  HIP_CHECK(hipGetLastError());
}
struct TransferPiece { void* device; const void* host_src; void* host_dst; size_t bytes; };
inline size_t piece_span(size_t bytes) { return (bytes + 15) & ~size_t(15); }
// host -> device, pieces of whole 4-byte words (pin_up.reset() by the caller)
void upload_batch(bpmpc_solver* s, const TransferPiece* pc, int n) {
  size_t total = 0;
  int live = 0;
  bool words = true;
  for (int i = 0; i < n; ++i) if (pc[i].bytes) { total += piece_span(pc[i].bytes); ++live; words = words && pc[i].bytes % 4 == 0; }
  char* pin = (words && live >= 2 && live <= CopyTable::kMax && total <= s->xfer_cap) ? static_cast<char*>(s->pin_up.take(total)) : nullptr;
  if (!pin) {                                             // no room in the arena (its first cycle): piece by piece
    for (int i = 0; i < n; ++i) upload_pinned(s, pc[i].device, pc[i].host_src, pc[i].bytes);
    return;
  }
  CopyTable t{};
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    if (!pc[i].bytes) continue;
    std::memcpy(pin + off, pc[i].host_src, pc[i].bytes);
    t.dst[t.n] = pc[i].device; t.src[t.n] = s->xfer + off; t.words[t.n] = (unsigned)(pc[i].bytes / 4); ++t.n;
    off += piece_span(pc[i].bytes);
  }
  HIP_CHECK(hipMemcpyAsync(s->xfer, pin, off, hipMemcpyHostToDevice, s->stream));
  launch_copy_table(s, t);
}
// device -> host: enqueue() behind the work on the stream, the caller waits for the stream, finish() hands the pieces to their owners
struct Downloads {
  static constexpr size_t kPackLimit = size_t(512) << 10;   // larger pieces travel on their own (a packed piece is copied once more on the device)
  struct Item { TransferPiece pc; void* pin; };
  std::vector<Item> items;
  void add(void* host, const void* device, size_t bytes) { if (host && bytes) items.push_back({{const_cast<void*>(device), nullptr, host, bytes}, nullptr}); }
  void enqueue(bpmpc_solver* s) {
    s->pin_down.reset();
    size_t total = 0;
    int live = 0;
    for (const Item& it : items) if (it.pc.bytes <= kPackLimit && it.pc.bytes % 4 == 0) { total += piece_span(it.pc.bytes); ++live; }
    char* pin = (live >= 2 && live <= CopyTable::kMax && total <= s->xfer_cap) ? static_cast<char*>(s->pin_down.take(total)) : nullptr;
    if (pin) {
      CopyTable t{};
      size_t off = 0;
      for (Item& it : items) {
        if (!(it.pc.bytes <= kPackLimit && it.pc.bytes % 4 == 0)) continue;
        t.dst[t.n] = s->xfer + off; t.src[t.n] = it.pc.device; t.words[t.n] = (unsigned)(it.pc.bytes / 4); ++t.n;
        it.pin = pin + off;
        off += piece_span(it.pc.bytes);
      }
      launch_copy_table(s, t);
      HIP_CHECK(hipMemcpyAsync(pin, s->xfer, off, hipMemcpyDeviceToHost, s->stream));
    }
    for (Item& it : items) {
      if (it.pin) continue;
      void* own = it.pc.bytes <= (size_t(4) << 20) ? s->pin_down.take(it.pc.bytes) : nullptr;      // small results land in pinned memory, large ones in the caller's arrays
      HIP_CHECK(hipMemcpyAsync(own ? own : it.pc.host_dst, it.pc.device, it.pc.bytes, hipMemcpyDeviceToHost, s->stream));
      it.pin = own;
    }
  }
  void finish() const { for (const Item& it : items) if (it.pin) std::memcpy(it.pc.host_dst, it.pin, it.pc.bytes); }
};

void copy_pairs(bpmpc_solver* s, const double* a_src, double* a_dst, size_t na, const double* b_src, double* b_dst, size_t nb, bool rearm) {
  const size_t work = (na > nb ? na : nb) / 2;
  const int grid = (int)std::min<size_t>((work + 255) / 256, 2048);
  kl::copy_pairs(grid, s->stream, a_src, a_dst, na, b_src, b_dst, nb, rearm ? s->buf.iterations : nullptr, s->buf.active, s->batch);
  HIP_CHECK(hipGetLastError());
}

// receding-horizon warm start: keep the previous solution and its grid on the device before anything is overwritten.  The
// solution buffers trade places with the "previous" ones (x and u are rewritten by k_prepare, K by the Riccati sweep for every node
// of the new grid), the grid tables are copied device to device; nothing waits on the host.
void preserve_previous(bpmpc_solver* s, int batch, bool warm_arrays) {
  const size_t N = s->settings.max_nodes, B = batch;
  if (warm_arrays) throw std::invalid_argument("warm start arrays and from_previous are exclusive");
  if (!s->has_solution || batch != s->batch) throw std::invalid_argument("setup_from_previous needs a completed solve of the same batch");
  Buffers& bp = s->buf;
  if (!bp.K) throw std::invalid_argument("setup_from_previous needs the feedback gains (return_gains with reference kernels)");
  std::swap(bp.x, bp.x_prev); std::swap(bp.u, bp.u_prev); std::swap(bp.K, bp.K_prev);
  s->named["x"].first = bp.x; s->named["u"].first = bp.u; s->named["K"].first = bp.K;
  s->named["x_prev"].first = bp.x_prev; s->named["u_prev"].first = bp.u_prev; s->named["K_prev"].first = bp.K_prev;
  {   // the grid tables of the previous solve, one launch instead of four copies
    CopyTable t{};
    auto keep = [&](void* dst, const void* src, size_t bytes) { t.dst[t.n] = dst; t.src[t.n] = src; t.words[t.n] = (unsigned)(bytes / 4); ++t.n; };
    keep(bp.tp_time, bp.g_time, B * (N + 1) * sizeof(double)); keep(bp.tp_kind, bp.g_kind, B * N * sizeof(int));
    keep(bp.tp_nodes, bp.g_nodes, B * sizeof(int)); keep(bp.tp_grid, bp.p_grid, B * sizeof(int));
    launch_copy_table(s, t);
  }
  if (s->is_ddp()) {      // the previous DDP solution lives on the time points of its roll-out, not on the grid it was computed on
    kl::ddp_keep_times(batch, (int)N, s->stream, s->ddp, bp.tp_time, bp.tp_kind, bp.tp_nodes, bp.tp_grid);
    HIP_CHECK(hipGetLastError());
  }
}

// common tail of every setup flavour: initial iterate (initializer, warm arrays or shifted previous solution), activation
void finish_setup(bpmpc_solver* s, int batch, const double* warm_x, const double* warm_u, bool from_previous) {
  const int N = s->settings.max_nodes, NX = s->nx, NU = s->nu;
  Buffers& bf = s->buf;
  if (!s->cold) {
    HIP_CHECK(hipMemcpyAsync(bf.x, warm_x, (size_t)batch * (N + 1) * NX * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_CHECK(hipMemcpyAsync(bf.u, warm_u, (size_t)batch * N * NU * sizeof(double), hipMemcpyHostToDevice, s->stream));
  }
  s->stage_prepare();
  if (from_previous) {
    const Launch L = s->launch_params();
    kl::warm_shift(s->rm.nj, batch * L.N, s->stream, L);
    HIP_CHECK(hipGetLastError());
    if (s->is_ddp()) s->ddp_nominal_rollout();      // the nominal state trajectory of a DDP tick is the roll-out of the previous controller from the measured state
  }
  copy_pairs(s, bf.x, bf.x_init, (size_t)batch * (N + 1) * NX, bf.u, bf.u_init, (size_t)batch * N * NU, true);
  // only the caller's warm-start arrays are still being read at this point: everything else was uploaded before the callers' own
  // synchronisation, and whatever follows on the handle is ordered behind these kernels by the stream
  if (!s->cold) HIP_CHECK(hipStreamSynchronize(s->stream));
}

void setup(bpmpc_solver* s, int batch, double horizon, const double* t0, const double* x0, const bpmpc_mode_schedule* schedules,
           int n_schedules, const bpmpc_target* targets, const double* warm_x, const double* warm_u, bool from_previous = false) {
  if (batch < 1 || batch > s->settings.max_batch) throw std::length_error("batch exceeds the solver's max_batch");
  if (!(horizon > 0) || !t0 || !x0 || !schedules || !targets) throw std::invalid_argument("solve: null or invalid argument");
  if (n_schedules != 1 && n_schedules != batch) throw std::invalid_argument("n_schedules must be 1 or batch");
  if ((warm_x == nullptr) != (warm_u == nullptr)) throw std::invalid_argument("warm_x and warm_u must be given together");
  const int N = s->settings.max_nodes, NX = s->nx;
  const double dt = s->settings.dt > 0 ? s->settings.dt : s->rm.sqp.dt;
  if (n_schedules == 1)
    for (int b = 1; b < batch; ++b)
      if (t0[b] != t0[0]) throw std::invalid_argument("a shared schedule needs identical t0 for all problems");
  // One grid (time discretisation, modes, swing references) per DISTINCT (t0, mode schedule): a gait-library sweep hands over
  // thousands of problems but only a handful of schedules, and the host pre-pass is per grid.
  std::vector<int> grid_of(batch, 0), first_problem;       // first_problem[g]: a problem that owns grid g
  {
    std::map<std::string, int> seen;
    for (int b = 0; b < batch; ++b) {
      const bpmpc_mode_schedule& sc = schedules[n_schedules == 1 ? 0 : b];
      if (sc.n_events < 0 || !sc.modes || (sc.n_events > 0 && !sc.event_times)) throw std::invalid_argument("invalid mode schedule");
      std::string key(reinterpret_cast<const char*>(&t0[b]), sizeof(double));
      key.append(reinterpret_cast<const char*>(sc.event_times), sizeof(double) * sc.n_events);
      key.append(reinterpret_cast<const char*>(sc.modes), sizeof(int) * (sc.n_events + 1));
      auto it = seen.find(key);
      if (it == seen.end()) { it = seen.emplace(key, (int)first_problem.size()).first; first_problem.push_back(b); }
      grid_of[b] = it->second;
      if (n_schedules == 1) { std::fill(grid_of.begin(), grid_of.end(), 0); break; }
    }
  }
  const int G = (int)first_problem.size();
  const size_t S = (size_t)G * N;
  // staging vectors live in the solver: releasing megabytes of freshly DMA-ed pageable memory after every setup made the
  // next solve stall for 10-30 ms on some boxes (tools/setup_time_probe.py)
  HostStaging& hs = s->staging;
  std::vector<int>&kind = hs.kind, &mode = hs.mode, &nodes = hs.nodes, &pgrid = hs.pgrid;
  std::vector<double>&gdt = hs.gdt, &gstart = hs.gstart, &zref = hs.zref, &zdref = hs.zdref;
  kind.assign(S, 0); mode.assign(S, STANCE); nodes.assign(G, 0); pgrid.assign(batch, 0);
  gdt.assign(S, 0.0); gstart.assign(S, 0.0); zref.assign(S * 4, 0.0); zdref.assign(S * 4, 0.0);
  // (the handle's own node times are replaced only after every check below has passed: a rejected call leaves the handle as it was)
  std::vector<double> node_times((size_t)G * (N + 1), 0.0);
  SwingPlanner planner(s->rm.swing);
  int nmax = 0, rows_max = 12, vrows_max = 4;
  for (int g = 0; g < G; ++g) {
    const bpmpc_mode_schedule& sc = schedules[n_schedules == 1 ? 0 : first_problem[g]];
    ModeSchedule ms;
    ms.event_times.assign(sc.event_times, sc.event_times + sc.n_events);
    ms.modes.assign(sc.modes, sc.modes + sc.n_events + 1);
    planner.update(ms);
    const double ts = t0[first_problem[g]];
    const NodeTable tab = build_node_table(s->rm, ts, ts + horizon, dt, ms, planner);
    if (tab.N > N) throw std::length_error("time grid has " + std::to_string(tab.N) + " intervals, solver max_nodes is " + std::to_string(N));
    nodes[g] = tab.N;
    nmax = std::max(nmax, tab.N);
    for (int k = 0; k < tab.N; ++k) {
      const size_t i = (size_t)g * N + k;
      kind[i] = tab.kind[k]; mode[i] = tab.mode[k]; gdt[i] = tab.dt[k]; gstart[i] = tab.start[k];
      if (tab.kind[k] == 0) {                            // rows per contact: 3 in stance, 4 in swing (two contact points per foot)
        const int left = (tab.mode[k] == 1 || tab.mode[k] == 3) ? 3 : 4, right = (tab.mode[k] == 2 || tab.mode[k] == 3) ? 3 : 4;
        rows_max = std::max(rows_max, 2 * left + 2 * right);
        vrows_max = std::max(vrows_max, (left == 3 ? 6 : 2) + (right == 3 ? 6 : 2));
      }
      for (int c = 0; c < 4; ++c) { zref[4 * i + c] = tab.zref[4 * k + c]; zdref[4 * i + c] = tab.zdref[4 * k + c]; }
    }
    std::copy(tab.node_time.begin(), tab.node_time.end(), node_times.begin() + (size_t)g * (N + 1));
  }
  std::vector<double>&tgt_t = hs.tgt_t, &tgt_x = hs.tgt_x;
  std::vector<int>& tgt_n = hs.tgt_n;
  tgt_t.assign((size_t)batch * kMaxTargetPoints, 0.0); tgt_x.assign((size_t)batch * kMaxTargetPoints * NX, 0.0); tgt_n.assign(batch, 0);
  for (int b = 0; b < batch; ++b) {
    pgrid[b] = grid_of[b];
    const bpmpc_target& t = targets[b];
    if (t.n_points < 1 || !t.times || !t.states) throw std::invalid_argument("target trajectories need at least one point");
    // A TargetTrajectories of any length (the ROS reference manager hands its own over verbatim, integration/HipSqpSolver.h): only the
    // points that the piecewise-linear interpolation can touch for query times in [t0, t0 + horizon] are kept - the last one before the
    // window, the ones inside, the first one behind it (interpolate_targets: segment [lower_bound(t) - 1, lower_bound(t)], constant
    // extrapolation) - which gives bit-identical references.  The device tables hold kMaxTargetPoints of them.
    const double lo = t0[b], hi = t0[b] + horizon;
    const int i0 = std::max(0, static_cast<int>(std::lower_bound(t.times, t.times + t.n_points, lo) - t.times) - 1);
    const int i1 = std::min(t.n_points - 1, static_cast<int>(std::lower_bound(t.times, t.times + t.n_points, hi) - t.times));
    const int np = i1 - i0 + 1;
    if (np > kMaxTargetPoints)
      throw std::invalid_argument("target trajectory has " + std::to_string(np) + " points inside the horizon, the solver holds " + std::to_string(kMaxTargetPoints));
    tgt_n[b] = np;
    std::copy(t.times + i0, t.times + i0 + np, tgt_t.begin() + (size_t)b * kMaxTargetPoints);
    std::copy(t.states + (size_t)i0 * NX, t.states + (size_t)(i0 + np) * NX, tgt_x.begin() + (size_t)b * kMaxTargetPoints * NX);
  }
  // every host-side check (grids, targets) has passed: only now do the solution buffers trade places with the kept copy, so a
  // rejected call leaves the handle exactly as it was
  if (from_previous) preserve_previous(s, batch, warm_x != nullptr);
  s->node_times.swap(node_times);
  s->batch = batch; s->n_grids = G; s->n_nodes_max = nmax; s->cold = (warm_x == nullptr);
  s->max_rows = rows_max; s->max_vel_rows = vrows_max;
  s->grid_nodes = nodes; s->grid_of_problem = pgrid; s->grid_kind = kind; s->has_solution = false;
  Buffers& bf = s->buf;
  {
    s->pin_up.reset();                                    // the previous call waited for its transfers (the synchronisation below)
    auto piece = [](auto* dev, const auto& v) { return TransferPiece{dev, v.data(), nullptr, v.size() * sizeof(v[0])}; };
    const TransferPiece up[13] = {piece(bf.g_kind, kind), piece(bf.g_mode, mode), piece(bf.g_nodes, nodes), piece(bf.g_dt, gdt), piece(bf.g_start, gstart),
                                  piece(bf.g_zref, zref), piece(bf.g_zdref, zdref), piece(bf.p_grid, pgrid), piece(bf.p_tgt_t, tgt_t), piece(bf.p_tgt_x, tgt_x),
                                  piece(bf.p_tgt_n, tgt_n), piece(bf.g_time, s->node_times), {bf.p_x0, x0, nullptr, (size_t)batch * NX * sizeof(double)}};
    upload_batch(s, up, 13);
  }
  HIP_CHECK(hipStreamSynchronize(s->stream));  // the host staging vectors go out of scope
  finish_setup(s, batch, warm_x, warm_u, from_previous);
}

void check_rollout_status(bpmpc_solver* s, int* steps);

// Device-side reference generation: the same tables as setup(), built on the GPU from gait
// templates and velocity commands; the host only groups problems by (t0, gait, gait start) and reads the grid sizes back.
void setup_commands(bpmpc_solver* s, int batch, double horizon, const double* t0, const double* x0, const bpmpc_gait_template* gaits, int n_gaits,
                    const int* gait_of_problem, const double* gait_start, const double* cmd_vel, int command_kind, double time_to_target,
                    bool from_previous) {
  if (batch < 1 || batch > s->settings.max_batch) throw std::length_error("batch exceeds the solver's max_batch");
  if (!(horizon > 0) || !t0 || !cmd_vel || n_gaits < 0 || (n_gaits > 0 && !gaits)) throw std::invalid_argument("setup_commands: null or invalid argument");
  if (!x0 && (!s->has_rollout || batch != s->batch)) throw std::invalid_argument("setup_commands: x0 == NULL needs a rollout of the same batch on the handle");
  if (!x0 && s->rollout_unchecked) check_rollout_status(s, nullptr);
  if (command_kind != 0 && command_kind != 1) throw std::invalid_argument("setup_commands: command_kind is 0 (velocity) or 1 (goal pose)");
  if (n_gaits > 0 && (!gait_of_problem || !gait_start)) throw std::invalid_argument("setup_commands: gait_of_problem and gait_start are needed with templates");
  const int N = s->settings.max_nodes, NX = s->nx;
  const double dt = s->settings.dt > 0 ? s->settings.dt : s->rm.sqp.dt;
  std::vector<int> pgrid(batch), ggait;
  std::vector<double> gt0, gstart;
  {
    std::map<std::tuple<double, int, double>, int> seen;
    for (int b = 0; b < batch; ++b) {
      const int gi = gait_of_problem ? gait_of_problem[b] : -1;
      if (gi >= n_gaits) throw std::invalid_argument("gait_of_problem refers to a template that was not passed");
      const double st = gi >= 0 ? gait_start[b] : 0.0;
      auto key = std::make_tuple(t0[b], gi < 0 ? -1 : gi, st);
      auto it = seen.find(key);
      if (it == seen.end()) { it = seen.emplace(key, (int)gt0.size()).first; gt0.push_back(t0[b]); ggait.push_back(gi < 0 ? -1 : gi); gstart.push_back(st); }
      pgrid[b] = it->second;
    }
  }
  const int G = (int)gt0.size();
  // gait library: the passed templates, then defaultModeSequenceTemplate; initialModeSchedule behind them
  std::vector<double> lib_d;
  std::vector<int> first_mode{0}, lib_modes;
  auto add_template = [&](const double* sw, const int* modes, int n) {
    if (n < 0 || (n > 0 && (!sw || !modes))) throw std::invalid_argument("invalid gait template");
    lib_d.insert(lib_d.end(), sw, sw + (n > 0 ? n + 1 : 0));
    if (n == 0) lib_d.push_back(0.0);
    lib_modes.insert(lib_modes.end(), modes, modes + n);
    first_mode.push_back((int)lib_modes.size());
  };
  for (int g = 0; g < n_gaits; ++g) add_template(gaits[g].switching_times, gaits[g].modes, gaits[g].n_modes);
  const ModeTemplate& dflt = s->rm.default_template;
  if (!dflt.modes.empty() && dflt.switching_times.size() != dflt.modes.size() + 1) throw std::invalid_argument("default gait template is malformed");
  add_template(dflt.switching_times.data(), dflt.modes.data(), (int)dflt.modes.size());
  const ModeSchedule& init = s->rm.initial_mode_schedule;
  const size_t sw_count = lib_d.size(), mode_count = lib_modes.size();
  lib_d.insert(lib_d.end(), init.event_times.begin(), init.event_times.end());
  std::vector<int> lib_i = first_mode;
  lib_i.insert(lib_i.end(), lib_modes.begin(), lib_modes.end());
  lib_i.insert(lib_i.end(), init.modes.begin(), init.modes.end());
  if (lib_d.size() > (size_t)kRefLibCapacity || lib_i.size() > (size_t)kRefLibCapacity) throw std::length_error("gait library exceeds the device capacity");
  if (from_previous) preserve_previous(s, batch, false);
  Buffers& bf = s->buf;
  s->pin_up.reset();                                      // the previous call waited for its transfers (the synchronisation below)
  {
    const TransferPiece up[9] = {{bf.rg_t0, gt0.data(), nullptr, gt0.size() * sizeof(double)}, {bf.rg_gait, ggait.data(), nullptr, ggait.size() * sizeof(int)},
                                 {bf.rg_start, gstart.data(), nullptr, gstart.size() * sizeof(double)}, {bf.lib_d, lib_d.data(), nullptr, lib_d.size() * sizeof(double)},
                                 {bf.lib_i, lib_i.data(), nullptr, lib_i.size() * sizeof(int)}, {bf.p_grid, pgrid.data(), nullptr, pgrid.size() * sizeof(int)},
                                 {bf.p_t0, t0, nullptr, (size_t)batch * sizeof(double)}, {bf.p_cmd, cmd_vel, nullptr, (size_t)batch * 4 * sizeof(double)},
                                 {bf.p_x0, x0, nullptr, x0 ? (size_t)batch * NX * sizeof(double) : 0}};
    upload_batch(s, up, 9);
  }
  if (!x0) HIP_CHECK(hipMemcpyAsync(bf.p_x0, bf.roll_x, (size_t)batch * NX * sizeof(double), hipMemcpyDeviceToDevice, s->stream));   // closed loop on the device
  ReferenceGenArgs a{};
  a.lib.switching = bf.lib_d; a.lib.first_mode = bf.lib_i; a.lib.modes = bf.lib_i + first_mode.size(); a.lib.n_templates = n_gaits + 1;
  a.lib.init_events = bf.lib_d + sw_count; a.lib.init_modes = bf.lib_i + first_mode.size() + mode_count; a.lib.init_n_events = (int)init.event_times.size();
  a.lib.transition_stance_time = s->rm.phase_transition_stance_time;
  a.n_grids = G; a.N = N; a.horizon = horizon; a.dt = dt; a.dt_min = 1e-8;
  a.t0 = bf.rg_t0; a.gait = bf.rg_gait; a.gait_start = bf.rg_start;
  a.lift_off_velocity = s->rm.swing.lift_off_velocity; a.touch_down_velocity = s->rm.swing.touch_down_velocity;
  a.swing_height = s->rm.swing.swing_height; a.swing_time_scale = s->rm.swing.swing_time_scale;
  a.kind = bf.g_kind; a.mode = bf.g_mode; a.nodes = bf.g_nodes; a.status = bf.rg_status; a.rows = bf.rg_rows;
  a.gdt = bf.g_dt; a.gstart = bf.g_start; a.zref = bf.g_zref; a.zdref = bf.g_zdref; a.node_time = bf.g_time;
  hipLaunchKernelGGL(k_reference_grids, dim3(G), dim3(64), 0, s->stream, a);
  HIP_CHECK(hipGetLastError());
  CommandTargetArgs c{};
  c.batch = batch; c.nx = NX; c.nj = s->rm.nj; c.time_to_target = time_to_target > 0 ? time_to_target : horizon; c.com_height = s->rm.com_height;
  c.goal = command_kind; c.displacement_velocity = s->rm.target_displacement_velocity; c.rotation_velocity = s->rm.target_rotation_velocity;
  for (int j = 0; j < s->rm.nj; ++j) c.default_joint_state[j] = s->rm.default_joint_state[j];
  c.t0 = bf.p_t0; c.x0 = bf.p_x0; c.cmd_vel = bf.p_cmd; c.tgt_t = bf.p_tgt_t; c.tgt_x = bf.p_tgt_x; c.tgt_n = bf.p_tgt_n;
  hipLaunchKernelGGL(k_command_targets, dim3((batch + 63) / 64), dim3(64), 0, s->stream, c);
  HIP_CHECK(hipGetLastError());
  std::vector<int> nodes(G), status(G), rows(G), kind((size_t)G * N);
  std::vector<double> node_times((size_t)G * (N + 1), 0.0);   // becomes the handle's copy once every grid has been accepted
  {
    Downloads down;
    down.add(nodes.data(), bf.g_nodes, G * sizeof(int)); down.add(status.data(), bf.rg_status, G * sizeof(int)); down.add(rows.data(), bf.rg_rows, G * sizeof(int));
    down.add(kind.data(), bf.g_kind, kind.size() * sizeof(int)); down.add(node_times.data(), bf.g_time, node_times.size() * sizeof(double));
    down.enqueue(s);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    down.finish();
  }
  s->has_solution = false;
  s->batch = 0;                                           // stays unusable if a grid is rejected below
  int nmax = 0, rows_max = 12, vrows_max = 4;
  for (int g = 0; g < G; ++g) {
    const int st = status[g];
    if (st == kRefTileOrder) throw std::runtime_error("The initial time for template-tiling is not greater than the last event time.");
    if (st == kRefCapacity) throw std::length_error("mode schedule exceeds the device capacity of " + std::to_string(kRefMaxEvents) + " events");
    if (st == kRefGridTooLong) throw std::length_error("time grid has " + std::to_string(nodes[g]) + " intervals, solver max_nodes is " + std::to_string(N));
    if (st >= kRefNoTouchDown) throw std::runtime_error("The time of touch-down for the last swing of the EE with ID " + std::to_string(st - kRefNoTouchDown) + " is not defined.");
    if (st >= kRefNoTakeOff) throw std::runtime_error("The time of take-off for the first swing of the EE with ID " + std::to_string(st - kRefNoTakeOff) + " is not defined.");
    nmax = std::max(nmax, nodes[g]);
    rows_max = std::max(rows_max, rows[g] & 255);
    vrows_max = std::max(vrows_max, rows[g] >> 8);
  }
  s->node_times.swap(node_times);
  s->batch = batch; s->n_grids = G; s->n_nodes_max = nmax; s->cold = true; s->max_rows = rows_max; s->max_vel_rows = vrows_max;
  s->grid_nodes = nodes; s->grid_of_problem = pgrid; s->grid_kind = kind;
  finish_setup(s, batch, nullptr, nullptr, from_previous);
}


// reads the per-problem flags of the last rollout back (synchronises) and reports failures like the reference's integrator does
void check_rollout_status(bpmpc_solver* s, int* steps) {
  const int B = s->batch;
  std::vector<int> status(B), st(2 * B);
  HIP_CHECK(hipMemcpyAsync(status.data(), s->buf.roll_status, B * sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIP_CHECK(hipMemcpyAsync(st.data(), s->buf.roll_steps, 2 * B * sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  s->rollout_unchecked = false;
  if (steps) std::copy(st.begin(), st.end(), steps);
  for (int b = 0; b < B; ++b) {
    if (status[b] == 1) throw std::runtime_error("rollout of problem " + std::to_string(b) + ": integration terminated, max number of steps reached");
    if (status[b] == 2) throw std::runtime_error("rollout of problem " + std::to_string(b) + ": max number of iterations exceeded, a new step size was not found");
    if (status[b] == 3) throw std::length_error("rollout of problem " + std::to_string(b) + ": more than " + std::to_string(kRolloutMaxEvents) + " events in the window");
  }
}

// MRT_BASE::rolloutPolicy for the whole batch (kernels/rollout.h): integrates every problem from (t_start, x_start) over `duration`
// under the LinearController of the last solve.  NULL t_start / x_start: the initial time / measured state of that solve.
void rollout(bpmpc_solver* s, const double* t_start, const double* x_start, double duration, double* x_end, double* u_end, int* steps) {
  if (!s->has_solution) throw std::invalid_argument("bpmpc_solver_rollout needs a completed solve on the handle");
  if (s->is_ddp()) throw Unsupported("bpmpc_solver_rollout: the DDP solution is a FeedforwardController on the time points of its own roll-out (fetch them with bpmpc_solver_fetch), not on the shooting grid this roll-out interpolates on");
  if (!(duration >= 0)) throw std::invalid_argument("bpmpc_solver_rollout: negative duration");
  Buffers& bf = s->buf;
  if (!bf.K) throw std::invalid_argument("bpmpc_solver_rollout needs the feedback gains (return_gains with reference kernels)");
  const int B = s->batch, N = s->settings.max_nodes, NX = s->nx, NU = s->nu;
  std::vector<double> ts(B);
  for (int b = 0; b < B; ++b) ts[b] = t_start ? t_start[b] : s->node_times[(size_t)s->grid_of_problem[b] * (N + 1)];
  upload(s, bf.roll_t, ts);
  if (x_start) HIP_CHECK(hipMemcpyAsync(bf.roll_x0, x_start, (size_t)B * NX * sizeof(double), hipMemcpyHostToDevice, s->stream));
  RolloutArgs a{};
  a.batch = B; a.N = N; a.p_grid = bf.p_grid; a.g_nodes = bf.g_nodes; a.g_kind = bf.g_kind; a.g_time = bf.g_time;
  a.x = bf.x; a.u = bf.u; a.K = bf.K; a.t_start = bf.roll_t; a.x_start = x_start ? bf.roll_x0 : bf.p_x0;
  a.duration = duration; a.abs_tol = s->rm.rollout.abs_tol; a.rel_tol = s->rm.rollout.rel_tol; a.time_step = s->rm.rollout.time_step;
  a.max_steps = (int)(s->rm.rollout.max_steps_per_second * std::max(1.0, duration));
  a.feedback = s->feedback();
  a.x_end = bf.roll_x; a.u_end = bf.roll_u; a.steps = bf.roll_steps; a.status = bf.roll_status;
  kl::rollout(s->rm.nj, s->dm.serial_legs && !s->force_tables, B, s->stream, s->d_model, a);
  HIP_CHECK(hipGetLastError());
  s->has_rollout = true;
  if (!x_end && !u_end && !steps) {                 // nothing to hand back: stay asynchronous; the status is looked at by the next
    s->rollout_unchecked = true;                    // setup_commands(x0 = NULL), which reads its own flags back anyway
    return;
  }
  check_rollout_status(s, steps);
  if (x_end) HIP_CHECK(hipMemcpy(x_end, bf.roll_x, (size_t)B * NX * sizeof(double), hipMemcpyDeviceToHost));
  if (u_end) HIP_CHECK(hipMemcpy(u_end, bf.roll_u, (size_t)B * NU * sizeof(double), hipMemcpyDeviceToHost));
}

void reset(bpmpc_solver* s) {
  const size_t N = s->settings.max_nodes;
  if (s->is_ddp()) s->has_solution = false;      // the initial iterate of the setup is back on the grid: the next run is a first iteration again
  copy_pairs(s, s->buf.x_init, s->buf.x, (size_t)s->batch * (N + 1) * s->nx, s->buf.u_init, s->buf.u, (size_t)s->batch * N * s->nu, true);
}

void fetch(bpmpc_solver* s, double* out_t, double* out_x, double* out_u, double* out_K, bpmpc_stats* stats) {
  const size_t N = s->settings.max_nodes, NX = s->nx, NU = s->nu, B = s->batch;
  std::vector<double> raw(stats ? B * kStatsStride : 0);
  {
    // all transfers enqueued behind the solve, one wait; small results (the batch = 1 loop of the reference) land in pinned memory
    // and are copied out by the host, large ones go straight to the caller's (pageable) arrays
    Downloads down;
    down.add(out_x, s->buf.x, B * (N + 1) * NX * sizeof(double)); down.add(out_u, s->buf.u, B * N * NU * sizeof(double));
    down.add(out_K, s->buf.K, B * N * NU * NX * sizeof(double)); down.add(stats ? raw.data() : nullptr, s->buf.stats, raw.size() * sizeof(double));
    down.enqueue(s);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    down.finish();
  }
  std::vector<int> ddp_points;
  if (s->is_ddp() && s->has_solution) {       // the solution lives on the time points of the accepted roll-out (k_ddp_finish); a failed problem keeps its grid
    ddp_points.resize(B);
    HIP_CHECK(hipMemcpy(ddp_points.data(), s->ddp.n_points, B * sizeof(int), hipMemcpyDeviceToHost));
  }
  if (out_t) {
    std::vector<double> tp;
    if (!ddp_points.empty()) { tp.resize(B * (size_t)s->ddp.cap); HIP_CHECK(hipMemcpy(tp.data(), s->ddp.sol_t, tp.size() * sizeof(double), hipMemcpyDeviceToHost)); }
    for (size_t b = 0; b < B; ++b) {
      if (!ddp_points.empty() && ddp_points[b] > 0) { std::copy(tp.begin() + b * s->ddp.cap, tp.begin() + b * s->ddp.cap + ddp_points[b], out_t + b * (N + 1)); continue; }
      const int g = s->grid_of_problem[b];
      std::copy(s->node_times.begin() + (size_t)g * (N + 1), s->node_times.begin() + (size_t)g * (N + 1) + s->grid_nodes[g] + 1, out_t + b * (N + 1));
    }
  }
  if (stats) {
    for (size_t b = 0; b < B; ++b) {
      const double* r = &raw[b * kStatsStride];
      bpmpc_stats& st = stats[b];
      st.n_nodes = (!ddp_points.empty() && ddp_points[b] > 0) ? ddp_points[b] - 1 : s->grid_nodes[s->grid_of_problem[b]];
      st.iterations = (int)r[1]; st.status = (int)r[2]; st.reserved = 0;
      st.merit_before = r[3]; st.dynamics_sse_before = r[4]; st.equality_sse_before = r[5];
      st.merit_after = r[6]; st.dynamics_sse_after = r[7]; st.equality_sse_after = r[8];
      st.step_size = r[9]; st.armijo_descent = r[10]; st.dx_norm = r[11]; st.du_norm = r[12];
    }
  }
  s->collect_timers();   // the stream is idle here; a closed loop of advance() calls never reaches bpmpc_solver_sync
}

}  // namespace

extern "C" {

int bpmpc_solver_create(const bpmpc_model* model, const bpmpc_settings* settings, bpmpc_solver** out) {
  if (!model || !settings || !out) { set_last_error("bpmpc_solver_create: null argument"); return BPMPC_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  if (settings->max_batch < 1 || settings->max_nodes < 1) { set_last_error("bpmpc_solver_create: max_batch and max_nodes must be positive"); return BPMPC_ERR_INVALID_ARGUMENT; }
  if (settings->max_nodes > kMaxRiccatiStages) { set_last_error("bpmpc_solver_create: max_nodes exceeds 512"); return BPMPC_ERR_CAPACITY; }
  if ((long long)settings->max_batch * settings->max_nodes > 0x3fffffffLL) { set_last_error("bpmpc_solver_create: max_batch * max_nodes exceeds 2^30"); return BPMPC_ERR_CAPACITY; }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || settings->device < 0 || settings->device >= count) {
    set_last_error("bpmpc_solver_create: no usable HIP device (this engine has no CPU path)");
    return BPMPC_ERR_NO_DEVICE;
  }
  std::unique_ptr<bpmpc_solver> s(new bpmpc_solver);
  try {
    s->rm = model_of(model);
    if (s->rm.nj != 10 && s->rm.nj != 12) { set_last_error("only 10- and 12-joint bipeds are instantiated"); return BPMPC_ERR_UNSUPPORTED; }
    if (settings->reg_prim != 0.0 && settings->reference_kernels) { set_last_error("reg_prim is implemented by the fast kernels only"); return BPMPC_ERR_UNSUPPORTED; }
    if (!(settings->reg_prim >= 0.0)) { set_last_error("reg_prim must be >= 0"); return BPMPC_ERR_INVALID_ARGUMENT; }
    s->dm = make_device_model(s->rm);
    s->settings = *settings;
    if (settings->solver != BPMPC_SOLVER_SQP && settings->solver != BPMPC_SOLVER_DDP) { set_last_error("bpmpc_solver_create: unknown solver"); return BPMPC_ERR_INVALID_ARGUMENT; }
    if (settings->feedback_policy < 0 || settings->feedback_policy > 2) { set_last_error("bpmpc_solver_create: feedback_policy must be 0, 1 or 2"); return BPMPC_ERR_INVALID_ARGUMENT; }
    if (settings->solver == BPMPC_SOLVER_DDP) {
      // what the slice implements is what the reference configures (task.info:115-156); every other value of the ddp block is refused, not ignored
      const DdpConfig& d = s->rm.ddp;
      if (d.algorithm != 1) { set_last_error("DDP: only ddp.algorithm ILQR is implemented (SLQ integrates continuous-time Riccati equations)"); return BPMPC_ERR_UNSUPPORTED; }
      if (d.strategy != 0) { set_last_error("DDP: only ddp.strategy LINE_SEARCH is implemented"); return BPMPC_ERR_UNSUPPORTED; }
      if (d.ls_hessian_correction_strategy != 0) { set_last_error("DDP: only lineSearch.hessianCorrectionStrategy DIAGONAL_SHIFT is implemented"); return BPMPC_ERR_UNSUPPORTED; }
      if (d.max_num_iterations != 1 || settings->sqp_iterations > 1) { set_last_error("DDP: one ILQR iteration per run (ddp.maxNumIterations 1): later iterations live on the roll-out's adaptive time grid"); return BPMPC_ERR_UNSUPPORTED; }
      if (!(d.ls_min_step_length > 0.0) || !(d.ls_max_step_length >= d.ls_min_step_length)) { set_last_error("DDP: lineSearch.minStepLength / maxStepLength"); return BPMPC_ERR_INVALID_ARGUMENT; }
      if (d.use_feedback_policy && settings->feedback_policy != 2) { set_last_error("DDP: ddp.useFeedbackPolicy true is not implemented (the gains live on the nominal grid, the solution on the roll-out's time points)"); return BPMPC_ERR_UNSUPPORTED; }
      if (settings->feedback_policy == 1) { set_last_error("DDP: feedback_policy 1 (LinearController) is not implemented for the DDP solution"); return BPMPC_ERR_UNSUPPORTED; }
      // round 6: the backward pass runs on the fast kernels (reference_kernels = 1 keeps the lane-emulated bodies: cross-check); the ILQR form of the
      // fast lineariser exists for serial-leg robots (every robot of the reference), any other tree runs the reference bodies
      s->settings.return_gains = 1;
      s->settings.pipeline_chunks = 1;
    }
    s->nx = s->rm.nx; s->nu = s->rm.nu;
    HIP_CHECK(hipSetDevice(settings->device));
    { hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, settings->device)); s->num_cus = prop.multiProcessorCount; }
    { const char* e = std::getenv("BPMPC_WT_JOINT_ROWS"); s->wt_joint_rows = e && e[0] == '1'; }
    { const char* e = std::getenv("BPMPC_LIN_TABLES"); s->force_tables = e && e[0] == '1'; }
    { const char* e = std::getenv("BPMPC_LIN_COMPACT"); s->lin_compact = !(e && e[0] == '0'); }
    { const char* e = std::getenv("BPMPC_RICCATI_WAVE"); s->riccati_wave = e ? std::atoi(e) : 1; }
    { const char* e = std::getenv("BPMPC_R8_ROUNDS"); s->r8_rounds = e ? std::atoi(e) : 0; }
    {
      bool block_diagonal = true;
      for (int c = 0; c < 12; ++c)
        for (int j = 12; j < s->nu; ++j) block_diagonal = block_diagonal && s->dm.R[c * s->nu + j] == 0.0 && s->dm.R[j * s->nu + c] == 0.0;
      const char* e = std::getenv("BPMPC_DENSE_PROJECT");
      s->structured_project = block_diagonal && !(e && e[0] == '1');
    }
    if (s->is_ddp() && !(s->dm.serial_legs && !s->force_tables)) s->settings.reference_kernels = 1;
    if (settings->stream) { s->stream = static_cast<hipStream_t>(settings->stream); }
    else { HIP_CHECK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)); s->own_stream = true; }
    if (s->settings.pipeline_chunks <= 0) s->settings.pipeline_chunks = 1;
    if (s->settings.pipeline_chunks > 16) s->settings.pipeline_chunks = 16;
    HIP_CHECK(hipStreamCreateWithFlags(&s->producer_stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&s->ev_go, hipEventDisableTiming));
    s->ev_chunk.resize(s->settings.pipeline_chunks);
    for (auto& e : s->ev_chunk) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    {   // the model and, behind it, the image of the lane-per-coordinate kernels' LDS model block (linearize_fast.h: kLinImageOffset, fill_shared_image)
      std::vector<char> host(kLinImageOffset + sizeof(LinFastShared<12, true>), 0);
      std::memcpy(host.data(), &s->dm, sizeof(DeviceModel));
      if (s->rm.nj == 10) fill_shared_image<10>(s->dm, *reinterpret_cast<LinFastShared<10, true>*>(host.data() + kLinImageOffset));
      else if (s->rm.nj == 12) fill_shared_image<12>(s->dm, *reinterpret_cast<LinFastShared<12, true>*>(host.data() + kLinImageOffset));
      HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->d_model), host.size()));
      HIP_CHECK(hipMemcpy(s->d_model, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s->h_remaining), sizeof(int)));
    const SqpConfig& q = s->rm.sqp;
    s->ls = LineSearchSettings{q.g_max, q.g_min, q.alpha_decay, q.alpha_min, q.gamma_c, q.armijo_factor, q.delta_tol, q.cost_tol,
                               settings->sqp_iterations > 0 ? settings->sqp_iterations : q.sqp_iteration};
    allocate(s.get());
    HIP_CHECK(hipStreamSynchronize(s->stream));
  } catch (const std::exception& e) {
    const int rc = translate(e);
    bpmpc_solver_destroy(s.release());
    return rc;
  }
  *out = s.release();
  return BPMPC_OK;
}

void bpmpc_solver_destroy(bpmpc_solver* s) {
  if (!s) return;
  // teardown: nothing useful can be done with an error here
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->producer_stream) { (void)hipStreamSynchronize(s->producer_stream); (void)hipStreamDestroy(s->producer_stream); }
  if (s->ev_go) (void)hipEventDestroy(s->ev_go);
  for (hipEvent_t e : s->ev_chunk) if (e) (void)hipEventDestroy(e);
  for (void* p : s->allocations) (void)hipFree(p);
  if (s->d_model) (void)hipFree(s->d_model);
  if (s->h_remaining) (void)hipHostFree(s->h_remaining);
  s->pin_up.release(); s->pin_down.release();
  if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

#define API_GUARD(solver, ...)                                                                    \
  if (!(solver)) { set_last_error("null solver handle"); return BPMPC_ERR_INVALID_ARGUMENT; }      \
  try { HIP_CHECK(hipSetDevice((solver)->settings.device)); __VA_ARGS__; }                               \
  catch (const std::exception& e) {                                                               \
    /* a call that threw between enqueueing copies from / to the pinned arenas and its own synchronisation: wait for them before the */ \
    /* next call recycles (or frees) that memory */                                               \
    if ((solver)->stream) (void)hipStreamSynchronize((solver)->stream);                           \
    if ((solver)->producer_stream) (void)hipStreamSynchronize((solver)->producer_stream);         \
    return translate(e);                                                                          \
  }                                                                                               \
  return BPMPC_OK;

int bpmpc_solver_setup(bpmpc_solver* s, int batch, double horizon, const double* t0, const double* x0, const bpmpc_mode_schedule* schedules,
                       int n_schedules, const bpmpc_target* targets, const double* warm_x, const double* warm_u) {
  API_GUARD(s, setup(s, batch, horizon, t0, x0, schedules, n_schedules, targets, warm_x, warm_u))
}
int bpmpc_solver_setup_from_previous(bpmpc_solver* s, int batch, double horizon, const double* t0, const double* x0,
                                     const bpmpc_mode_schedule* schedules, int n_schedules, const bpmpc_target* targets) {
  API_GUARD(s, setup(s, batch, horizon, t0, x0, schedules, n_schedules, targets, nullptr, nullptr, true))
}
int bpmpc_solver_reset(bpmpc_solver* s) { API_GUARD(s, reset(s)) }
int bpmpc_solver_run(bpmpc_solver* s) {
  API_GUARD(s, { if (s->batch < 1) throw std::invalid_argument("bpmpc_solver_run before bpmpc_solver_setup"); s->run_iterations(); })
}
int bpmpc_solver_sync(bpmpc_solver* s) { API_GUARD(s, { HIP_CHECK(hipStreamSynchronize(s->stream)); s->collect_timers(); }) }
int bpmpc_solver_fetch(bpmpc_solver* s, double* out_t, double* out_x, double* out_u, double* out_K, bpmpc_stats* stats) {
  API_GUARD(s, fetch(s, out_t, out_x, out_u, out_K, stats))
}
int bpmpc_solver_setup_commands(bpmpc_solver* s, int batch, double horizon, const double* t0, const double* x0, const bpmpc_gait_template* gaits,
                                int n_gaits, const int* gait_of_problem, const double* gait_start, const double* cmd_vel, int command_kind,
                                double time_to_target, int from_previous) {
  API_GUARD(s, { setup_commands(s, batch, horizon, t0, x0, gaits, n_gaits, gait_of_problem, gait_start, cmd_vel, command_kind, time_to_target, from_previous != 0); })
}
int bpmpc_solver_rollout(bpmpc_solver* s, const double* t_start, const double* x_start, double duration, double* x_end, double* u_end, int* steps) {
  API_GUARD(s, { rollout(s, t_start, x_start, duration, x_end, u_end, steps); })
}
int bpmpc_solver_set_materialize(bpmpc_solver* s, int materialize_lq) {
  if (!s) { set_last_error("bpmpc_solver_set_materialize: null solver handle"); return BPMPC_ERR_INVALID_ARGUMENT; }
  s->settings.materialize_lq = materialize_lq != 0;
  return BPMPC_OK;
}
int bpmpc_solver_set_profile(bpmpc_solver* s, int level) {
  if (!s || level < 0 || level > 2) { set_last_error("bpmpc_solver_set_profile: bad argument"); return BPMPC_ERR_INVALID_ARGUMENT; }
  s->settings.profile = level;
  return BPMPC_OK;
}
int bpmpc_solve_batch(bpmpc_solver* s, int batch, double horizon, const double* t0, const double* x0, const bpmpc_mode_schedule* schedules,
                      int n_schedules, const bpmpc_target* targets, const double* warm_x, const double* warm_u, double* out_t, double* out_x,
                      double* out_u, double* out_K, bpmpc_stats* stats) {
  API_GUARD(s, {
    setup(s, batch, horizon, t0, x0, schedules, n_schedules, targets, warm_x, warm_u);
    s->run_iterations();
    fetch(s, out_t, out_x, out_u, out_K, stats);
    s->collect_timers();
  })
}
int bpmpc_solver_stage(bpmpc_solver* s, const char* stage) {
  API_GUARD(s, {
    if (!stage || s->batch < 1) throw std::invalid_argument("bpmpc_solver_stage: no stage name or no setup");
    const std::string n(stage);
    // an explicit stage request always applies to every problem of the batch
    HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)s->buf.active, 1, s->batch, s->stream));
    if (n == "linearize") s->stage_linearize();
    else if (n == "project") s->stage_project();
    else if (n == "riccati") s->stage_riccati();
    else if (n == "linesearch") s->stage_linesearch();
    else throw std::invalid_argument("unknown stage " + n);
  })
}
int bpmpc_solver_read(bpmpc_solver* s, const char* name, double* out, long capacity) {
  if (!s || !name) { set_last_error("bpmpc_solver_read: null argument"); return BPMPC_ERR_INVALID_ARGUMENT; }
  try {
    if (!out) {                                             // size query
#if defined(BPMPC_EVAL_PROFILE)
      if (std::string(name) == "evprof") return 16;
#endif
      auto q = s->named.find(name);
      if (q == s->named.end()) throw std::invalid_argument(std::string("unknown buffer ") + name);
      if (q->second.second > 0x7fffffffu) throw std::length_error("bpmpc_solver_read: buffer exceeds 2^31 elements");
      return (int)q->second.second;
    }
    HIP_CHECK(hipSetDevice(s->settings.device));
#if defined(BPMPC_EVAL_PROFILE)
    if (std::string(name) == "evprof") {
      long long tmp[16];
      HIP_CHECK(hipStreamSynchronize(s->stream));
      HIP_CHECK(hipMemcpyFromSymbol(tmp, HIP_SYMBOL(g_evprof), sizeof(tmp)));
      for (int i = 0; i < 16; ++i) out[i] = (double)tmp[i];
      long long zero[16] = {0};
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_evprof), zero, sizeof(zero)));
      return 16;
    }
#endif
    auto it = s->named.find(name);
    if (it == s->named.end()) throw std::invalid_argument(std::string("unknown buffer ") + name);
    const size_t n = it->second.second;
    if ((long)n > capacity) throw std::length_error("bpmpc_solver_read: capacity too small");
    if (n > 0x7fffffffu) throw std::length_error("bpmpc_solver_read: buffer exceeds 2^31 elements");
    HIP_CHECK(hipStreamSynchronize(s->stream));
    if (s->is_int[name]) {
      std::vector<int> tmp(n);
      HIP_CHECK(hipMemcpy(tmp.data(), it->second.first, n * sizeof(int), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) out[i] = tmp[i];
    } else {
      HIP_CHECK(hipMemcpy(out, it->second.first, n * sizeof(double), hipMemcpyDeviceToHost));
    }
    return (int)n;
  } catch (const std::exception& e) { return translate(e); }
}
int bpmpc_solver_device_trajectories(bpmpc_solver* s, double** x_dev, double** u_dev) {
  if (!s) { set_last_error("null solver handle"); return BPMPC_ERR_INVALID_ARGUMENT; }
  if (x_dev) *x_dev = s->buf.x;
  if (u_dev) *u_dev = s->buf.u;
  return BPMPC_OK;
}
int bpmpc_solver_export_trajectories(bpmpc_solver* s, double* x_dst_dev, double* u_dst_dev) {
  API_GUARD(s, {
    const size_t N = s->settings.max_nodes;
    if (x_dst_dev && u_dst_dev && ((uintptr_t)x_dst_dev % 16 == 0) && ((uintptr_t)u_dst_dev % 16 == 0)) {
      copy_pairs(s, s->buf.x, x_dst_dev, (size_t)s->batch * (N + 1) * s->nx, s->buf.u, u_dst_dev, (size_t)s->batch * N * s->nu, false);
    } else {
      if (x_dst_dev) HIP_CHECK(hipMemcpyAsync(x_dst_dev, s->buf.x, (size_t)s->batch * (N + 1) * s->nx * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
      if (u_dst_dev) HIP_CHECK(hipMemcpyAsync(u_dst_dev, s->buf.u, (size_t)s->batch * N * s->nu * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    }
  })
}
int bpmpc_solver_constraint_values(bpmpc_solver* s, double* values, int* rows, int* modes) {
  API_GUARD(s, {
    if (!values || s->batch < 1) throw std::invalid_argument("bpmpc_solver_constraint_values: null output or no setup");
    if (s->settings.reference_kernels) throw std::invalid_argument("bpmpc_solver_constraint_values needs the fast kernels");
    if (s->is_ddp() && s->has_solution) throw Unsupported("bpmpc_solver_constraint_values: the DDP solution lives on the time points of its roll-out, not on the shooting grid the constraint rows are evaluated on");
    const size_t N = s->settings.max_nodes, S = (size_t)s->batch * N;
    double* d_eqv = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_eqv), S * kMaxEqRows * sizeof(double)));     // a debugging path (solver observers): allocated per call
    try {
      HIP_CHECK(hipMemsetAsync(d_eqv, 0, S * kMaxEqRows * sizeof(double), s->stream));
      const Launch L = s->launch_params();
      kl::constraint_values(s->rm.nj, s->batch * L.klen, s->stream, L, d_eqv);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(values, d_eqv, S * kMaxEqRows * sizeof(double), hipMemcpyDeviceToHost, s->stream));
      HIP_CHECK(hipStreamSynchronize(s->stream));
    } catch (...) { (void)hipFree(d_eqv); throw; }
    (void)hipFree(d_eqv);
    // rows and contact mode per node from the host copies of the grids: an intermediate node has 12 + (swing contacts) rows, an event node none
    std::vector<int> mode_h;
    if (rows || modes) {
      mode_h.resize((size_t)s->n_grids * N);
      HIP_CHECK(hipMemcpy(mode_h.data(), s->buf.g_mode, mode_h.size() * sizeof(int), hipMemcpyDeviceToHost));
      for (int b = 0; b < s->batch; ++b) {
        const int g = s->grid_of_problem[b];
        for (size_t k = 0; k < N; ++k) {
          const bool live = (int)k < s->grid_nodes[g] && s->grid_kind[(size_t)g * N + k] == 0;
          const int m = mode_h[(size_t)g * N + k] & 3;
          const int swing = (m & 1 ? 0 : 2) + (m & 2 ? 0 : 2);            // mode bit 0: left foot (contacts 0, 1) in stance, bit 1: right foot
          if (rows) rows[(size_t)b * N + k] = live ? 12 + swing : 0;
          if (modes) modes[(size_t)b * N + k] = live ? m : -1;
        }
      }
    }
  })
}
int bpmpc_solver_kernel_time(bpmpc_solver* s, const char* kernel, int reset_after, double* total_ms, int* launches) {
  API_GUARD(s, {
    if (!kernel) throw std::invalid_argument("null kernel class");
    HIP_CHECK(hipStreamSynchronize(s->stream));
    s->collect_timers();
    KernelTimer& t = s->timers[kernel];
    if (total_ms) *total_ms = t.total_ms;
    if (launches) *launches = t.launches;
    if (reset_after) { t.total_ms = 0.0; t.launches = 0; }
  })
}
int bpmpc_solver_layout(const bpmpc_solver* s, int* batch, int* n_nodes_max, int* n_grids, int* nx, int* nu) {
  if (!s) { set_last_error("null solver handle"); return BPMPC_ERR_INVALID_ARGUMENT; }
  if (batch) *batch = s->batch;
  if (n_nodes_max) *n_nodes_max = s->n_nodes_max;
  if (n_grids) *n_grids = s->n_grids;
  if (nx) *nx = s->nx;
  if (nu) *nu = s->nu;
  return BPMPC_OK;
}

}  // extern "C"
