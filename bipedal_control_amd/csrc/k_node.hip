// Per-node kernels in the lane-per-coordinate mapping (kernels/linearize_fast.h), line search, warm start, policy rollout.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <stdexcept>

#include "kernel_launchers.h"
#include "launch.h"
#include "kernels/rollout.h"

namespace bpmpc {

template <int NJ>
__global__ __launch_bounds__(kWave) void k_prepare(Launch L) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  const int s = blockIdx.x, b = s / L.N, k = s % L.N;
  const int g = L.buf.p_grid[b];
  const int n = L.buf.g_nodes[g];
  if (k >= n) { if (threadIdx.x == 0) L.buf.n_info[s] = 0; return; }
  const size_t gs = (size_t)g * L.N + k;
  if (threadIdx.x == 0) L.buf.n_info[s] = 8 | ((L.buf.g_mode[gs] & 3) << 1) | (L.buf.g_kind[gs] == 1 ? 1 : 0);
  if (threadIdx.x < kNodeAux) {       // the node record: dt, zref[4], zdref[4] (launch.h Buffers::n_aux)
    const int t = threadIdx.x;
    L.buf.n_aux[(size_t)s * kNodeAux + t] = t == 0 ? L.buf.g_dt[gs] : t < 5 ? L.buf.g_zref[gs * 4 + t - 1] : t < 9 ? L.buf.g_zdref[gs * 4 + t - 5] : 0.0;
  }
  prepare_node<NJ>(*L.model, L.buf.g_kind[gs], L.buf.g_mode[gs], L.buf.g_start[gs], L.cold != 0, k == n - 1, L.buf.p_tgt_n[b],
                   L.buf.p_tgt_t + (size_t)b * kMaxTargetPoints, L.buf.p_tgt_x + (size_t)b * kMaxTargetPoints * NX, L.buf.p_x0 + (size_t)b * NX,
                   L.buf.xref + (size_t)s * NX, L.buf.x + ((size_t)b * (L.N + 1) + k) * NX, L.buf.u + (size_t)s * NU,
                   L.buf.x + ((size_t)b * (L.N + 1) + k + 1) * NX);
}

template <int NJ>
__global__ __launch_bounds__(kWave) void k_linearize(Launch L) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  __shared__ NodeWorkspace<NJ> ws;
  const int sidx = blockIdx.x, b = sidx / L.N, k = sidx % L.N;
  if (!L.buf.active[b]) return;
  if (k >= L.buf.g_nodes[L.buf.p_grid[b]]) return;
  const size_t s = sidx;
  const NodeInputs in = node_inputs<NJ>(L, b, k);
  NodeLQOut out;
  out.A = L.buf.A + s * NX * NX; out.B = L.buf.B + s * NX * NU; out.b = L.buf.b + s * NX;
  out.Q = L.buf.Q + s * NX * NX; out.R = L.buf.R + s * NU * NU; out.P = L.buf.P + s * NU * NX;
  out.q = L.buf.q + s * NX; out.r = L.buf.r + s * NU; out.c = L.buf.c + s;
  out.C = L.buf.C + s * kMaxEqRows * NX; out.D = L.buf.D + s * kMaxEqRows * NU; out.e = L.buf.e + s * kMaxEqRows;
  out.nc = L.buf.nc + s; out.perf = L.buf.perf + s * 3;
  out.prof = (b == 0 && k < 64) ? L.buf.rprof + 8 * k : nullptr;
  linearize_node<NJ>(*L.model, ws, in, out, L.ilqr != 0, L.ilqr_shift);
}

constexpr int kTrialWaves = 4; // same for the value-only trial kernel (smaller per-node LDS: four waves, 8 waves per CU)
// wavefronts per workgroup of the linearisation kernel: they share one copy of the model block in LDS.  Four: 74.3 KB at nx = 22, 79.1 KB
// at nx = 24 (a wave serves four nodes of 4.2 KB each, packed lanes, LinFastCfg) - two workgroups, eight waves per CU
template <int NJ> constexpr int lin_waves() { return 4; }      // (five - ten waves per CU - was measured: 0.302 against 0.223 ms at batch 256; the kernel is not short of waves, experiments/LOG.md)
template <int NJ, bool MAT, bool CHAIN, bool ILQR = false>      // ILQR: the DDP solver's Euler-discretised model (linearize_fast.h)
__global__ __launch_bounds__(lin_waves<NJ>() * kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_linearize_fast(Launch L) {
  using C = LinFastCfg<NJ, true, CHAIN>;
  constexpr int LPN = C::LPN, NPW = C::NPW, kLinWaves = lin_waves<NJ>();
  // the stage-one columns wait in LDS when two workgroups per CU (eight waves: the registers allow no more) still fit, else in HBM scratch
  constexpr bool PARK = sizeof(LinFastNodeLds<NJ, true, CHAIN, true>) * (kLinWaves * NPW) + sizeof(LinFastShared<NJ>) <= 80 * 1024;
  using NL = LinFastNodeLds<NJ, true, CHAIN, PARK>;
  __shared__ NL lds[kLinWaves * NPW];
  __shared__ LinFastShared<NJ> shared;     // model constants indexed per lane, shared by the nodes of the workgroup
#ifdef BPMPC_LIN_TIMELINE                  // per workgroup: start, model staged, end (10 ns ticks) and the hardware id -> tools/lin_timeline.py
  const long long tl0 = wall_clock64();
#endif
  // The memory round trips of a workgroup's start overlap: (1) the problem's flags and grid, the node's word (valid | mode | kind, written
  // by k_prepare) and the lane's entries of the iterate - addresses that only depend on the slot, loaded before anything is known about
  // the node - are in flight while (2) the model block is staged; (3) what hangs on the grid (dt, swing references) follows.
  const int sub = threadIdx.x / LPN, g = threadIdx.x % LPN;       // sub: node slot of the workgroup
  const int widx = blockIdx.x * (kLinWaves * NPW) + sub;          // batch * max_nodes < 2^31 is checked at creation
  bool valid, event_wg = false;
  int b, k;
  if (L.lin_ev_n >= 0) {            // one grid for the batch: intermediate nodes first, the event nodes on workgroups of their own behind them (Launch::lin_ev)
    constexpr int per_wg = kLinWaves * NPW;
    const int n_comp = L.batch * L.lin_inter, wg_comp = (n_comp + per_wg - 1) / per_wg;
    if ((int)blockIdx.x < wg_comp) {
      valid = widx < n_comp;
      b = valid ? widx / L.lin_inter : 0;
      const int j = valid ? widx % L.lin_inter : 0;
      int kk = j;                   // the j-th intermediate node: every event i with ev_i - i <= j lies before it
#pragma unroll
      for (int i = 0; i < kLinMaxEvents; ++i) kk += (i < L.lin_ev_n && L.lin_ev[i] - i <= j) ? 1 : 0;
      k = kk;
    } else {
      event_wg = true;              // nothing but event nodes: no evaluation, no model block
      const int e = widx - wg_comp * per_wg;
      valid = e < L.batch * L.lin_ev_n;
      b = valid ? e / L.lin_ev_n : 0;
      const int i = valid ? e % L.lin_ev_n : 0;
      int kk = L.lin_ev[0];
#pragma unroll
      for (int q = 1; q < kLinMaxEvents; ++q) kk = (i == q) ? L.lin_ev[q] : kk;
      k = kk;
    }
  } else {
    valid = widx < L.batch * L.klen;
    b = valid ? widx / L.klen : 0; k = valid ? L.k0 + widx % L.klen : 0;
  }
  const int act = L.buf.active[b];
  const size_t s0 = (size_t)b * L.N + k;
  const int info = L.buf.n_info[s0];
  constexpr int NXc = 12 + NJ;
  const double* xk = L.buf.x + ((size_t)b * (L.N + 1) + k) * NXc;
  // round 6: dt and the swing references come from the node record (n_aux, written by k_prepare), not from the grid tables behind p_grid[b]:
  // nothing a workgroup waits for before its barrier is more than one memory round trip away
  const LinFastPre pre = linearize_preload<C>(xk, xk + NXc, L.buf.u + s0 * NXc, L.buf.xref + s0 * NXc, g, L.buf.n_aux + s0 * kNodeAux);
  if (!event_wg) load_shared_model<kLinWaves * kWave>(*L.model, shared, threadIdx.x);
  NodeInputs in;
  in.kind = info & 1; in.mode = (info >> 1) & 3;                  // (the same facts as the grid tables hold, one hop earlier)
  in.dt = 0.0;                                                    // (the node record carries it: LinFastPre::aux)
  in.x = xk; in.xnext = xk + NXc; in.u = L.buf.u + s0 * NXc; in.xref = L.buf.xref + s0 * NXc; in.zref = nullptr; in.zdref = nullptr;
  __syncthreads();
#ifdef BPMPC_LIN_TIMELINE
  const long long tl1 = wall_clock64();
#endif
  valid = valid && act != 0 && info != 0;
  const size_t s = valid ? (size_t)b * L.N + k : 0;
  LinFastOut out;
  out.A = L.buf.A; out.B = L.buf.B; out.b = L.buf.b; out.Q = L.buf.Q; out.R = L.buf.R; out.q = L.buf.q; out.r = L.buf.r; out.c = L.buf.c;
  out.C = L.buf.C; out.D = L.buf.D; out.e = L.buf.e; out.perf = L.buf.perf; out.nc = L.buf.nc;
  out.park = L.buf.lin_park;
  out.dump = L.buf.lin_dump;
  out.qrd = L.buf.qrd;
  out.s = s;
  out.ilqr_shift = L.ilqr_shift;
#ifndef BPMPC_LIN_PROF_PROBLEM
#define BPMPC_LIN_PROF_PROBLEM 0     // the problem whose first 64 nodes report their phase cycles (-DBPMPC_LINFAST_PROFILE): 0 runs on an empty chip, batch / 2 in steady state
#endif
  out.prof = (valid && b == BPMPC_LIN_PROF_PROBLEM && k < 64) ? L.buf.rprof + 8 * k : nullptr;
  linearize_fast<NJ, MAT, C, NL, ILQR>(*L.model, shared, lds[sub], valid, in, pre, out, g);      // g: lane inside the node's group
#ifdef BPMPC_LIN_TIMELINE
  if (threadIdx.x % kWave == 0 && blockIdx.x < 2048) {      // every wave: the workgroup's end is the latest of its waves
    double* t = L.buf.rprof + 16 * blockIdx.x + 4 * (threadIdx.x / kWave);
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    t[0] = (double)tl0; t[1] = (double)tl1; t[2] = (double)wall_clock64(); t[3] = (double)(hw | ((unsigned long long)(xcc & 15) << 32));
  }
#endif
}

// Warm start of a receding-horizon solve from the previous solution, one wavefront per (problem, node).  [OCS2-upstream, recalled]
// SqpSolver::initializeStateInputTrajectories with a non-empty PrimalSolution: for an intermediate node with
// intervalStart <= second-to-last and intervalEnd <= last time of the previous solution,
//     u_i = uff(t) + K(t) x_i  (LinearController, sqp.useFeedbackPolicy true, task.info:80;  uff_j = u_j - K_j x_j, inputs and
//           gains of pre-event nodes and of the terminal node repeat the previous one: multiple_shooting::toPrimalSolution),
//     x_{i+1} = LinearInterpolation(intervalEnd, previous states);
// otherwise BipedalRobotInitializer::compute (already written by k_prepare) and x_{i+1} = x_i; event nodes copy the state.
// The state guess of node i is therefore the interpolation at the end of the last interpolating node before it (or the measured
// state), which every node finds on its own - no sequential sweep (the first version, one wavefront per problem walking the
// horizon, took 0.53 ms at batch 256 and was the longest kernel of a closed-loop tick).
// Oracle: oracle/reference_py.py warm_start_from_previous.
template <int NJ>
__global__ __launch_bounds__(kWave) void k_warm_shift(Launch L) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  __shared__ double xi[NX];
  __shared__ double Ks[2][NU * NX];
  const int b = blockIdx.x / L.N, i = blockIdx.x % L.N, l = threadIdx.x;
  const int N = L.N;
  const int g = L.buf.p_grid[b], n = L.buf.g_nodes[g];
  if (i >= n) return;
  const int gp = L.buf.tp_grid[b], np = L.buf.tp_nodes[gp];
  if (np < 1) return;
  const double* tp = L.buf.tp_time + (size_t)gp * (N + 1);
  const int* kp = L.buf.tp_kind + (size_t)gp * N;
  const double* xp = L.buf.x_prev + (size_t)b * (N + 1) * NX;
  const double* up = L.buf.u_prev + (size_t)b * N * NU;
  const double* Kp = L.buf.K_prev + (size_t)b * N * NU * NX;
  double* x = L.buf.x + (size_t)b * (N + 1) * NX;
  double* u = L.buf.u + (size_t)b * N * NU;
  const double state_till = tp[np], input_till = tp[np - 1];
  auto effective = [&](int j) { while (j > 0 && (j == np || kp[j] == 1)) --j; return j; };   // repeated input / gain
  auto interpolates = [&](int k) {
    const size_t gs = (size_t)g * N + k;
    if (L.buf.g_kind[gs] != 0) return false;
    const double t = L.buf.g_start[gs], tn = t + L.buf.g_dt[gs];
    return !(t > input_till || tn > state_till);
  };
  auto state_after = [&](int k, int c) {            // component c of the guess of x_{k+1} for an interpolating node k
    const size_t gs = (size_t)g * N + k;
    int j2;
    double a2;
    time_segment(tp, np + 1, L.buf.g_start[gs] + L.buf.g_dt[gs], &j2, &a2);
    return a2 * xp[(size_t)j2 * NX + c] + (1.0 - a2) * xp[(size_t)(j2 + 1) * NX + c];
  };
  int src = i - 1;
  while (src >= 0 && !interpolates(src)) --src;
  const bool mine = interpolates(i);
  if (l < NX) {
    const double v = src < 0 ? L.buf.p_x0[(size_t)b * NX + l] : state_after(src, l);
    xi[l] = v;
    if (i == 0) x[l] = v;
    x[(size_t)(i + 1) * NX + l] = mine ? state_after(i, l) : v;
  }
  if (!mine) return;
  int j;
  double a;
  time_segment(tp, np + 1, L.buf.g_start[(size_t)g * N + i], &j, &a);
  const int e0 = effective(j), e1 = effective(j + 1);
  if (!L.feedback) {      // FeedforwardController (sqp.useFeedbackPolicy false): the interpolated input trajectory, no state feedback
    if (l < NU) u[(size_t)i * NU + l] = a * up[(size_t)e0 * NU + l] + (1.0 - a) * up[(size_t)e1 * NU + l];
    return;
  }
  for (int idx = l; idx < NU * NX; idx += kWave) {
    Ks[0][idx] = Kp[(size_t)e0 * NU * NX + idx];
    Ks[1][idx] = Kp[(size_t)e1 * NU * NX + idx];
  }
  __syncthreads();
  if (l < NU) {
    double uff0 = up[(size_t)e0 * NU + l], uff1 = up[(size_t)e1 * NU + l], kx = 0.0;
    for (int c = 0; c < NX; ++c) {
      const double k0 = Ks[0][l * NX + c], k1 = Ks[1][l * NX + c];
      uff0 -= k0 * xp[(size_t)j * NX + c];
      uff1 -= k1 * xp[(size_t)(j + 1) * NX + c];
      kx += (a * k0 + (1.0 - a) * k1) * xi[c];
    }
    u[(size_t)i * NU + l] = a * uff0 + (1.0 - a) * uff1 + kx;
  }
}

template <int NJ>
__global__ __launch_bounds__(kWave) void k_ls_begin(Launch L) {
  __shared__ double partial[3 * kWave + 5];
  linesearch_begin<NJ>(partial, problem_ls<NJ>(L, blockIdx.x));
}

template <int NJ>
__global__ __launch_bounds__(kWave) void k_trial(Launch L) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  __shared__ TrialWorkspace<NJ> ws;
  const int sidx = blockIdx.x, b = sidx / L.N, k = sidx % L.N;
  if (L.buf.done[b]) return;
  if (k >= L.buf.g_nodes[L.buf.p_grid[b]]) return;
  const NodeInputs in = node_inputs<NJ>(L, b, k);
  const double* dx = L.buf.dx + ((size_t)b * (L.N + 1) + k) * NX;
  trial_node<NJ>(*L.model, ws, in, L.buf.alpha[b], dx, L.buf.du + (size_t)sidx * NU, dx + NX, L.buf.trial_perf + (size_t)sidx * 3);
}

// Waves per SIMD of the value-only kernel: three (168 registers; the LDS fits three workgroups per CU since the node tables share one storage) -
// nx = 24: line search 0.492 -> 0.404 ms on G1 / 1024, 0.152 -> 0.129 at batch 256.  nx = 22 had a second variant at four waves per SIMD (128
// registers, 64 B of scratch) for launches of many rounds (round 3: 1.18 -> 1.13 ms at batch 4096); since the inputs are requested before
// the model block is staged (round 4) three waves win everywhere (line search 1.049 -> 0.975 ms at 4096, 0.290 -> 0.263 at 1024): removed.
template <int NJ, bool CHAIN>
__global__ __launch_bounds__(kTrialWaves * kWave) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_trial_fast(Launch L) {
  using C = LinFastCfg<NJ, true, CHAIN>;
  constexpr int NX = 12 + NJ, NU = 12 + NJ, LPN = C::LPN, NPW = C::NPW;
  __shared__ LinFastNodeLds<NJ, false, CHAIN> lds[kTrialWaves * NPW];
  __shared__ LinFastShared<NJ, false> shared;     // model constants indexed per lane, shared by the nodes of the workgroup
  const int sub = threadIdx.x / LPN, g = threadIdx.x % LPN;
  const int widx = blockIdx.x * (kTrialWaves * NPW) + sub;
  {   // a workgroup whose problems have all finished their line search has nothing to evaluate (the second round: all but a few workgroups)
    const int total = L.batch * L.klen, w0 = blockIdx.x * (kTrialWaves * NPW), w1 = (w0 + kTrialWaves * NPW < total ? w0 + kTrialWaves * NPW : total) - 1;
    bool any = false;
    for (int pb = w0 / L.klen; pb <= w1 / L.klen; ++pb) any = any || L.buf.done[pb] == 0;
    if (!any) return;
  }
  bool valid = widx < L.batch * L.klen;
  const int b = valid ? widx / L.klen : 0, k = valid ? L.k0 + widx % L.klen : 0;
  const double* dx = L.buf.dx + ((size_t)b * (L.N + 1) + k) * NX;
  {
    // the node's facts and the lane's entries of iterate and step are requested before the model block is staged, as in k_linearize_fast
    // (line search 0.099 -> 0.096 ms at batch 256, 1.049 -> 0.975 at 4096)
    const int fin = L.buf.done[b];
    const double alpha = L.buf.alpha[b];
    const int info = L.buf.n_info[(size_t)b * L.N + k];
    const size_t s0 = (size_t)b * L.N + k;
    const double* xk = L.buf.x + ((size_t)b * (L.N + 1) + k) * NX;
    const TrialPre pre = trial_preload<C>(xk, xk + NX, L.buf.u + s0 * NU, L.buf.xref + s0 * NX, dx, dx + NX, L.buf.du + s0 * NU, g, L.buf.n_aux + s0 * kNodeAux);
    load_shared_model<kTrialWaves * kWave>(*L.model, shared, threadIdx.x);
    NodeInputs in;                      // dt and the swing references come with the node record (TrialPre::x.aux)
    in.kind = info & 1; in.mode = (info >> 1) & 3; in.dt = 0.0;
    in.x = xk; in.xnext = xk + NX; in.u = L.buf.u + s0 * NU; in.xref = L.buf.xref + s0 * NX; in.zref = nullptr; in.zdref = nullptr;
    __syncthreads();
    valid = valid && fin == 0 && info != 0;
    const size_t s = valid ? s0 : 0;
    trial_fast<NJ, C>(*L.model, shared, lds[sub], valid, in, alpha, dx, L.buf.du + s * NU, dx + NX, L.buf.trial_perf + s * 3, g, nullptr, &pre);
  }
}

// Values of the active equality rows at the CURRENT iterate (after a solve: the solution), per node in registration order: the value-only
// evaluation of the line search with a zero step and one more output (linearize_fast.h trial_fast<.., EQV>).
template <int NJ, bool CHAIN>
__global__ __launch_bounds__(kTrialWaves * kWave) void k_constraint_values(Launch L, double* eqv) {
  using C = LinFastCfg<NJ, true, CHAIN>;
  constexpr int NX = 12 + NJ, NU = 12 + NJ, LPN = C::LPN, NPW = C::NPW;
  __shared__ LinFastNodeLds<NJ, false, CHAIN> lds[kTrialWaves * NPW];
  __shared__ LinFastShared<NJ, false> shared;
  load_shared_model<kTrialWaves * kWave>(*L.model, shared, threadIdx.x);
  __syncthreads();
  const int sub = threadIdx.x / LPN, g = threadIdx.x % LPN;
  const int widx = blockIdx.x * (kTrialWaves * NPW) + sub;
  bool valid = widx < L.batch * L.klen;
  const int b = valid ? widx / L.klen : 0, k = valid ? L.k0 + widx % L.klen : 0;
  const int grid = L.buf.p_grid[b];
  valid = valid && k < L.buf.g_nodes[grid] && L.buf.g_kind[(size_t)grid * L.N + k] == 0;
  const size_t s = valid ? (size_t)b * L.N + k : 0;
  const NodeInputs in = node_inputs<NJ>(L, b, k);
  const double* dx = L.buf.dx + ((size_t)b * (L.N + 1) + k) * NX;
  double perf[3];
  trial_fast<NJ, C, true>(*L.model, shared, lds[sub], valid, in, 0.0, dx, L.buf.du + s * NU, dx + NX, perf, g, eqv + s * kMaxEqRows);
}

template <int NJ, bool CHAIN>
__global__ __launch_bounds__(kWave) void k_rollout(const DeviceModel* model, RolloutArgs a) {
  __shared__ RolloutLds<NJ, CHAIN> w;
  load_shared_model<kWave>(*model, w.shared, threadIdx.x);
  __syncthreads();
  rollout_policy<NJ, CHAIN>(*model, w, a);
}

constexpr int kDecideThreads = 256;
template <int NJ>
__global__ __launch_bounds__(kDecideThreads) void k_ls_decide(Launch L, int look_first) {
  __shared__ double partial[3 * kDecideThreads + 5];
  linesearch_decide<NJ, kDecideThreads>(partial, problem_ls<NJ>(L, blockIdx.x), L.ls, look_first != 0);
}

// Back-tracking rounds after the first one, entirely on the device: a workgroup per problem that has not accepted yet (normally
// none: the block exits at once) evaluates its own trial nodes, thirty-two at a time, and decides, until the problem accepts or gives
// up.  The host never reads a flag back inside a solve, so consecutive solves queue without a gap.  The few problems that back-track
// are alone on their CUs and what they cost is the latency of their rounds: eight waves walk the horizon in half the chunks of four
// (the sums are the ones of k_ls_decide term by term while the horizon has at most kDecideThreads nodes).
constexpr int kTailThreads = 512;
template <int NJ, bool CHAIN>
__global__ __launch_bounds__(kTailThreads) void k_ls_tail(Launch L, int first_round, int max_trials, int pending) {
  using C = LinFastCfg<NJ, true, CHAIN>;
  constexpr int NX = 12 + NJ, NU = 12 + NJ, LPN = C::LPN, NPW = C::NPW, CHUNK = (kTailThreads / kWave) * NPW;
  const int b = blockIdx.x;
  if (L.buf.done[b]) return;
  __shared__ LinFastNodeLds<NJ, false, CHAIN> lds[CHUNK];
  __shared__ LinFastShared<NJ, false> shared;
  __shared__ double partial[3 * kTailThreads + 5];
  volatile const int* done = L.buf.done + b;
  volatile const double* alpha = L.buf.alpha + b;
  if (pending) {
    // the trial of the last round that ran as a launch over all nodes has not been judged yet: that decision here, in the workgroup that would go on
    // with the problem anyway, instead of a launch of k_ls_decide in which all but a few workgroups only read a flag
    int bb = b;
    asm volatile("" : "+s"(bb));
    linesearch_decide<NJ, kTailThreads, true>(partial, problem_ls<NJ>(L, bb), L.ls);
    __threadfence();
    __syncthreads();
    if (*done) return;
  }
  load_shared_model<kTailThreads>(*L.model, shared, threadIdx.x);
  __syncthreads();
  const int sub = threadIdx.x / LPN, g = threadIdx.x % LPN;
  const int n = L.buf.g_nodes[L.buf.p_grid[b]];
  for (int round = first_round; round < max_trials; ++round) {
    const double al = *alpha;
    for (int k0 = 0; k0 < n; k0 += CHUNK) {
      const int k = k0 + sub;
      const bool valid = k < n;
      const int kk = valid ? k : 0;
      const size_t s = (size_t)b * L.N + kk;
      const NodeInputs in = node_inputs<NJ>(L, b, kk);
      const double* dx = L.buf.dx + ((size_t)b * (L.N + 1) + kk) * NX;
      // the model block through an offset the compiler cannot see through: its scalar constants (LinFastScalars) are loop invariant, and hoisted
      // out of the round loop they lived in registers across the whole evaluation (256 registers + 20 .. 28 B of scratch)
      unsigned opaque = 0;
      asm volatile("" : "+v"(opaque));
      const auto& sh = *reinterpret_cast<const LinFastShared<NJ, false>*>(reinterpret_cast<const char*>(&shared) + opaque);
      trial_fast<NJ, C>(*L.model, sh, lds[sub], valid, in, al, dx, L.buf.du + s * NU, dx + NX, L.buf.trial_perf + s * 3, g);
    }
    __threadfence();
    __syncthreads();
    // the problem's pointers are formed HERE, from an index the compiler cannot see through: hoisted out of the round loop they lived across the trial
    // evaluation, and at nx = 24 (256 registers) three of them went to scratch memory
    int bb = b;
    asm volatile("" : "+s"(bb));
    const ProblemLS p = problem_ls<NJ>(L, bb);
    linesearch_decide<NJ, kTailThreads, false>(partial, p, L.ls);
    __threadfence();
    __syncthreads();
    if (*done) break;
  }
}

// One launch instead of several driver copies / fills (each costs a dispatch gap of a few microseconds between kernels):
// copies two pairs of double arrays and optionally re-arms the per-problem flags.
__global__ __launch_bounds__(256) void k_copy_pairs(const double* a_src, double* a_dst, size_t na, const double* b_src, double* b_dst, size_t nb,
                                                     int* iterations, int* active, int batch) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  const double2* a2 = reinterpret_cast<const double2*>(a_src);
  const double2* b2 = reinterpret_cast<const double2*>(b_src);
  double2* ad = reinterpret_cast<double2*>(a_dst);
  double2* bd = reinterpret_cast<double2*>(b_dst);
  for (size_t i = i0; i < na / 2; i += stride) ad[i] = a2[i];
  for (size_t i = i0; i < nb / 2; i += stride) bd[i] = b2[i];
  if (i0 == 0) { if (na & 1) a_dst[na - 1] = a_src[na - 1]; if (nb & 1) b_dst[nb - 1] = b_src[nb - 1]; }
  if (iterations) for (size_t i = i0; i < (size_t)batch; i += stride) { iterations[i] = 0; active[i] = 1; }
}

#define KL_NJ(nj, ...)                                                          \
  do {                                                                          \
    if ((nj) == 10) { constexpr int NJ = 10; __VA_ARGS__; }                     \
    else if ((nj) == 12) { constexpr int NJ = 12; __VA_ARGS__; }                \
    else throw std::runtime_error("unsupported joint count");                   \
  } while (0)

namespace kl {

void prepare(int nj, int slots, hipStream_t st, const Launch& L) { KL_NJ(nj, hipLaunchKernelGGL(k_prepare<NJ>, dim3(slots), dim3(kWave), 0, st, L)); }
void linearize_reference(int nj, int slots, hipStream_t st, const Launch& L) { KL_NJ(nj, hipLaunchKernelGGL(k_linearize<NJ>, dim3(slots), dim3(kWave), 0, st, L)); }
// ev_start / ev_stop (both or neither): HIP events attached to the kernel's own dispatch (hipExtLaunchKernelGGL) - their elapsed time is the
// kernel's duration, what the roofline of bench.py is defined on; a pair of hipEventRecord calls around the launch also brackets two barrier
// packets and the dispatch latency (6 .. 12 us of a 175 us kernel, depending on the box)
void linearize_fast(int nj, bool materialise, int nodes, hipStream_t st, const Launch& L, hipEvent_t ev_start, hipEvent_t ev_stop) {
  KL_NJ(nj, {
    constexpr int per_wg = lin_waves<NJ>() * LinFastCfg<NJ, true>::NPW;
    const int wgs = L.lin_ev_n >= 0 ? (L.batch * L.lin_inter + per_wg - 1) / per_wg + (L.batch * L.lin_ev_n + per_wg - 1) / per_wg : (nodes + per_wg - 1) / per_wg;
    const dim3 grid(wgs), block(lin_waves<NJ>() * kWave);
    auto launch = [&](auto kernel) {
      if (ev_start && ev_stop) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, ev_start, ev_stop, 0, L);
      else hipLaunchKernelGGL(kernel, grid, block, 0, st, L);
    };
    if (L.ilqr) {                   // the DDP solver (solver.hip run_ddp)
      if (!L.serial_legs) throw std::runtime_error("linearize_fast: the ILQR lineariser is instantiated for serial-leg robots only");
      if (materialise) launch(k_linearize_fast<NJ, true, true, true>);
      else launch(k_linearize_fast<NJ, false, true, true>);
    } else if (L.serial_legs) {     // two serial legs in lane order (DeviceModel::serial_legs): tree walks by DPP row shifts
      if (materialise) launch(k_linearize_fast<NJ, true, true>);
      else launch(k_linearize_fast<NJ, false, true>);
    } else {                        // any tree: walks over LDS tables
      if (materialise) launch(k_linearize_fast<NJ, true, false>);
      else launch(k_linearize_fast<NJ, false, false>);
    }
  });
}
void warm_shift(int nj, int slots, hipStream_t st, const Launch& L) { KL_NJ(nj, hipLaunchKernelGGL(k_warm_shift<NJ>, dim3(slots), dim3(kWave), 0, st, L)); }
void ls_begin(int nj, int batch, hipStream_t st, const Launch& L) { KL_NJ(nj, hipLaunchKernelGGL(k_ls_begin<NJ>, dim3(batch), dim3(kWave), 0, st, L)); }
void trial_reference(int nj, int slots, hipStream_t st, const Launch& L) { KL_NJ(nj, hipLaunchKernelGGL(k_trial<NJ>, dim3(slots), dim3(kWave), 0, st, L)); }
int trial_fast_workgroups(int nj, int nodes) {
  const int per_wg = kTrialWaves * (nj == 10 ? LinFastCfg<10, true>::NPW : LinFastCfg<12, true>::NPW);
  return (nodes + per_wg - 1) / per_wg;
}
void trial_fast(int nj, int nodes, hipStream_t st, const Launch& L) {
  const int grid = trial_fast_workgroups(nj, nodes);
  KL_NJ(nj, {
    if (L.serial_legs) hipLaunchKernelGGL((k_trial_fast<NJ, true>), dim3(grid), dim3(kTrialWaves * kWave), 0, st, L);
    else hipLaunchKernelGGL((k_trial_fast<NJ, false>), dim3(grid), dim3(kTrialWaves * kWave), 0, st, L);
  });
}
void ls_decide(int nj, int batch, hipStream_t st, const Launch& L, bool look_first) { KL_NJ(nj, hipLaunchKernelGGL(k_ls_decide<NJ>, dim3(batch), dim3(kDecideThreads), 0, st, L, look_first ? 1 : 0)); }
void ls_tail(int nj, int batch, hipStream_t st, const Launch& L, int first_round, int max_trials, bool pending) {
  KL_NJ(nj, {
    if (L.serial_legs) hipLaunchKernelGGL((k_ls_tail<NJ, true>), dim3(batch), dim3(kTailThreads), 0, st, L, first_round, max_trials, pending ? 1 : 0);
    else hipLaunchKernelGGL((k_ls_tail<NJ, false>), dim3(batch), dim3(kTailThreads), 0, st, L, first_round, max_trials, pending ? 1 : 0);
  });
}
void constraint_values(int nj, int nodes, hipStream_t st, const Launch& L, double* eqv) {
  const int grid = trial_fast_workgroups(nj, nodes);
  KL_NJ(nj, {
    if (L.serial_legs) hipLaunchKernelGGL((k_constraint_values<NJ, true>), dim3(grid), dim3(kTrialWaves * kWave), 0, st, L, eqv);
    else hipLaunchKernelGGL((k_constraint_values<NJ, false>), dim3(grid), dim3(kTrialWaves * kWave), 0, st, L, eqv);
  });
}
void rollout(int nj, bool serial_legs, int batch, hipStream_t st, const DeviceModel* model, const RolloutArgs& a) {
  KL_NJ(nj, {
    constexpr bool kChainFits = LinFastCfg<NJ, false, true>::LPN == 16;      // 6 + nj coordinates in one DPP row (nj = 10)
    const dim3 grid((batch + LinFastCfg<NJ>::NPW - 1) / LinFastCfg<NJ>::NPW);
    if constexpr (kChainFits) {
      if (serial_legs) { hipLaunchKernelGGL((k_rollout<NJ, true>), grid, dim3(kWave), 0, st, model, a); return; }
    }
    hipLaunchKernelGGL((k_rollout<NJ, false>), grid, dim3(kWave), 0, st, model, a);
  });
}
void copy_pairs(int grid, hipStream_t st, const double* a_src, double* a_dst, size_t na, const double* b_src, double* b_dst, size_t nb,
                int* iterations, int* active, int batch) {
  hipLaunchKernelGGL(k_copy_pairs, dim3(grid > 0 ? grid : 1), dim3(256), 0, st, a_src, a_dst, na, b_src, b_dst, nb, iterations, active, batch);
}

}  // namespace kl
}  // namespace bpmpc
