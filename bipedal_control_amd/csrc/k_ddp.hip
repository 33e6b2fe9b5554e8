// Kernels of the DDP slice (GaussNewtonDDP_MPC of ocs2_bipedal_robot_ros/src/BipedalRobotDdpMpcNode.cpp:70-71 with
// `algorithm ILQR`, settings task.info:115-156): what one ILQR iteration needs beside the kernels the SQP path already has.
//   backward pass   the lineariser with Launch::ilqr (Euler discretisation of the continuous-time model, Hessian shift; linearize_fast.h ILQR,
//                   reference body node_lq.h), then the constraint elimination, change of variables and Riccati sweep of the SQP path (round 6:
//                   the fast kernels) - any exact solution of the equality-constrained stage problems is THE ILQR policy; k_ddp_policy reads it
//                   back as du = lff + K dx
//   line search     ALL step lengths in one launch each of: the policy roll-out of kernels/rollout.h with its observer (TimeTriggeredRollout,
//                   ODE45; every accepted step is a time point; planned inputs u_nom + alpha lff formed where they are loaded), k_ddp_cost
//                   (intermediate cost at every time point), k_ddp_select (trapezoidal performance indices, Armijo test against the
//                   step-length-0 baseline, the largest accepted step)
//   result          k_ddp_finish: the accepted roll-out ON ITS OWN TIME POINTS becomes the solution (x, u, times, stats)
// [OCS2-upstream, recalled] throughout (oracle/ddp_py.py states what is and what is not restated; tests/test_recalled_behaviours.py names it).
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernel_launchers.h"
#include "launch.h"
#include "kernels/node_lq.h"
#include "kernels/linesearch.h"
#include "kernels/rollout.h"

namespace bpmpc {

namespace {

// [OCS2-upstream] LinearInterpolation::timeSegment on the node times of a grid: the interval (t_j, t_{j+1}] that holds t (clamped)
__device__ inline int ddp_interval(const double* t, int n_nodes, double q) {
  if (q <= t[0]) return 0;
  if (q >= t[n_nodes]) return n_nodes - 1;
  int lo = 0, hi = n_nodes + 1;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (t[mid] < q) lo = mid + 1; else hi = mid; }
  int i = lo - 1;
  return i < 0 ? 0 : (i > n_nodes - 1 ? n_nodes - 1 : i);
}

}  // namespace

// lff_k = du_k - K_k dx_k of the sweep's linear roll-out (zero at event nodes), int |lff|^2 dt, and the flags of the search
template <int NJ>
__global__ __launch_bounds__(kWave) void k_ddp_policy(Launch L, DdpBuffers d) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int grid = L.buf.p_grid[b], n = L.buf.g_nodes[grid];
  __shared__ double part[kWave];
  double acc = 0.0;
  for (int k = 0; k < n; ++k) {
    const size_t s = (size_t)b * L.N + k;
    const int kind = L.buf.g_kind[(size_t)grid * L.N + k];
    double l = 0.0;
    if (tid < NU && kind == 0) {
      const double* Kr = L.buf.K + s * NU * NX + (size_t)tid * NX;
      const double* dx = L.buf.dx + ((size_t)b * (L.N + 1) + k) * NX;
      double t = L.buf.du[s * NU + tid];
      for (int c = 0; c < NX; ++c) t -= Kr[c] * dx[c];
      l = t;
    }
    if (tid < NU) d.lff[s * NU + tid] = l;
    acc += L.buf.g_dt[(size_t)grid * L.N + k] * l * l;
  }
  part[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < NU; ++i) s += part[i];
    d.update_is[b] = s;
    d.accepted[b] = 0;
    d.alpha[b] = 0.0;
    d.failed[b] = L.buf.summary[(size_t)b * 4 + 3] != 0.0 ? 1 : 0;      // the sweep met a non-positive pivot: no policy
  }
}

// intermediate cost at every time point of every recorded roll-out: the node metric of the transcription with dt = 1 (tracking + soft cones)
template <int NJ>
__global__ __launch_bounds__(kWave) void k_ddp_cost(Launch L, DdpBuffers d) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  __shared__ NodeWorkspace<NJ> ws;
  __shared__ double xref[NX], zero4[4], perf[3];
  const int vb = blockIdx.x / d.cap, i = blockIdx.x % d.cap, tid = threadIdx.x;      // vb: step length * batch + problem
  if (i >= d.rec_n[vb]) return;
  const int b = vb % L.batch;
  if (d.failed[b]) return;
  const int grid = L.buf.p_grid[b], n = L.buf.g_nodes[grid];
  const size_t at = (size_t)vb * d.cap + i;
  const double t = d.rec_t[at];
  const int j = ddp_interval(L.buf.g_time + (size_t)grid * (L.N + 1), n, t);
  // x_ref(t): TargetTrajectories::getDesiredState (clamped linear interpolation), as prepare_node
  {
    const int n_pts = L.buf.p_tgt_n[b];
    const double* tt = L.buf.p_tgt_t + (size_t)b * kMaxTargetPoints;
    const double* tx = L.buf.p_tgt_x + (size_t)b * kMaxTargetPoints * NX;
    if (tid < NX) {
      double val;
      if (n_pts == 1 || t <= tt[0]) val = tx[tid];
      else if (t >= tt[n_pts - 1]) val = tx[(n_pts - 1) * NX + tid];
      else {
        int q = 0;
        while (q + 1 < n_pts - 1 && tt[q + 1] < t) ++q;
        const double al = (tt[q + 1] - t) / (tt[q + 1] - tt[q]);
        val = al * tx[q * NX + tid] + (1.0 - al) * tx[(q + 1) * NX + tid];
      }
      xref[tid] = val;
    }
    if (tid < 4) zero4[tid] = 0.0;
  }
  __syncthreads();
  NodeInputs in;
  in.kind = 0; in.mode = L.buf.g_mode[(size_t)grid * L.N + j]; in.dt = 1.0;
  in.x = d.rec_x + at * NX; in.xnext = in.x; in.u = d.rec_u + at * NU; in.xref = xref; in.zref = zero4; in.zdref = zero4;
  node_performance<NJ>(*L.model, ws, in, perf);
  __syncthreads();
  if (tid == 0) d.cost[at * 3] = perf[0];
}

// The same on the lane-per-coordinate evaluation of the line search (linearize_fast.h trial_fast with a zero step and dt = 1: its cost entry is the
// cost rate), four time points per wave, sixteen per workgroup - serial-leg robots (every robot of the reference); round 6: the last kernel of the
// DDP path that ran on a lane-emulated body (64 B of scratch, 2.0 ms of the 20.5 ms step at 256 x 100).
constexpr int kCostWaves = 4;
template <int NJ>
__global__ __launch_bounds__(kCostWaves * kWave) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ddp_cost_fast(Launch L, DdpBuffers d) {
  using C = LinFastCfg<NJ, true, true>;
  constexpr int NX = 12 + NJ, NU = 12 + NJ, LPN = C::LPN, PTS = kCostWaves * C::NPW;
  static_assert(LPN == 16, "sixteen lanes per time point");
  __shared__ LinFastNodeLds<NJ, false, true> lds[PTS];
  __shared__ LinFastShared<NJ, false> shared;
  __shared__ double xref[PTS][NX];
  const int sub = threadIdx.x / LPN, g = threadIdx.x % LPN;
  const long long widx = (long long)blockIdx.x * PTS + sub, total = (long long)d.nv * L.batch * d.cap;
  bool valid = widx < total;
  const int vb = valid ? (int)(widx / d.cap) : 0, i = valid ? (int)(widx % d.cap) : 0;      // vb: step length * batch + problem
  const int b = vb % L.batch;
  {   // a workgroup beyond the recorded points of its roll-out(s) has nothing to do (most of them: the record holds cap points, a roll-out ~80)
    const long long w0 = (long long)blockIdx.x * PTS, w1 = w0 + PTS - 1 < total ? w0 + PTS - 1 : total - 1;
    const int v0 = (int)(w0 / d.cap), v1 = (int)(w1 / d.cap);
    if (v0 == v1 && (int)(w0 % d.cap) >= d.rec_n[v0]) return;
  }
  load_shared_model<kCostWaves * kWave>(*L.model, shared, threadIdx.x);
  valid = valid && i < d.rec_n[vb] && !d.failed[b];
  const int grid = L.buf.p_grid[b], n = L.buf.g_nodes[grid];
  const size_t at = valid ? (size_t)vb * d.cap + i : 0;
  const double t = d.rec_t[at];
  const int j = ddp_interval(L.buf.g_time + (size_t)grid * (L.N + 1), n, t);
  {   // x_ref(t): TargetTrajectories::getDesiredState (clamped linear interpolation), as prepare_node
    const int n_pts = L.buf.p_tgt_n[b];
    const double* tt = L.buf.p_tgt_t + (size_t)b * kMaxTargetPoints;
    const double* tx = L.buf.p_tgt_x + (size_t)b * kMaxTargetPoints * NX;
    for (int c = g; c < NX; c += LPN) {
      double val;
      if (n_pts == 1 || t <= tt[0]) val = tx[c];
      else if (t >= tt[n_pts - 1]) val = tx[(n_pts - 1) * NX + c];
      else {
        int q = 0;
        while (q + 1 < n_pts - 1 && tt[q + 1] < t) ++q;
        const double al = (tt[q + 1] - t) / (tt[q + 1] - tt[q]);
        val = al * tx[q * NX + c] + (1.0 - al) * tx[(q + 1) * NX + c];
      }
      xref[sub][c] = val;
    }
  }
  __syncthreads();
  NodeInputs in;
  in.kind = 0; in.mode = L.buf.g_mode[(size_t)grid * L.N + j]; in.dt = 1.0;
  in.x = d.rec_x + at * NX; in.xnext = in.x; in.u = d.rec_u + at * NU; in.xref = xref[sub];
  in.zref = L.buf.zero_page + 4; in.zdref = L.buf.zero_page + 4;      // (a row of zeros)
  trial_fast<NJ, C>(*L.model, shared, lds[sub], valid, in, 0.0, in.x, in.u, in.x, d.cost + at * 3, g);
}

// performance index of every recorded roll-out (trapezoidalIntegration over its time points), Armijo test of the step lengths against the
// step-length-0 baseline in descending order - the LARGEST accepted step wins ([OCS2-upstream] LineSearchStrategy evaluates them concurrently
// and keeps the largest that passes) -, and the winner (else the baseline) becomes the solution record
template <int NJ>
__global__ __launch_bounds__(kWave) void k_ddp_select(Launch L, DdpBuffers d, double armijo) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ int take;
  if (tid == 0) {
    auto merit_of = [&](int v) {
      const size_t vb = (size_t)v * L.batch + b;
      const int n = d.rec_n[vb];
      const double* t = d.rec_t + vb * d.cap;
      const double* c = d.cost + vb * d.cap * 3;      // (three metrics per point as the trial evaluation writes them: cost rate, defect, equality SSE)
      double m = 0.0;
      for (int i = 0; i + 1 < n; ++i) m += 0.5 * (c[3 * (i + 1)] + c[3 * i]) * (t[i + 1] - t[i]);
      return m;
    };
    auto rolled = [&](int v) { const size_t vb = (size_t)v * L.batch + b; return d.roll_status[vb] == 0 && d.rec_n[vb] >= 2; };
    int tk = -1;
    if (!d.failed[b]) {
      if (!rolled(0)) {
        d.failed[b] = 2;                                  // no baseline: nothing to compare with, the nominal trajectories stay (2: the roll-out failed, 1: the sweep)
        d.merit0[b] = 0.0; d.merit[b] = 0.0;
      } else {
        const double m0 = merit_of(0);
        d.merit0[b] = m0; d.merit[b] = m0;
        tk = 0;
        for (int v = 1; v < d.nv; ++v) {
          if (!rolled(v)) continue;
          const double m = merit_of(v);
          if (m < m0 - armijo * d.alpha_v[v] * d.update_is[b]) { d.accepted[b] = 1; d.alpha[b] = d.alpha_v[v]; d.merit[b] = m; tk = v; break; }
        }
      }
    }
    take = tk;
  }
  __syncthreads();
  if (take < 0) return;
  const size_t vb = (size_t)take * L.batch + b;
  const int n = d.rec_n[vb];
  if (tid == 0) d.sol_n[b] = n;
  for (int i = tid; i < n; i += kWave) d.sol_t[(size_t)b * d.cap + i] = d.rec_t[vb * d.cap + i];
  for (int i = tid; i < n * NX; i += kWave) d.sol_x[(size_t)b * d.cap * NX + i] = d.rec_x[vb * d.cap * NX + i];
  for (int i = tid; i < n * NU; i += kWave) d.sol_u[(size_t)b * d.cap * NU + i] = d.rec_u[vb * d.cap * NU + i];
}

// the solution in the solver's own arrays: x[b][i], u[b][i] at the roll-out's time points (the input at the last point is dropped: the
// arrays hold one input per interval), statistics
template <int NJ>
__global__ __launch_bounds__(kWave) void k_ddp_finish(Launch L, DdpBuffers d) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int why = d.failed[b];
  const bool failed = why != 0;
  const int n = failed ? 0 : d.sol_n[b];
  double* x = L.buf.x + (size_t)b * (L.N + 1) * NX;
  double* u = L.buf.u + (size_t)b * L.N * NU;
  for (int i = tid; i < n * NX; i += kWave) x[i] = d.sol_x[(size_t)b * d.cap * NX + i];
  for (int i = tid; i < (n - 1) * NU; i += kWave) u[i] = d.sol_u[(size_t)b * d.cap * NU + i];
  if (tid == 0) {
    double* s = L.buf.stats + (size_t)b * kStatsStride;
    const int it = L.buf.iterations[b] + 1;
    L.buf.iterations[b] = it;
    s[0] = (double)(failed ? L.buf.g_nodes[L.buf.p_grid[b]] : n - 1);
    s[1] = (double)it;
    // 0: a step was accepted, 1: none (the baseline roll-out is the solution), 2: numerical failure in the Riccati sweep (nominal kept),
    // 3: the baseline roll-out failed - integrator out of steps or more time points than the record holds (nominal kept)
    s[2] = failed ? (why == 2 ? 3.0 : 2.0) : (d.accepted[b] ? 0.0 : 1.0);
    s[3] = d.merit0[b]; s[4] = 0.0; s[5] = 0.0;
    s[6] = d.merit[b]; s[7] = 0.0; s[8] = 0.0;
    s[9] = d.alpha[b];
    s[10] = -d.update_is[b];
    s[11] = 0.0; s[12] = 0.0;
    L.buf.active[b] = 0;
    d.n_points[b] = failed ? 0 : n;
  }
}

// Nominal trajectories of a receding-horizon tick ([OCS2-upstream, recalled] GaussNewtonDDP::rolloutInitialTrajectory with a controller from the
// previous run: the previous controller - a FeedforwardController, here already shifted onto the new grid by k_warm_shift, the initializer's input
// beyond its end - is ROLLED OUT from the measured state; the state trajectory of that roll-out, not the previous solution, is what the LQ
// approximation is built on, so that the dynamics bias vanishes as ILQR assumes).  The roll-out lives on its own time points; the backward pass of
// this engine on the shooting grid: node k takes LinearInterpolation(t_k) of the record (the states are continuous across the events, pre- and
// post-event node share their time).  A problem whose roll-out failed keeps the shifted previous solution.  One wave per (problem, node).
template <int NJ>
__global__ __launch_bounds__(kWave) void k_ddp_nominal(Launch L, DdpBuffers d) {
  constexpr int NX = 12 + NJ;
  const int b = blockIdx.x / (L.N + 1), k = blockIdx.x % (L.N + 1), l = threadIdx.x;
  const int grid = L.buf.p_grid[b], n = L.buf.g_nodes[grid];
  if (k > n) return;
  const int np = d.rec_n[b];
  if (d.roll_status[b] != 0 || np < 2) return;
  const double* rt = d.rec_t + (size_t)b * d.cap;
  const double tq = L.buf.g_time[(size_t)grid * (L.N + 1) + k];
  int i;
  double al;
  time_segment(rt, np, tq, &i, &al);
  if (l < NX) {
    const double* r0 = d.rec_x + ((size_t)b * d.cap + i) * NX;
    L.buf.x[((size_t)b * (L.N + 1) + k) * NX + l] = k == 0 ? L.buf.p_x0[(size_t)b * NX + l] : al * r0[l] + (1.0 - al) * r0[NX + l];
  }
}

// receding-horizon warm start of the next run: the previous solution is a FeedforwardController on the roll-out's own time points, one time
// trajectory per problem (k_warm_shift reads tp_* per grid: problem b becomes its own "grid"), no pre-event entries
__global__ void k_ddp_keep_times(DdpBuffers d, int batch, int N, double* tp_time, int* tp_kind, int* tp_nodes, int* tp_grid) {
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int n = d.n_points[b];
  if (n <= 0) return;      // a failed problem keeps its nominal trajectories ON THE GRID: the copies preserve_previous made of the grid tables stay
  for (int i = threadIdx.x; i < N + 1; i += blockDim.x) tp_time[(size_t)b * (N + 1) + i] = i < n ? d.sol_t[(size_t)b * d.cap + i] : 0.0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) tp_kind[(size_t)b * N + i] = 0;
  if (threadIdx.x == 0) { tp_nodes[b] = n - 1; tp_grid[b] = b; }
}

#define KL_NJ(nj, ...)                                                          \
  do {                                                                          \
    if ((nj) == 10) { constexpr int NJ = 10; __VA_ARGS__; }                     \
    else if ((nj) == 12) { constexpr int NJ = 12; __VA_ARGS__; }                \
    else throw std::runtime_error("unsupported joint count");                   \
  } while (0)

namespace kl {

void ddp_policy(int nj, int batch, hipStream_t st, const Launch& L, const DdpBuffers& d) { KL_NJ(nj, hipLaunchKernelGGL(k_ddp_policy<NJ>, dim3(batch), dim3(kWave), 0, st, L, d)); }
void ddp_cost(int nj, int batch, bool fast, hipStream_t st, const Launch& L, const DdpBuffers& d) {
  if (fast) {
    const long long pts = (long long)batch * d.nv * d.cap;
    KL_NJ(nj, { constexpr int per = kCostWaves * LinFastCfg<NJ, true, true>::NPW; hipLaunchKernelGGL(k_ddp_cost_fast<NJ>, dim3((unsigned)((pts + per - 1) / per)), dim3(kCostWaves * kWave), 0, st, L, d); });
  } else KL_NJ(nj, hipLaunchKernelGGL(k_ddp_cost<NJ>, dim3(batch * d.nv * d.cap), dim3(kWave), 0, st, L, d));
}
void ddp_select(int nj, int batch, hipStream_t st, const Launch& L, const DdpBuffers& d, double armijo) {
  KL_NJ(nj, hipLaunchKernelGGL(k_ddp_select<NJ>, dim3(batch), dim3(kWave), 0, st, L, d, armijo));
}
void ddp_nominal(int nj, int batch, hipStream_t st, const Launch& L, const DdpBuffers& d) { KL_NJ(nj, hipLaunchKernelGGL(k_ddp_nominal<NJ>, dim3(batch * (L.N + 1)), dim3(kWave), 0, st, L, d)); }
void ddp_keep_times(int batch, int N, hipStream_t st, const DdpBuffers& d, double* tp_time, int* tp_kind, int* tp_nodes, int* tp_grid) {
  hipLaunchKernelGGL(k_ddp_keep_times, dim3(batch), dim3(kWave), 0, st, d, batch, N, tp_time, tp_kind, tp_nodes, tp_grid);
}
void ddp_finish(int nj, int batch, hipStream_t st, const Launch& L, const DdpBuffers& d) { KL_NJ(nj, hipLaunchKernelGGL(k_ddp_finish<NJ>, dim3(batch), dim3(kWave), 0, st, L, d)); }

}  // namespace kl
}  // namespace bpmpc
