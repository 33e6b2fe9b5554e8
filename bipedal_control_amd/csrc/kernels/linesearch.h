// Reference generation per node and the filter line search, batched.
//
//   prepare_node      x_ref(t) by clamped linear interpolation of the target trajectory ([OCS2-upstream]
//                     TargetTrajectories::getDesiredState; use site include/ocs2_bipedal_robot/cost/BipedalRobotQuadraticTrackingCost.h:57-63)
//                     and the cold-start iterate of BipedalRobotInitializer::compute (src/initialization/BipedalRobotInitializer.cpp:56-63).
//   linesearch_begin  PerformanceIndex of the linearised iterate (sum over nodes + initial-state mismatch).
//   trial_node        metrics of x + alpha dx, u + alpha du at one node ([OCS2-upstream] SqpSolver::computePerformance).
//   linesearch_decide [OCS2-upstream] FilterLinesearch::acceptStep + the alpha back-tracking loop of SqpSolver::takeStep
//                     and SqpSolver::checkConvergence (settings task.info:70-73; defaults alpha_decay 0.5, alpha_min 1e-4,
//                     gamma_c 1e-6, armijoFactor 1e-4, costTol 1e-4).
#pragma once
#include "node_lq.h"

namespace bpmpc {

constexpr int kMaxTargetPoints = 8;
constexpr int kStatsStride = 16;  // doubles per problem, see bpmpc_stats packing in solver

struct LineSearchSettings {
  double g_max, g_min, alpha_decay, alpha_min, gamma_c, armijo_factor, delta_tol, cost_tol;
  int max_iterations;
};

template <int NJ>
BP_DEVICE void prepare_node(const DeviceModel& md, int kind, int mode, double t_start, bool cold, bool last, int n_pts, const double* tgt_t,
                            const double* tgt_x, const double* x0, double* xref, double* x, double* u, double* x_terminal) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  BP_LANES(tid, kWave) {
    if (tid < NX) {
      double val;
      if (n_pts == 1 || t_start <= tgt_t[0]) {
        val = tgt_x[tid];
      } else if (t_start >= tgt_t[n_pts - 1]) {
        val = tgt_x[(n_pts - 1) * NX + tid];
      } else {
        int i = 0;
        while (i + 1 < n_pts - 1 && tgt_t[i + 1] < t_start) ++i;  // largest i with tgt_t[i] < t_start (lower_bound - 1)
        const double alpha = (tgt_t[i + 1] - t_start) / (tgt_t[i + 1] - tgt_t[i]);
        val = alpha * tgt_x[i * NX + tid] + (1.0 - alpha) * tgt_x[(i + 1) * NX + tid];
      }
      xref[tid] = val;
      if (cold) {
        x[tid] = x0[tid];
        if (last) x_terminal[tid] = x0[tid];
      }
    }
    if (cold && tid < NU) u[tid] = (kind == 1) ? 0.0 : nominal_input(md, mode, tid);
  }
}

template <int NJ>
struct TrialWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  NodeWorkspace<NJ> node;
  double tx[NX], tu[NU], txn[NX];
};

template <int NJ>
BP_DEVICE void trial_node(const DeviceModel& md, TrialWorkspace<NJ>& ws, NodeInputs in, double alpha, const double* dx, const double* du,
                          const double* dxn, double* perf) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  BP_LANES(tid, kWave) {
    if (tid < NX) {
      ws.tx[tid] = in.x[tid] + alpha * dx[tid];
      ws.txn[tid] = in.xnext[tid] + alpha * dxn[tid];
    }
    if (tid < NU) ws.tu[tid] = (in.kind == 1) ? 0.0 : in.u[tid] + alpha * du[tid];
  }
  BP_SYNC();
  in.x = ws.tx;
  in.u = ws.tu;
  in.xnext = ws.txn;
  node_performance<NJ>(md, ws.node, in, perf);
}

struct ProblemLS {
  int n_nodes;
  const double* node_perf;     // [n_nodes*3] of the linearisation
  const double* trial_perf;    // [n_nodes*3] of the current trial
  const double* x0;            // measured state
  double *x, *u;               // iterate (n_nodes+1)*NX, n_nodes*NU
  const double *dx, *du;
  const double* summary;       // riccati summary: armijo, |dx|^2, |du|^2, status
  double* base;                // [3] merit, dyn SSE, eq SSE of the linearised iterate
  double* alpha;               // current trial step size
  int* done;                   // line search finished for this problem
  int* active;                 // problem still iterating (SQP level)
  int* iterations;
  double* stats;               // [kStatsStride]
  int* remaining;              // global counter of unfinished line searches
};

template <int NJ>
BP_DEVICE void linesearch_begin(double* partial /*kWave*3 + 5 LDS*/, const ProblemLS& p) {
  constexpr int NX = 12 + NJ;
  BP_LANES(tid, kWave) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = tid; k < p.n_nodes; k += kWave) { a += p.node_perf[3 * k]; b += p.node_perf[3 * k + 1]; c += p.node_perf[3 * k + 2]; }
    if (tid < NX) { const double d = p.x0[tid] - p.x[tid]; b += d * d; }
    partial[tid] = a; partial[kWave + tid] = b; partial[2 * kWave + tid] = c;
  }
  BP_SYNC();
  // the three sums in lane order, one lane each (same numbers as one lane doing all three, a third of the time)
  BP_LANES(tid, kWave) {
    if (tid < 3) {
      double s = 0.0;
      for (int i = 0; i < kWave; ++i) s += partial[tid * kWave + i];
      partial[3 * kWave + 2 + tid] = s;
    }
  }
  BP_SYNC();
  BP_LANES(tid, kWave) {
    if (tid == 0) {
      const double a = partial[3 * kWave + 2], b = partial[3 * kWave + 3], c = partial[3 * kWave + 4];
      p.base[0] = a; p.base[1] = b; p.base[2] = c;
      p.alpha[0] = 1.0;
      const bool run = p.active[0] != 0;
      p.done[0] = run ? 0 : 1;
      if (run) {
#if defined(BPMPC_HOST_EMULATION)
        p.remaining[0] += 1;
#else
        atomicAdd(p.remaining, 1);
#endif
      }
    }
  }
}

// NL lanes work on one problem (64 in the lane emulation, 256 on the GPU: the accepted step touches (2N+1) nx doubles)
template <int NJ, int NL = kWave, bool PRE = true>     // PRE (device): the update's operands are requested before the sums (k_ls_decide; the back-tracking kernel has no registers for them)
BP_DEVICE void linesearch_decide(double* partial /*NL*3 LDS + 2 flags + 3 sums*/, const ProblemLS& p, const LineSearchSettings& st, bool look_first = false) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  if (look_first && p.done[0]) return;       // a launch in which nearly every problem is finished already (the second round): no requests for those
#if !defined(BPMPC_HOST_EMULATION)
  // Device: the entries of the iterate and of the step this lane updates if the trial is accepted are requested HERE, before the sums - their
  // addresses depend on the lane only.  As a loop of x[idx] += alpha dx[idx] behind the decision (x and dx may alias as far as the compiler
  // knows) the update was ten dependent memory round trips on four waves per CU: 16 us of a 20 us kernel at batch 256.
  constexpr int UPD = PRE ? (2400 + NL - 1) / NL : 0;             // covers (n + 1) nx of the reference's horizon; longer horizons finish in the loop below
  const int nxe = (p.n_nodes + 1) * NX, nue = p.n_nodes * NU;
  double xv[UPD + 1], dxv[UPD + 1], uv[UPD + 1], duv[UPD + 1];
#pragma unroll
  for (int j = 0; j < UPD; ++j) {
    const int idx = (int)threadIdx.x + j * NL;
    const int ix = idx < nxe ? idx : 0, iu = idx < nue ? idx : 0;
    xv[j] = p.x[ix]; dxv[j] = p.dx[ix]; uv[j] = p.u[iu]; duv[j] = p.du[iu];
  }
#endif
  if (p.done[0]) return;
  const double alpha = p.alpha[0];
  // the node sums are formed by the first NSUM lanes whatever the size of the workgroup, so that the first round (k_ls_decide, 256 lanes) and
  // the back-tracking rounds (k_ls_tail, 512 lanes) add the same terms in the same order at every horizon length
  constexpr int NSUM = NL < 256 ? NL : 256;
  BP_LANES(tid, NL) {
    double a = 0.0, b = 0.0, c = 0.0;
    if (tid < NSUM)
      for (int k = tid; k < p.n_nodes; k += NSUM) { a += p.trial_perf[3 * k]; b += p.trial_perf[3 * k + 1]; c += p.trial_perf[3 * k + 2]; }
    if (tid < NX) { const double d = p.x0[tid] - (p.x[tid] + alpha * p.dx[tid]); b += d * d; }
    partial[tid] = a; partial[NL + tid] = b; partial[2 * NL + tid] = c;
  }
  BP_SYNC();
  // the three sums in lane order, one lane each; lanes beyond the nodes (and beyond the nx mismatch terms) hold exact zeros and are
  // skipped - the numbers equal those of one lane adding all NSUM entries
  BP_LANES(tid, NL) {
    if (tid < 3) {
      int lim = p.n_nodes > NX ? p.n_nodes : NX;
      if (lim > NSUM) lim = NSUM;
      double s = 0.0;
      for (int i = 0; i < lim; ++i) s += partial[tid * NL + i];
      partial[3 * NL + 2 + tid] = s;
    }
  }
  BP_SYNC();
  BP_LANES(tid, NL) {
    if (tid == 0) {
      const double merit = partial[3 * NL + 2], dyn = partial[3 * NL + 3], eq = partial[3 * NL + 4];
      const double merit0 = p.base[0];
      const double viol0 = sqrt(p.base[1] + p.base[2]);
      const double viol = sqrt(dyn + eq);
      const double descent = alpha * p.summary[0];
      bool accepted;
      if (viol > st.g_max) {
        accepted = viol < (1.0 - st.gamma_c) * viol0;
      } else if (viol < st.g_min && viol0 < st.g_min && descent < 0.0) {
        accepted = merit < merit0 + st.armijo_factor * descent;
      } else {
        accepted = merit < (merit0 - st.gamma_c * viol0) || viol < (1.0 - st.gamma_c) * viol0;
      }
      const bool numerical = p.summary[3] != 0.0;
      if (numerical) accepted = false;
      const double next_alpha = alpha * st.alpha_decay;
      // [OCS2-upstream] SqpSolver::takeStep: back-tracking also stops (no step taken) once the next trial step would be shorter than
      // deltaTol in both the state and the input norm
      const bool tiny = next_alpha * sqrt(p.summary[1]) < st.delta_tol && next_alpha * sqrt(p.summary[2]) < st.delta_tol;
      const bool give_up = !accepted && (numerical || tiny || !(next_alpha >= st.alpha_min));
      partial[3 * NL] = accepted ? 1.0 : 0.0;
      partial[3 * NL + 1] = give_up ? 1.0 : 0.0;
      if (accepted || give_up) {
        double* s = p.stats;
        const int it = p.iterations[0] + 1;
        p.iterations[0] = it;
        const double dxn = accepted ? alpha * sqrt(p.summary[1]) : 0.0, dun = accepted ? alpha * sqrt(p.summary[2]) : 0.0;
        s[0] = (double)p.n_nodes;
        s[1] = (double)it;
        s[2] = numerical ? 2.0 : (accepted ? 0.0 : 1.0);
        s[3] = merit0; s[4] = p.base[1]; s[5] = p.base[2];
        s[6] = accepted ? merit : merit0; s[7] = accepted ? dyn : p.base[1]; s[8] = accepted ? eq : p.base[2];
        s[9] = accepted ? alpha : 0.0;
        s[10] = p.summary[0];
        s[11] = dxn; s[12] = dun;
        // [OCS2-upstream] SqpSolver::checkConvergence
        bool keep_going = it < st.max_iterations;
        if (keep_going && s[9] < st.alpha_min) keep_going = false;
        if (keep_going && fabs(s[6] - merit0) < st.cost_tol && sqrt(s[7] + s[8]) < st.g_min) keep_going = false;
        if (keep_going && dxn < st.delta_tol && dun < st.delta_tol) keep_going = false;
        p.active[0] = keep_going ? 1 : 0;
        p.done[0] = 1;
#if defined(BPMPC_HOST_EMULATION)
        p.remaining[0] -= 1;
#else
        atomicSub(p.remaining, 1);
#endif
      } else {
        p.alpha[0] = next_alpha;
      }
    }
  }
  BP_SYNC();
  const bool accepted = partial[3 * NL] != 0.0;
  if (accepted) {
#if !defined(BPMPC_HOST_EMULATION)
#pragma unroll
    for (int j = 0; j < UPD; ++j) {
      const int idx = (int)threadIdx.x + j * NL;
      if (idx < nxe) p.x[idx] = xv[j] + alpha * dxv[j];
      if (idx < nue) p.u[idx] = uv[j] + alpha * duv[j];
    }
    int t0 = (int)threadIdx.x;          // (opaque: as an invariant of the back-tracking kernel's round loop the lane's byte offset lived across the trial evaluation - in scratch memory at nx = 24)
    if constexpr (!PRE) asm volatile("" : "+v"(t0));
    for (int idx = t0 + UPD * NL; idx < nxe; idx += NL) p.x[idx] += alpha * p.dx[idx];
    for (int idx = t0 + UPD * NL; idx < nue; idx += NL) p.u[idx] += alpha * p.du[idx];
#else
    BP_LANES(tid, NL) {
      for (int idx = tid; idx < (p.n_nodes + 1) * NX; idx += NL) p.x[idx] += alpha * p.dx[idx];
      for (int idx = tid; idx < p.n_nodes * NU; idx += NL) p.u[idx] += alpha * p.du[idx];
    }
#endif
  }
}

}  // namespace bpmpc
