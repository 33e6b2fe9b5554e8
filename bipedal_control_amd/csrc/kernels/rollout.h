// MRT side of the MPC loop on the device: MRT_BASE::rolloutPolicy for a whole batch, i.e.
// TimeTriggeredRollout::run over [t, t + duration] under the LinearController of the last solution,
//   u(t, x) = uff(t) + K(t) x,  uff_j = u_j - K_j x_j,  linear interpolation in time (LinearController::computeInput),
// integrated with the controlled Dormand-Prince 5(4) stepper of boost::numeric::odeint (integrate_adaptive, FSAL, error
// norm |err_i| / (abs + rel (|x_i| + dt |dxdt_i|)), step factors 0.9 err^(-1/3) >= 0.2 on rejection, 0.9 max(err, 5^-5)^(-1/5) on
// success when err < 0.5), restarted at every mode-schedule event inside the window with the begin time nudged by
// weakEpsilon (RolloutBase::findActiveModesTimeInterval).  Used by MRT_ROS_Dummy_Loop
// (ocs2_bipedal_robot_ros/src/BipedalRobotDummyNode.cpp:61,72-86) and BipedalController.cpp:322 with the rollout block of
// task.info:158-167.  [OCS2-upstream / boost, recalled]; oracle: oracle/reference_py.py time_triggered_rollout.
//
// Mapping: as the line-search trial kernel - one lane per generalised coordinate, 16 (32) lanes per problem, the flow map is the
// value-only eval_lane of linearize_fast.h.  Lane g < 6 carries normalised momentum g, lane g < 6 + NJ carries coordinate g.
// Problems of one wavefront step in lock step (finished ones idle); everything else is per-problem state in registers.
#pragma once
#include "linearize_fast.h"

namespace bpmpc {

constexpr int kRolloutMaxEvents = 32;

// [OCS2-upstream] LinearInterpolation::timeSegment: value(q) = alpha v[idx] + (1 - alpha) v[idx + 1]
__device__ __forceinline__ void time_segment(const double* t, int n, double q, int* idx, double* alpha) {
  if (q <= t[0]) { *idx = 0; *alpha = 1.0; return; }
  if (q >= t[n - 1]) { *idx = n - 2; *alpha = 0.0; return; }
  int lo = 0, hi = n;                       // lower_bound: first element >= q
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (t[mid] < q) lo = mid + 1; else hi = mid; }
  int i = lo - 1;
  i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
  *idx = i;
  *alpha = (t[i + 1] - q) / (t[i + 1] - t[i]);
}

struct RolloutArgs {
  int batch, N;                       // N = node stride of the solution arrays
  const int* p_grid;                  // problem -> grid
  const int* g_nodes;                 // per grid: number of intervals
  const int* g_kind;                  // [grid][N]: 1 = pre-event node
  const double* g_time;               // [grid][N + 1] node times
  const double *x, *u, *K;            // solution: [batch][N + 1][NX], [batch][N][NU], [batch][N][NU][NX]
  const double* t_start;              // [batch]
  const double* x_start;              // [batch][NX]
  double duration, abs_tol, rel_tol, time_step;
  int max_steps;
  int feedback;                       // sqp.useFeedbackPolicy: 1 LinearController u = uff(t) + K(t) x, 0 FeedforwardController u = u(t)
  double* x_end;                      // [batch][NX]
  double* u_end;                      // [batch][NU]: controller at the final time and state
  int* steps;                         // [batch][2]: accepted, rejected
  int* status;                        // [batch]: 0 ok, 1 max steps, 2 no step size found, 3 too many events in the window, 4 more time points than rec_cap
  // optional record of the roll-out's own time points ([OCS2-upstream] the observer of integrate_adaptive: the begin of every segment, every
  // accepted step): rec_t [batch][rec_cap], rec_x [batch][rec_cap][NX], rec_u [batch][rec_cap][NU] (the controller at the point), rec_n [batch]
  double *rec_t, *rec_x, *rec_u;
  int* rec_n;
  int rec_cap;
  // optional variants of the planned inputs (the DDP line search rolls every step length out in ONE launch): `batch` counts virtual problems
  // bq = variant * n_problems + problem; everything that is read (solution, grid, start) is indexed by the problem, everything that is written
  // (x_end, u_end, steps, status, rec_*) by bq; the planned input of variant v is u + alpha[v] lff.  n_problems = 0: no variants.
  int n_problems;
  const double* lff;                  // [n_problems][N][NU]
  double alpha[kMaxDdpSteps];
  // t_start = nullptr: every problem starts at the first time of its grid; duration < 0: ... and runs to the last
};

// CHAIN (round 6): the tree walks of the flow map by DPP between neighbouring lanes (linearize_fast.h), for robots of two serial legs whose 6 + nj
// coordinates fit one 16-lane row (nj = 10: H1, Hunter); else the LDS tables
template <int NJ, bool CHAIN = false>
struct RolloutLds {
  using C = LinFastCfg<NJ, false, CHAIN>;
  static_assert(!CHAIN || C::LPN == 16, "the DPP walks stay inside a 16-lane row");
  LinFastNodeLds<NJ, false, CHAIN> node[C::NPW];
  LinFastShared<NJ, false> shared;
  double event[C::NPW][kRolloutMaxEvents];
  int n_events[C::NPW];
  // controller data of the time segment the integrator is in: K and u of the two (effective) nodes, the two planned states
  double Kc[C::NPW][2][C::NU * C::NX], uc[C::NPW][2][C::NU], xc[C::NPW][2][C::NX];
};

template <int LPN>
__device__ __forceinline__ double node_allreduce_max(double x) {
#pragma unroll
  for (int m = 1; m < LPN; m <<= 1) x = fmax(x, __shfl_xor(x, m));
  return x;
}

template <int NJ, bool CHAIN = false>
__device__ __forceinline__ void rollout_policy(const DeviceModel& md, RolloutLds<NJ, CHAIN>& w, const RolloutArgs& a) {
  using C = LinFastCfg<NJ, false, CHAIN>;
  constexpr int G = C::G, NX = C::NX, NU = C::NU, LPN = C::LPN, NPW = C::NPW;
  const int sub = threadIdx.x / LPN, g = threadIdx.x % LPN;
  const int bq = blockIdx.x * NPW + sub;
  const bool valid = bq < a.batch;
  const int ob = valid ? bq : 0;                                    // where this roll-out writes
  const int b = a.n_problems > 0 ? ob % a.n_problems : ob;          // the problem it reads
  const double step_len = a.n_problems > 0 ? a.alpha[ob / a.n_problems] : 0.0;
  const double* lf = a.n_problems > 0 ? a.lff + (size_t)b * a.N * NU : nullptr;
  LinFastNodeLds<NJ, false, CHAIN>& nl = w.node[sub];
  const LinFastShared<NJ, false>& sh = w.shared;
  const int N = a.N, grid = a.p_grid[b], n = a.g_nodes[grid];
  const double* tp = a.g_time + (size_t)grid * (N + 1);
  const int* kp = a.g_kind + (size_t)grid * N;
  const double* xp = a.x + (size_t)b * (N + 1) * NX;
  const double* up = a.u + (size_t)b * N * NU;
  const double* Kp = a.K + (size_t)b * N * NU * NX;
  const bool is_joint = g >= 6 && g < G;
  LaneBody lb;
  {
    const int body = (g >= 5 && g < G) ? g - 5 : 0;
    lb.body = body;
    lb.depth = sh.depth[body];
    lb.subtree = sh.subtree[body];
  }
  const int* path = sh.path[lb.body];
  const double t0 = a.t_start ? a.t_start[b] : tp[0], tf = a.duration < 0.0 ? tp[n] : t0 + a.duration;
  // events of the window: the pre-event nodes of the solution grid with t0 < t <= tf (upper_bound on both ends)
  if (g == 0) {
    int cnt = 0;
    for (int k = 0; k < n; ++k)
      if (kp[k] == 1 && tp[k] > t0 && tp[k] <= tf) { if (cnt < kRolloutMaxEvents) w.event[sub][cnt] = tp[k]; ++cnt; }
    w.n_events[sub] = cnt;
  }
  lds_wave_sync();
  const int n_events = w.n_events[sub];
  int status = n_events > kRolloutMaxEvents ? 3 : 0;

  auto effective = [&](int j) { while (j > 0 && (j == n || kp[j] == 1)) --j; return j; };   // repeated input / gain (toPrimalSolution)
  // publish a state for the other lanes of the problem
  auto publish = [&](double vh, double vq) {
    lds_wave_sync();
    if (g < 6) nl.x[g] = vh;
    if (g < G) nl.x[6 + g] = vq;
    lds_wave_sync();
  };
  // LinearController::computeInput at time ts for the published state -> nl.u.  The seven stages of a step nearly always fall into
  // the same segment of the solution's time grid, so its index, the two node times and the controller data (K, u, planned x of both
  // ends) are kept - the segment test below is LinearInterpolation::timeSegment's own (t_j < t <= t_{j+1}, clamped at the ends).
  int seg_j = -1;
  double seg_lo = 0.0, seg_hi = 0.0;
  const double t_first = tp[0], t_last = tp[n];
  auto controller = [&](double ts) {
    const bool hit = seg_j >= 0 && ((ts > seg_lo && ts <= seg_hi) || (seg_j == 0 && ts <= seg_lo) || (seg_j == n - 1 && ts >= seg_hi));
    if (!hit) {
      int j;
      double unused;
      time_segment(tp, n + 1, ts, &j, &unused);
      seg_j = j; seg_lo = tp[j]; seg_hi = tp[j + 1];
      const int e0 = effective(j), e1 = effective(j + 1);
      lds_wave_sync();                                  // earlier readers of the cached rows are done
      // Round 6: the gains of the two nodes arrive in batches of kBatch requests per lane and node (all issued before the first is stored: one memory
      // round trip per batch, where the plain copy loop paid one per element - 30 dependent trips per segment change, and a roll-out changes segment
      // at nearly every step: the DDP line search spent most of its 22 ms here)
      {
        constexpr int kBatch = 8, kTotal = NU * NX;
        const double* K0g = Kp + (size_t)e0 * kTotal;
        const double* K1g = Kp + (size_t)e1 * kTotal;
        for (int base = g; base < kTotal; base += LPN * kBatch) {
          double r0[kBatch], r1[kBatch];
#pragma unroll
          for (int c = 0; c < kBatch; ++c) { const int idx = base + c * LPN; const int at = idx < kTotal ? idx : 0; r0[c] = K0g[at]; r1[c] = K1g[at]; }
#pragma unroll
          for (int c = 0; c < kBatch; ++c) asm volatile("" : "+v"(r0[c]), "+v"(r1[c]));
#pragma unroll
          for (int c = 0; c < kBatch; ++c) { const int idx = base + c * LPN; if (idx < kTotal) { w.Kc[sub][0][idx] = r0[c]; w.Kc[sub][1][idx] = r1[c]; } }
        }
      }
      if (lf) for (int idx = g; idx < NU; idx += LPN) { w.uc[sub][0][idx] = up[(size_t)e0 * NU + idx] + step_len * lf[(size_t)e0 * NU + idx]; w.uc[sub][1][idx] = up[(size_t)e1 * NU + idx] + step_len * lf[(size_t)e1 * NU + idx]; }
      else for (int idx = g; idx < NU; idx += LPN) { w.uc[sub][0][idx] = up[(size_t)e0 * NU + idx]; w.uc[sub][1][idx] = up[(size_t)e1 * NU + idx]; }
      for (int idx = g; idx < NX; idx += LPN) { w.xc[sub][0][idx] = xp[(size_t)j * NX + idx]; w.xc[sub][1][idx] = xp[(size_t)(j + 1) * NX + idx]; }
      if (a.feedback) {     // LinearController's own form: uff_j = u_j - K_j x_j once per segment (round 6), so that a stage costs K x instead of K (x - x_j)
        lds_wave_sync();
        for (int r = g; r < NU; r += LPN) {
          const double* K0 = w.Kc[sub][0] + r * NX;
          const double* K1 = w.Kc[sub][1] + r * NX;
          double f0 = w.uc[sub][0][r], f1 = w.uc[sub][1][r];
          for (int c = 0; c < NX; ++c) { f0 -= K0[c] * w.xc[sub][0][c]; f1 -= K1[c] * w.xc[sub][1][c]; }
          w.uc[sub][0][r] = f0; w.uc[sub][1][r] = f1;
        }
      }
    }
    lds_wave_sync();
    double al;                                          // as time_segment: clamped outside the grid
    if (ts <= t_first) al = 1.0;
    else if (ts >= t_last) al = 0.0;
    else al = (seg_hi - ts) / (seg_hi - seg_lo);
    for (int r = g; r < NU; r += LPN) {
      const double* K0 = w.Kc[sub][0] + r * NX;
      const double* K1 = w.Kc[sub][1] + r * NX;
      double s0 = w.uc[sub][0][r], s1 = w.uc[sub][1][r];
      if (a.feedback)
        for (int c = 0; c < NX; ++c) {
          const double xc = nl.x[c];
          s0 += K0[c] * xc;
          s1 += K1[c] * xc;
        }
      nl.u[r] = al * s0 + (1.0 - al) * s1;
    }
    lds_wave_sync();
  };
  // dx/dt of the controlled system at (ts, state): this lane's momentum and coordinate rows
  auto deriv = [&](double ts, double vh, double vq, double& kh, double& kq) {
    publish(vh, vq);
    controller(ts);
    const double ujg = is_joint ? nl.u[12 + g - 6] : 0.0;
    LaneEval e;
    LaneKin<NJ> kin;
    eval_lane<NJ, false, false, LinFastNodeLds<NJ, false, CHAIN>, LinFastShared<NJ, false>, C>(md, sh, nl, 0, lb, path, g, nl.x, vq, ujg, e, kin);
    kh = lane_pick6(e.fh, g);
    kq = e.vg;
  };

  // the observer: one time point (t, x, u(t, x)).  Every lane of the WAVE calls it (publish / controller synchronise the wave's LDS traffic);
  // only the problems with `mine` write
  int n_rec = 0;
  auto record = [&](double ts, double vh, double vq, bool mine) {
    publish(vh, vq);
    controller(ts);
    if (mine) {
      if (valid && n_rec < a.rec_cap) {
        const size_t at = (size_t)ob * a.rec_cap + n_rec;
        if (g == 0) a.rec_t[at] = ts;
        if (g < 6) a.rec_x[at * NX + g] = vh;
        if (g < G) a.rec_x[at * NX + 6 + g] = vq;
        for (int r = g; r < NU; r += LPN) a.rec_u[at * NU + r] = nl.u[r];
      }
      ++n_rec;
    }
  };
  double xh = g < 6 ? a.x_start[(size_t)b * NX + g] : 0.0, xq = g < G ? a.x_start[(size_t)b * NX + 6 + g] : 0.0;
  double k1h = 0.0, k1q = 0.0;
  double t = t0, dt = a.time_step, seg_end = t0;
  int seg = -1;                        // index of the current interval; -1: none opened yet
  bool done = !valid || status != 0, have_k1 = false, fresh = true, opened = false;
  int accepted = 0, rejected = 0, failed_in_a_row = 0;
  bool stepped = false;
  constexpr double kEps = 2.220446049250313e-16, kWeakEps = 1e-6;
  for (;;) {
    // ---- integrate_adaptive bookkeeping: open the next interval when the current one is exhausted
    for (int guard = 0; guard < kRolloutMaxEvents + 2 && !done && !(seg >= 0 && seg_end - t > kEps); ++guard) {
      if (seg == n_events) { done = true; break; }            // the interval ending at tf is finished
      ++seg;
      const double begin = seg == 0 ? t0 : w.event[sub][seg - 1];
      seg_end = seg == n_events ? tf : w.event[sub][seg];
      const double nudged = begin + kWeakEps;
      t = nudged < seg_end ? nudged : seg_end;
      dt = a.time_step;
      have_k1 = false;                                          // a new controlled stepper: m_first_call
      fresh = true;
      opened = true;
    }
    if (a.rec_t && __any(opened)) { record(t, xh, xq, opened && !done); opened = false; }
    if (!__any(!done)) break;
    if (fresh && !done && (t + dt) - seg_end > kEps) dt = seg_end - t;
    fresh = false;
    // ---- try_step
    if (__any(!done && !have_k1)) {
      double fh, fq;
      deriv(t, xh, xq, fh, fq);
      if (!have_k1) { k1h = fh; k1q = fq; have_k1 = true; }
    }
    double k2h, k2q, k3h, k3q, k4h, k4q, k5h, k5q, k6h, k6q, k7h, k7q;
    deriv(t + dt * (1.0 / 5.0), xh + dt * (1.0 / 5.0) * k1h, xq + dt * (1.0 / 5.0) * k1q, k2h, k2q);
    deriv(t + dt * (3.0 / 10.0), xh + dt * (3.0 / 40.0 * k1h + 9.0 / 40.0 * k2h), xq + dt * (3.0 / 40.0 * k1q + 9.0 / 40.0 * k2q), k3h, k3q);
    deriv(t + dt * (4.0 / 5.0), xh + dt * (44.0 / 45.0 * k1h - 56.0 / 15.0 * k2h + 32.0 / 9.0 * k3h),
          xq + dt * (44.0 / 45.0 * k1q - 56.0 / 15.0 * k2q + 32.0 / 9.0 * k3q), k4h, k4q);
    deriv(t + dt * (8.0 / 9.0), xh + dt * (19372.0 / 6561.0 * k1h - 25360.0 / 2187.0 * k2h + 64448.0 / 6561.0 * k3h - 212.0 / 729.0 * k4h),
          xq + dt * (19372.0 / 6561.0 * k1q - 25360.0 / 2187.0 * k2q + 64448.0 / 6561.0 * k3q - 212.0 / 729.0 * k4q), k5h, k5q);
    deriv(t + dt, xh + dt * (9017.0 / 3168.0 * k1h - 355.0 / 33.0 * k2h + 46732.0 / 5247.0 * k3h + 49.0 / 176.0 * k4h - 5103.0 / 18656.0 * k5h),
          xq + dt * (9017.0 / 3168.0 * k1q - 355.0 / 33.0 * k2q + 46732.0 / 5247.0 * k3q + 49.0 / 176.0 * k4q - 5103.0 / 18656.0 * k5q), k6h, k6q);
    const double nh = xh + dt * (35.0 / 384.0 * k1h + 500.0 / 1113.0 * k3h + 125.0 / 192.0 * k4h - 2187.0 / 6784.0 * k5h + 11.0 / 84.0 * k6h);
    const double nq = xq + dt * (35.0 / 384.0 * k1q + 500.0 / 1113.0 * k3q + 125.0 / 192.0 * k4q - 2187.0 / 6784.0 * k5q + 11.0 / 84.0 * k6q);
    deriv(t + dt, nh, nq, k7h, k7q);
    constexpr double d1 = 35.0 / 384.0 - 5179.0 / 57600.0, d3 = 500.0 / 1113.0 - 7571.0 / 16695.0, d4 = 125.0 / 192.0 - 393.0 / 640.0,
                     d5 = -2187.0 / 6784.0 + 92097.0 / 339200.0, d6 = 11.0 / 84.0 - 187.0 / 2100.0, d7 = -1.0 / 40.0;
    const double eh = dt * (d1 * k1h + d3 * k3h + d4 * k4h + d5 * k5h + d6 * k6h + d7 * k7h);
    const double eq = dt * (d1 * k1q + d3 * k3q + d4 * k4q + d5 * k5q + d6 * k6q + d7 * k7q);
    double err = 0.0;
    if (g < 6) err = fabs(eh) / (a.abs_tol + a.rel_tol * (fabs(xh) + fabs(dt) * fabs(k1h)));
    if (g < G) err = fmax(err, fabs(eq) / (a.abs_tol + a.rel_tol * (fabs(xq) + fabs(dt) * fabs(k1q))));
    err = node_allreduce_max<LPN>(err);
    if (!done) {
      if (err > 1.0) {                                          // reject: decrease_step (error order 4)
        dt *= fmax(0.9 * pow(err, -1.0 / 3.0), 0.2);
        ++rejected;
        if (++failed_in_a_row > 500) { status = 2; done = true; }
      } else {                                                  // accept: increase_step (stepper order 5)
        t += dt;
        xh = nh; xq = nq; k1h = k7h; k1q = k7q;
        if (err < 0.5) dt *= 0.9 * pow(fmax(3.2e-4, err), -1.0 / 5.0);   // 5^-5 = 3.2e-4
        ++accepted;
        failed_in_a_row = 0;
        fresh = true;
        stepped = true;
        if (accepted > a.max_steps) { status = 1; done = true; }
      }
    }
    if (a.rec_t && __any(stepped)) { record(t, xh, xq, stepped); stepped = false; }
    if (a.rec_t && !done && n_rec > a.rec_cap) { status = 4; done = true; }      // the record is full: this roll-out cannot become a solution, its wave need not wait for it
  }
  // inputTrajectory.back() = computeInput(final time, final state)
  publish(xh, xq);
  controller(t);
  if (valid) {
    if (g < 6) a.x_end[(size_t)ob * NX + g] = xh;
    if (g < G) a.x_end[(size_t)ob * NX + 6 + g] = xq;
    for (int r = g; r < NU; r += LPN) a.u_end[(size_t)ob * NU + r] = nl.u[r];
    if (g == 0) {
      a.steps[2 * ob] = accepted; a.steps[2 * ob + 1] = rejected;
      if (a.rec_t) { a.rec_n[ob] = n_rec < a.rec_cap ? n_rec : a.rec_cap; if (status == 0 && n_rec > a.rec_cap) status = 4; }
      a.status[ob] = status;
    }
  }
}

}  // namespace bpmpc
