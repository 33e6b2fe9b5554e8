// Device-side reference generation: for every distinct (t0, gait, gait start) of a batch the
// mode schedule (template tiling), the shooting grid with event nodes, the node table and the swing-height references are
// built by one workgroup in LDS; per problem the two-point target trajectory of a velocity command.  The arithmetic repeats
// the host pre-pass (reference_gen.cpp, which restates GaitSchedule.cpp:40-137, SwingTrajectoryPlanner.cpp / SplineCpg.cpp /
// CubicSpline.cpp, timeDiscretizationWithEvents and TargetTrajectoriesPublisher.cpp:30-62) operation by operation with
// floating-point contraction switched off, so the tables are bit-identical to the host path (tests/test_gpu_reference_gen.py);
// only sin/cos of the targets may differ in the last place.
#pragma once
#include <hip/hip_runtime.h>

#include "../robot_model.h"
#include "linesearch.h"

namespace bpmpc {

constexpr int kRefMaxEvents = 448;     // capacity of one tiled mode schedule
constexpr int kRefLibCapacity = 4096;   // doubles / ints of the uploaded gait library
constexpr int kRefMaxGrid = 520;       // >= kMaxRiccatiStages + 2 grid points
enum RefStatus { kRefOk = 0, kRefTileOrder = 1, kRefCapacity = 2, kRefGridTooLong = 3, kRefNoTakeOff = 8 /* + contact */, kRefNoTouchDown = 16 /* + contact */ };

struct GaitLibraryView {
  const double* switching;   // switching times of all templates, concatenated
  const int* modes;          // modes of all templates, concatenated
  const int* first_mode;     // [n_templates + 1]: template g owns modes[first_mode[g] .. first_mode[g+1]) and switching[first_mode[g] + g ..]
  int n_templates;           // the last entry is defaultModeSequenceTemplate of reference.info
  const double* init_events; // initialModeSchedule of reference.info
  const int* init_modes;
  int init_n_events;
  double transition_stance_time;
};

struct ReferenceGenArgs {
  GaitLibraryView lib;
  int n_grids, N;                       // N = node stride of the tables (max_nodes)
  double horizon, dt, dt_min;
  const double* t0;                     // per grid
  const int* gait;                      // per grid: template index, < 0 = no template inserted (initial schedule only)
  const double* gait_start;             // per grid: time the template was inserted
  double lift_off_velocity, touch_down_velocity, swing_height, swing_time_scale;
  int *kind, *mode, *nodes, *status, *rows;
  double *gdt, *gstart, *zref, *zdref, *node_time;
};

struct RefGenLds {
  double ev[kRefMaxEvents];
  int ms[kRefMaxEvents + 1];
  double time[kRefMaxGrid];
  int event[kRefMaxGrid];
  int base, ne, nm, n_grid, status, rows, vrows;
};

struct DevSchedule {
  double* ev;
  int* ms;
  int ne, nm, cap, status;
  double keep_from;      // events more than one phase in front of this time are never looked at again (GaitSchedule::getModeSchedule drops them)
};

#define EXACT_FP_BODY _Pragma("clang fp contract(off)")   // first statement of a body: no fused multiply-add, as on the host

__device__ inline int ref_lower_bound(const double* a, int n, double t) {   // first index with a[i] >= t
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < t) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ inline bool ref_contact(int mode, int c) { return c < 2 ? (mode == 1 || mode == 3) : (mode == 2 || mode == 3); }
// A long-running gait is tiled from its insertion time, so the schedule grows with t0; what lies more than one phase in front of the
// window is dropped on the way (exactly what getModeSchedule would erase at the end), the event times themselves are the same
// repeated additions.
__device__ inline bool ref_compact(DevSchedule& s) {
  const int drop = ref_lower_bound(s.ev, s.ne, s.keep_from) - 1;
  if (drop <= 0) { s.status = kRefCapacity; return false; }
  for (int i = 0; i + drop < s.ne; ++i) s.ev[i] = s.ev[i + drop];
  for (int i = 0; i + drop < s.nm; ++i) s.ms[i] = s.ms[i + drop];
  s.ne -= drop;
  s.nm -= drop;
  return true;
}
__device__ inline void ref_push_ev(DevSchedule& s, double t) { if (s.ne < s.cap || ref_compact(s)) s.ev[s.ne++] = t; }
__device__ inline void ref_push_ms(DevSchedule& s, int m) { if (s.nm < s.cap + 1 || ref_compact(s)) s.ms[s.nm++] = m; }

__device__ inline void ref_tile(DevSchedule& s, const double* sw, const int* modes, int phases, double start, double final_time) {
  EXACT_FP_BODY
  if (phases == 0) return;
  if (s.ne > 0 && start <= s.ev[s.ne - 1]) { s.status = kRefTileOrder; return; }
  ref_push_ev(s, start);
  while (s.status == kRefOk && s.ev[s.ne - 1] < final_time)
    for (int i = 0; i < phases; ++i) {
      ref_push_ms(s, modes[i]);
      ref_push_ev(s, s.ev[s.ne - 1] + (sw[i + 1] - sw[i]));
    }
  ref_push_ms(s, 3);
}

// GaitSchedule(initial, default template).insert_template(gait, start, t0 + 2 H) [if gait >= 0] followed by
// mode_schedule(t0 - H, t0 + 2 H): what SwitchedModelReferenceManager::modifyReferences asks for at solve time.
__device__ inline void ref_build_schedule(RefGenLds& w, const GaitLibraryView& lib, int gait, double start, double t0, double horizon) {
  EXACT_FP_BODY
  DevSchedule s{w.ev, w.ms, 0, 0, kRefMaxEvents, kRefOk, t0 - horizon};
  for (int i = 0; i < lib.init_n_events; ++i) ref_push_ev(s, lib.init_events[i]);
  for (int i = 0; i <= lib.init_n_events; ++i) ref_push_ms(s, lib.init_modes[i]);
  int tmpl = lib.n_templates - 1;
  const double upper = t0 + 2 * horizon, lower = t0 - horizon;
  if (gait >= 0) {
    tmpl = gait;
    const int cut = ref_lower_bound(s.ev, s.ne, start);
    if (cut < s.ne) { s.ne = cut; s.nm = cut + 1; }
    double transition = lib.transition_stance_time;
    if (s.nm > 0 && s.ms[s.nm - 1] == 3) transition = 0.0;
    if (transition > 0.0) { ref_push_ev(s, start); ref_push_ms(s, 3); }
    const int f = lib.first_mode[tmpl];
    ref_tile(s, lib.switching + f + tmpl, lib.modes + f, lib.first_mode[tmpl + 1] - f, start + transition, upper);
  }
  int base = 0;
  if (s.status == kRefOk) {
    const int first = ref_lower_bound(s.ev, s.ne, lower);
    if (first > 0) {                     // forget the past but keep one phase, relabelled STANCE, in front of `lower`
      base = first - 1;
      s.ev += base; s.ms += base; s.ne -= base; s.nm -= base; s.cap -= base;
      s.ms[0] = 3;
    }
    const double resume = s.ne == 0 ? upper : s.ev[s.ne - 1];
    if (s.ne > 0) --s.ne;
    if (s.nm > 0) --s.nm;
    const int f = lib.first_mode[tmpl];
    ref_tile(s, lib.switching + f + tmpl, lib.modes + f, lib.first_mode[tmpl + 1] - f, resume, upper);
  }
  // SwingTrajectoryPlanner::update: every swing phase needs a stance phase of that leg before and after it
  if (s.status == kRefOk)
    for (int c = 0; c < 4 && s.status == kRefOk; ++c) {
      int fc = -1, lc = -1;
      for (int p = 0; p < s.nm; ++p)
        if (ref_contact(s.ms[p], c)) { if (fc < 0) fc = p; lc = p; }
      if (s.nm > 0 && fc != 0) s.status = kRefNoTakeOff + c;
      else if (s.nm > 0 && lc != s.nm - 1) s.status = kRefNoTouchDown + c;
    }
  w.base = base; w.ne = s.ne; w.nm = s.nm; w.status = s.status;
}

// timeDiscretizationWithEvents: dt steps from t0, event times inserted as a pre-event / post-event pair, nodes closer than
// dt_min merged.  Returns the number of intervals (which may exceed what was stored; the caller reports that).
__device__ inline int ref_shooting_grid(RefGenLds& w, double t0, double tf, double dt, double dt_min) {
  EXACT_FP_BODY
  const double* ev = w.ev + w.base;
  int cnt = 1;
  double last_t = t0;
  w.time[0] = t0; w.event[0] = 0;
  int next_event = ref_lower_bound(ev, w.ne, t0);
  double t = t0;
  auto store = [&](int i, double tt, int e) { if (i < kRefMaxGrid) { w.time[i] = tt; w.event[i] = e; } };
  while (last_t < tf && cnt < 4 * kRefMaxGrid) {
    t = t + dt;
    int e = 0;
    if (next_event < w.ne && t >= ev[next_event]) { t = ev[next_event++]; e = 1; }
    if (t >= tf) { t = tf; e = 0; }
    if (t > last_t + dt_min) { store(cnt, t, e); ++cnt; }
    else store(cnt - 1, t, e);
    last_t = t;
    if (e == 1) { store(cnt, t, 2); ++cnt; }
  }
  return cnt - 1;
}

struct RefCubic { double t0, dt, c0, c1, c2, c3; };
__device__ inline RefCubic ref_cubic(double ta, double za, double va, double tb, double zb, double vb) {
  EXACT_FP_BODY
  RefCubic s;
  s.t0 = ta;
  s.dt = tb - ta;
  const double dp = zb - za, dv = vb - va;
  s.c0 = 0.0 * s.dt + za;
  s.c1 = va * s.dt;
  s.c2 = -(3.0 * va + dv) * s.dt + 3.0 * dp;
  s.c3 = (2.0 * va + dv) * s.dt - 2.0 * dp;
  return s;
}
__device__ inline void ref_cubic_eval(const RefCubic& s, double t, double* z, double* zd) {
  EXACT_FP_BODY
  const double tn = (t - s.t0) / s.dt;
  *z = s.c3 * tn * tn * tn + s.c2 * tn * tn + s.c1 * tn + s.c0;
  *zd = (3.0 * s.c3 * tn * tn + 2.0 * s.c2 * tn + s.c1) / s.dt;
}

__device__ inline void ref_swing(const RefGenLds& w, const ReferenceGenArgs& a, int c, int p, double t, double* z, double* zd) {
  EXACT_FP_BODY
  const double* ev = w.ev + w.base;
  const int* ms = w.ms + w.base;
  const double terrain = 0.0;
  if (ref_contact(ms[p], c)) { *z = terrain; *zd = 0.0; return; }
  int lift = 0, land = w.nm - 2;
  for (int i = p - 1; i >= 0; --i)
    if (ref_contact(ms[i], c)) { lift = i; break; }
  for (int i = p + 1; i < w.nm; ++i)
    if (ref_contact(ms[i], c)) { land = i - 1; break; }
  const double t_lo = ev[lift], t_td = ev[land];
  const double ratio = (t_td - t_lo) / a.swing_time_scale;
  const double scale = 1.0 < ratio ? 1.0 : ratio;
  const double mid = (t_lo + t_td) / 2;
  const double z_mid = terrain + scale * a.swing_height;
  if (t < mid) ref_cubic_eval(ref_cubic(t_lo, terrain, scale * a.lift_off_velocity, mid, z_mid, 0.0), t, z, zd);
  else ref_cubic_eval(ref_cubic(mid, z_mid, 0.0, t_td, terrain, scale * a.touch_down_velocity), t, z, zd);
}

__global__ __launch_bounds__(64) void k_reference_grids(ReferenceGenArgs a) {
  EXACT_FP_BODY
  __shared__ RefGenLds w;
  const int g = blockIdx.x, l = threadIdx.x;
  const double t0 = a.t0[g];
  if (l == 0) {
    w.rows = 12;
    w.vrows = 4;
    ref_build_schedule(w, a.lib, a.gait[g], a.gait_start[g], t0, a.horizon);
    int n = 0;
    if (w.status == kRefOk) {
      n = ref_shooting_grid(w, t0, t0 + a.horizon, a.dt, a.dt_min);
      if (n > a.N) w.status = kRefGridTooLong;
    }
    w.n_grid = n;
  }
  __syncthreads();
  const int n = w.n_grid, N = a.N;
  const bool ok = w.status == kRefOk;
  const double* ev = w.ev + w.base;
  const int* ms = w.ms + w.base;
  for (int k = l; k < N; k += 64) {
    int kind = 0, mode = 3;
    double start = 0.0, dt = 0.0, z[4] = {0.0, 0.0, 0.0, 0.0}, zd[4] = {0.0, 0.0, 0.0, 0.0};
    if (ok && k < n) {
      if (w.event[k] == 1) {
        kind = 1;
        start = w.time[k];
        mode = ms[ref_lower_bound(ev, w.ne, start)];
      } else {
        start = w.event[k] == 2 ? w.time[k] + 1e-6 : w.time[k];
        const double end = w.event[k + 1] == 1 ? w.time[k + 1] - 1e-6 : w.time[k + 1];
        dt = end - start;
        const int p = ref_lower_bound(ev, w.ne, start);
        mode = ms[p];
        for (int c = 0; c < 4; ++c) ref_swing(w, a, c, p, start, &z[c], &zd[c]);
        const int left = (mode == 1 || mode == 3) ? 3 : 4, right = (mode == 2 || mode == 3) ? 3 : 4;
        atomicMax(&w.rows, 2 * left + 2 * right);
        atomicMax(&w.vrows, (left == 3 ? 6 : 2) + (right == 3 ? 6 : 2));   // rows that constrain a contact velocity (structured projection)
      }
    }
    const size_t i = (size_t)g * N + k;
    a.kind[i] = kind; a.mode[i] = mode; a.gstart[i] = start; a.gdt[i] = dt;
    for (int c = 0; c < 4; ++c) { a.zref[4 * i + c] = z[c]; a.zdref[4 * i + c] = zd[c]; }
  }
  for (int k = l; k <= N; k += 64) a.node_time[(size_t)g * (N + 1) + k] = (ok && k <= n) ? w.time[k] : 0.0;
  __syncthreads();
  if (l == 0) { a.nodes[g] = n; a.status[g] = w.status; a.rows[g] = w.rows | (w.vrows << 8); }
}

struct CommandTargetArgs {
  int batch, nx, nj;
  int goal;                // 0: cmd = (vx, vy, vz, yaw rate) in the base frame; 1: cmd = goal pose (x, y, unused, yaw)
  double time_to_target, com_height, displacement_velocity, rotation_velocity;
  double default_joint_state[kMaxJoints];
  const double* t0;        // per problem
  const double* x0;        // per problem [nx]
  const double* cmd_vel;   // per problem [4]: vx, vy, vz (base frame), yaw rate
  double* tgt_t;           // [batch][kMaxTargetPoints]
  double* tgt_x;           // [batch][kMaxTargetPoints][nx]
  int* tgt_n;
};

// cmdVelToTargetTrajectories (TargetTrajectoriesPublisher.cpp:40-62 restated in reference_gen.cpp cmd_vel_to_targets)
__global__ __launch_bounds__(64) void k_command_targets(CommandTargetArgs a) {
  EXACT_FP_BODY
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.batch) return;
  const int nx = a.nx;
  const double* x = a.x0 + (size_t)b * nx;
  const double* cmd = a.cmd_vel + (size_t)b * 4;
  const double T = a.time_to_target, t_now = a.t0[b];
  double v[3] = {0.0, 0.0, 0.0}, pose[6], t_reach;
  if (a.goal) {            // goalToTargetTrajectories (TargetTrajectoriesPublisher.cpp:64-99 restated in reference_gen.cpp goal_to_targets)
    pose[0] = cmd[0]; pose[1] = cmd[1]; pose[2] = a.com_height; pose[3] = cmd[3]; pose[4] = 0.0; pose[5] = 0.0;
    const double dx = pose[0] - x[6], dy = pose[1] - x[7], dyaw = pose[3] - x[9];
    const double tr = fabs(dyaw) / a.rotation_velocity, td = sqrt(dx * dx + dy * dy) / a.displacement_velocity;
    t_reach = t_now + (tr < td ? td : tr);
  } else {
    const double yz = x[9], yy = x[10], yx = x[11];
    const double cz = cos(yz), sz = sin(yz), cy = cos(yy), sy = sin(yy), cx = cos(yx), sx = sin(yx);
    const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx, sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx, -sy, cy * sx, cy * cx};
    for (int i = 0; i < 3; ++i) v[i] = R[3 * i] * cmd[0] + R[3 * i + 1] * cmd[1] + R[3 * i + 2] * cmd[2];
    pose[0] = x[6] + v[0] * T; pose[1] = x[7] + v[1] * T; pose[2] = a.com_height; pose[3] = x[9] + cmd[3] * T; pose[4] = 0.0; pose[5] = 0.0;
    t_reach = t_now + T;
  }
  double* ts = a.tgt_t + (size_t)b * kMaxTargetPoints;
  double* xs = a.tgt_x + (size_t)b * kMaxTargetPoints * nx;
  for (int i = 0; i < 2 * nx; ++i) xs[i] = 0.0;
  ts[0] = t_now;
  ts[1] = t_reach;
  for (int i = 0; i < 6; ++i) { xs[6 + i] = x[6 + i]; xs[nx + 6 + i] = pose[i]; }
  xs[6 + 2] = a.com_height;
  xs[6 + 4] = 0.0;
  xs[6 + 5] = 0.0;
  for (int j = 0; j < a.nj; ++j) xs[12 + j] = xs[nx + 12 + j] = a.default_joint_state[j];
  if (!a.goal) for (int i = 0; i < 3; ++i) xs[i] = xs[nx + i] = v[i];
  a.tgt_n[b] = 2;
}

#undef EXACT_FP_BODY

}  // namespace bpmpc
