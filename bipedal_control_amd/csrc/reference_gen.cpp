#include "reference_gen.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

namespace bpmpc {

int phase_index(const std::vector<double>& event_times, double t) {
  return static_cast<int>(std::lower_bound(event_times.begin(), event_times.end(), t) - event_times.begin());
}

// ---------------------------------------------------------------- gait schedule
GaitSchedule::GaitSchedule(ModeSchedule initial, ModeTemplate tmpl, double phase_transition_stance_time)
    : schedule_(std::move(initial)), template_(std::move(tmpl)), stance_time_(phase_transition_stance_time) {}

void GaitSchedule::insert_template(const ModeTemplate& tmpl, double start_time, double final_time) {
  template_ = tmpl;
  auto& ev = schedule_.event_times;
  auto& ms = schedule_.modes;
  const size_t cut = phase_index(ev, start_time);
  if (cut < ev.size()) {
    ev.resize(cut);
    ms.resize(cut + 1);
  }
  double transition = stance_time_;
  if (!ms.empty() && ms.back() == STANCE) transition = 0.0;
  if (transition > 0.0) {
    ev.push_back(start_time);
    ms.push_back(STANCE);
  }
  tile(start_time + transition, final_time);
}

const ModeSchedule& GaitSchedule::mode_schedule(double lower, double upper) {
  auto& ev = schedule_.event_times;
  auto& ms = schedule_.modes;
  const int first = phase_index(ev, lower);
  if (first > 0) {  // forget the past but keep one phase, relabelled STANCE, in front of `lower`
    ev.erase(ev.begin(), ev.begin() + (first - 1));
    ms.erase(ms.begin(), ms.begin() + (first - 1));
    ms.front() = STANCE;
  }
  const double resume = ev.empty() ? upper : ev.back();
  if (!ev.empty()) ev.pop_back();
  if (!ms.empty()) ms.pop_back();  // the trailing STANCE placeholder
  tile(resume, upper);
  return schedule_;
}

void GaitSchedule::tile(double start_time, double final_time) {
  auto& ev = schedule_.event_times;
  auto& ms = schedule_.modes;
  const size_t phases = template_.modes.size();
  if (phases == 0) return;
  if (!ev.empty() && start_time <= ev.back()) throw std::runtime_error("The initial time for template-tiling is not greater than the last event time.");
  ev.push_back(start_time);
  while (ev.back() < final_time)
    for (size_t i = 0; i < phases; ++i) {
      ms.push_back(template_.modes[i]);
      ev.push_back(ev.back() + (template_.switching_times[i + 1] - template_.switching_times[i]));
    }
  ms.push_back(STANCE);
}

// ---------------------------------------------------------------- swing height profiles
HeightSegment HeightSegment::through(double ta, double za, double va, double tb, double zb, double vb) {
  HeightSegment s;
  s.t0 = ta;
  s.dt = tb - ta;
  const double dp = zb - za, dv = vb - va;
  s.c0 = 0.0 * s.dt + za;
  s.c1 = va * s.dt;
  s.c2 = -(3.0 * va + dv) * s.dt + 3.0 * dp;
  s.c3 = (2.0 * va + dv) * s.dt - 2.0 * dp;
  return s;
}
double HeightSegment::position(double t) const {
  const double tn = (t - t0) / dt;
  return c3 * tn * tn * tn + c2 * tn * tn + c1 * tn + c0;
}
double HeightSegment::velocity(double t) const {
  const double tn = (t - t0) / dt;
  return (3.0 * c3 * tn * tn + 2.0 * c2 * tn + c1) / dt;
}

namespace {
SwingProfile make_profile(double t_lo, double z_lo, double v_lo, double z_mid, double t_td, double z_td, double v_td) {
  SwingProfile p;
  p.mid_time = (t_lo + t_td) / 2;
  p.up = HeightSegment::through(t_lo, z_lo, v_lo, p.mid_time, z_mid, 0.0);
  p.down = HeightSegment::through(p.mid_time, z_mid, 0.0, t_td, z_td, v_td);
  return p;
}
}  // namespace

void SwingPlanner::update(const ModeSchedule& schedule, double terrain) {
  const int phases = static_cast<int>(schedule.modes.size());
  for (int c = 0; c < kNumContacts; ++c) {
    profiles_[c].clear();
    profiles_[c].reserve(phases);
    for (int p = 0; p < phases; ++p) {
      if (contact_flag(schedule.modes[p], c)) {
        profiles_[c].push_back(make_profile(0.0, terrain, 0.0, terrain, 1.0, terrain, 0.0));
        continue;
      }
      int lift = -1, land = phases - 1;
      for (int i = p - 1; i >= 0; --i)
        if (contact_flag(schedule.modes[i], c)) { lift = i; break; }
      for (int i = p + 1; i < phases; ++i)
        if (contact_flag(schedule.modes[i], c)) { land = i - 1; break; }
      if (lift < 0) throw std::runtime_error("The time of take-off for the first swing of the EE with ID " + std::to_string(c) + " is not defined.");
      if (land >= phases - 1) throw std::runtime_error("The time of touch-down for the last swing of the EE with ID " + std::to_string(c) + " is not defined.");
      const double t_lo = schedule.event_times[lift], t_td = schedule.event_times[land];
      const double scale = std::min(1.0, (t_td - t_lo) / cfg_.swing_time_scale);
      profiles_[c].push_back(make_profile(t_lo, terrain, scale * cfg_.lift_off_velocity, std::min(terrain, terrain) + scale * cfg_.swing_height, t_td,
                                          terrain, scale * cfg_.touch_down_velocity));
    }
  }
  events_ = schedule.event_times;
}
double SwingPlanner::z_position(int c, double t) const { return profiles_[c][phase_index(events_, t)].position(t); }
double SwingPlanner::z_velocity(int c, double t) const { return profiles_[c][phase_index(events_, t)].velocity(t); }

// ---------------------------------------------------------------- shooting grid
std::vector<GridNode> shooting_grid(double t0, double tf, double dt, const std::vector<double>& events, double dt_min) {
  if (!(dt > 0) || !(tf > t0)) throw std::runtime_error("shooting_grid: need dt > 0 and tf > t0");
  std::vector<GridNode> grid{{t0, kNone}};
  size_t next_event = phase_index(events, t0);
  double t = t0;
  while (grid.back().time < tf) {
    t = t + dt;
    int ev = kNone;
    if (next_event < events.size() && t >= events[next_event]) {
      t = events[next_event++];
      ev = kPreEvent;
    }
    if (t >= tf) {
      t = tf;
      ev = kNone;
    }
    if (t > grid.back().time + dt_min) grid.push_back({t, ev});
    else grid.back() = {t, ev};  // nodes closer than dt_min merge
    if (ev == kPreEvent) grid.push_back({t, kPostEvent});
  }
  return grid;
}
double interval_start(const GridNode& n) { return n.event == kPostEvent ? n.time + 1e-6 : n.time; }
double interval_end(const GridNode& n) { return n.event == kPreEvent ? n.time - 1e-6 : n.time; }

NodeTable build_node_table(const RobotModel& m, double t0, double tf, double dt, const ModeSchedule& schedule, const SwingPlanner& planner) {
  (void)m;
  const std::vector<GridNode> grid = shooting_grid(t0, tf, dt, schedule.event_times);
  NodeTable tab;
  tab.N = static_cast<int>(grid.size()) - 1;
  tab.node_time.resize(grid.size());
  for (size_t i = 0; i < grid.size(); ++i) tab.node_time[i] = grid[i].time;
  tab.kind.assign(tab.N, 0); tab.start.assign(tab.N, 0.0); tab.dt.assign(tab.N, 0.0); tab.mode.assign(tab.N, STANCE);
  tab.zref.assign(4 * tab.N, 0.0); tab.zdref.assign(4 * tab.N, 0.0);
  for (int k = 0; k < tab.N; ++k) {
    if (grid[k].event == kPreEvent) {
      tab.kind[k] = 1;
      tab.start[k] = grid[k].time;
      tab.mode[k] = mode_at(schedule, grid[k].time);
      continue;
    }
    const double ts = interval_start(grid[k]);
    tab.start[k] = ts;
    tab.dt[k] = interval_end(grid[k + 1]) - ts;
    tab.mode[k] = mode_at(schedule, ts);
    for (int c = 0; c < kNumContacts; ++c) {
      tab.zref[4 * k + c] = planner.z_position(c, ts);
      tab.zdref[4 * k + c] = planner.z_velocity(c, ts);
    }
  }
  return tab;
}

// ---------------------------------------------------------------- targets
namespace {
void pose_targets(const RobotModel& m, const double target_pose[6], double t_now, const double* x_now, double t_reach, double times[2], double* xs) {
  const int nx = m.nx;
  std::fill(xs, xs + 2 * nx, 0.0);
  times[0] = t_now;
  times[1] = t_reach;
  for (int i = 0; i < 6; ++i) { xs[6 + i] = x_now[6 + i]; xs[nx + 6 + i] = target_pose[i]; }
  xs[6 + 2] = m.com_height;
  xs[6 + 4] = 0.0;
  xs[6 + 5] = 0.0;
  for (int j = 0; j < m.nj; ++j) xs[12 + j] = xs[nx + 12 + j] = m.default_joint_state[j];
}
}  // namespace

void cmd_vel_to_targets(const RobotModel& m, const double cmd[4], double t_now, const double* x, double T, double times[2], double* xs) {
  const double z = x[9], y = x[10], r = x[11];
  const double cz = std::cos(z), sz = std::sin(z), cy = std::cos(y), sy = std::sin(y), cx = std::cos(r), sx = std::sin(r);
  const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx, sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx, -sy, cy * sx, cy * cx};
  double v[3];
  for (int i = 0; i < 3; ++i) v[i] = R[3 * i] * cmd[0] + R[3 * i + 1] * cmd[1] + R[3 * i + 2] * cmd[2];
  const double pose[6] = {x[6] + v[0] * T, x[7] + v[1] * T, m.com_height, x[9] + cmd[3] * T, 0.0, 0.0};
  pose_targets(m, pose, t_now, x, t_now + T, times, xs);
  for (int i = 0; i < 3; ++i) xs[i] = xs[m.nx + i] = v[i];
}

void goal_to_targets(const RobotModel& m, const double goal[4], double t_now, const double* x, double times[2], double* xs) {
  const double pose[6] = {goal[0], goal[1], m.com_height, goal[3], 0.0, 0.0};
  const double dx = pose[0] - x[6], dy = pose[1] - x[7], dyaw = pose[3] - x[9];
  const double reach = std::max(std::abs(dyaw) / m.target_rotation_velocity, std::sqrt(dx * dx + dy * dy) / m.target_displacement_velocity);
  pose_targets(m, pose, t_now, x, t_now + reach, times, xs);
}

void interpolate_targets(int n, const double* times, const double* xs, int nx, double t, double* out) {
  if (n == 1 || t <= times[0]) { std::copy(xs, xs + nx, out); return; }
  if (t >= times[n - 1]) { std::copy(xs + (n - 1) * nx, xs + n * nx, out); return; }
  const int i = static_cast<int>(std::lower_bound(times, times + n, t) - times) - 1;
  const double alpha = (times[i + 1] - t) / (times[i + 1] - times[i]);
  for (int j = 0; j < nx; ++j) out[j] = alpha * xs[i * nx + j] + (1.0 - alpha) * xs[(i + 1) * nx + j];
}

}  // namespace bpmpc
