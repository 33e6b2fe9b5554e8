// Host-only part of the C ABI (include/bpmpc.h): model ingest and the reference-manager pre-pass.
#include <algorithm>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "capi_internal.h"
#include "reference_gen.h"

struct bpmpc_model { bpmpc::RobotModel rm; };
struct bpmpc_gait { std::unique_ptr<bpmpc::GaitSchedule> schedule; };

namespace bpmpc {
namespace {
thread_local std::string g_last_error;
}
void set_last_error(const std::string& message) { g_last_error = message; }
const RobotModel& model_of(const bpmpc_model* handle) { return handle->rm; }
}  // namespace bpmpc

using namespace bpmpc;

namespace {
int fail(int code, const std::string& why) { set_last_error(why); return code; }

template <typename F>
int guarded(F&& body) {
  try {
    return body();
  } catch (const std::length_error& e) {
    return fail(BPMPC_ERR_CAPACITY, e.what());
  } catch (const std::invalid_argument& e) {
    return fail(BPMPC_ERR_INVALID_ARGUMENT, e.what());
  } catch (const UnsupportedSetting& e) {
    return fail(BPMPC_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) {
    return fail(BPMPC_ERR_IO, e.what());
  }
}

int copy_out(const double* src, size_t n, double* out, int capacity) {
  if ((size_t)capacity < n) throw std::length_error("output capacity too small");
  std::copy(src, src + n, out);
  return (int)n;
}
}  // namespace

extern "C" {

const char* bpmpc_last_error(void) { return g_last_error.c_str(); }
const char* bpmpc_version(void) { return "bpmpc 0.1 (gfx950, fp64)"; }

int bpmpc_model_create(const char* urdf, const char* task, const char* reference, bpmpc_model** out) {
  if (!urdf || !task || !reference || !out) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_model_create: null argument");
  *out = nullptr;
  return guarded([&] {
    auto m = std::make_unique<bpmpc_model>();
    m->rm = load_robot_model(urdf, task, reference);
    *out = m.release();
    return (int)BPMPC_OK;
  });
}
int bpmpc_model_create_ex(const char* urdf, const char* task, const char* reference, int use_hard_friction_cone, bpmpc_model** out) {
  const int rc = bpmpc_model_create(urdf, task, reference, out);
  if (rc == BPMPC_OK) (*out)->rm.hard_friction_cone = use_hard_friction_cone != 0;
  return rc;
}
void bpmpc_model_destroy(bpmpc_model* m) { delete m; }

int bpmpc_model_dims(const bpmpc_model* m, int* nx, int* nu, int* n_contacts, int* n_joints) {
  if (!m) return fail(BPMPC_ERR_INVALID_ARGUMENT, "null model");
  if (nx) *nx = m->rm.nx;
  if (nu) *nu = m->rm.nu;
  if (n_contacts) *n_contacts = kNumContacts;
  if (n_joints) *n_joints = m->rm.nj;
  return BPMPC_OK;
}

int bpmpc_model_get(const bpmpc_model* m, const char* name, double* out, int capacity) {
  if (!m || !name || !out) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_model_get: null argument");
  return guarded([&]() -> int {
    const RobotModel& r = m->rm;
    const std::string n(name);
    const int nj = r.nj, nb = nj + 1;
    auto scalar = [&](double v) { return copy_out(&v, 1, out, capacity); };
    auto list = [&](std::initializer_list<double> v) { std::vector<double> t(v); return copy_out(t.data(), t.size(), out, capacity); };
    if (n == "initial_state") return copy_out(r.initial_state.data(), r.initial_state.size(), out, capacity);
    if (n == "default_joint_state") return copy_out(r.default_joint_state.data(), r.default_joint_state.size(), out, capacity);
    if (n == "Q") return copy_out(r.Q.data(), r.Q.size(), out, capacity);
    if (n == "R") return copy_out(r.R.data(), r.R.size(), out, capacity);
    if (n == "robot_mass") return scalar(r.robot_mass);
    if (n == "com_height") return scalar(r.com_height);
    if (n == "time_horizon") return scalar(r.time_horizon);
    if (n == "position_error_gain") return scalar(r.position_error_gain);
    if (n == "phase_transition_stance_time") return scalar(r.phase_transition_stance_time);
    if (n == "body_mass") return copy_out(r.mass, nb, out, capacity);
    if (n == "body_com") return copy_out(&r.com[0][0], 3 * nb, out, capacity);
    if (n == "body_inertia") return copy_out(&r.inertia[0][0], 9 * nb, out, capacity);
    if (n == "joint_rotation") return copy_out(&r.Rfix[1][0], 9 * nj, out, capacity);
    if (n == "joint_offset") return copy_out(&r.pfix[1][0], 3 * nj, out, capacity);
    if (n == "joint_axis") return copy_out(&r.axis[1][0], 3 * nj, out, capacity);
    if (n == "contact_offset") return copy_out(&r.contact_off[0][0], 3 * kNumContacts, out, capacity);
    if (n == "joint_parent") { std::vector<double> t(nj); for (int j = 0; j < nj; ++j) t[j] = r.parent[j + 1]; return copy_out(t.data(), nj, out, capacity); }
    if (n == "contact_body") { std::vector<double> t(kNumContacts); for (int c = 0; c < kNumContacts; ++c) t[c] = r.contact_body[c]; return copy_out(t.data(), t.size(), out, capacity); }
    if (n == "cone") return list({r.friction_coefficient, r.cone_regularization, r.cone_gripper_force, r.cone_hessian_shift, r.barrier_mu, r.barrier_delta});
    if (n == "hard_cone") return list({r.hard_friction_cone ? 1.0 : 0.0, r.sqp_inequality_mu, r.sqp_inequality_delta});
    if (n == "swing") return list({r.swing.lift_off_velocity, r.swing.touch_down_velocity, r.swing.swing_height, r.swing.swing_time_scale});
    if (n == "rollout") return list({r.rollout.abs_tol, r.rollout.rel_tol, r.rollout.time_step, (double)r.rollout.max_steps_per_second, r.mrt_frequency, r.mpc_frequency});
    if (n == "sqp") return list({r.sqp.dt, (double)r.sqp.sqp_iteration, r.sqp.delta_tol, r.sqp.g_max, r.sqp.g_min, (double)r.sqp.use_feedback_policy,
                                 1.0 /* projectStateInputEqualityConstraints */, 0.0 /* integratorType: 0 = RK2 */});
    // settings blocks the reference loads beside sqp (BipedalRobotInterface.cpp:98-100); field order documented in include/bpmpc.h
    if (n == "ipm") {
      const IpmConfig& p = r.ipm;
      return list({p.dt, (double)p.ipm_iteration, p.delta_tol, p.g_max, p.g_min, (double)p.compute_lagrange_multipliers, (double)p.use_feedback_policy,
                   p.initial_barrier_parameter, p.target_barrier_parameter, p.barrier_linear_decrease_factor, p.barrier_superlinear_decrease_power,
                   p.barrier_reduction_cost_tol, p.barrier_reduction_constraint_tol, p.fraction_to_boundary_margin, (double)p.use_primal_step_size_for_dual,
                   p.initial_slack_lower_bound, p.initial_dual_lower_bound, p.initial_slack_margin_rate, p.initial_dual_margin_rate,
                   (double)p.n_threads, (double)p.thread_priority});
    }
    if (n == "ddp") {
      const DdpConfig& d = r.ddp;
      return list({(double)d.algorithm, (double)d.max_num_iterations, d.min_rel_cost, d.constraint_tolerance, d.abs_tol_ode, d.rel_tol_ode, d.time_step,
                   (double)d.max_num_steps_per_second, (double)d.backward_pass_integrator, d.constraint_penalty_initial_value, d.constraint_penalty_increase_rate,
                   (double)d.pre_compute_riccati_terms, (double)d.use_feedback_policy, (double)d.strategy, d.ls_min_step_length, d.ls_max_step_length,
                   (double)d.ls_hessian_correction_strategy, d.ls_hessian_correction_multiple, (double)d.n_threads, (double)d.thread_priority});
    }
    throw std::invalid_argument("bpmpc_model_get: unknown name " + n);
  });
}

int bpmpc_model_joint_name(const bpmpc_model* m, int j, char* out, int capacity) {
  if (!m || !out || j < 0 || j >= m->rm.nj) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_model_joint_name: bad argument");
  const std::string& s = m->rm.joint_names[j];
  if ((int)s.size() + 1 > capacity) return fail(BPMPC_ERR_CAPACITY, "name buffer too small");
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

int bpmpc_gait_create(const bpmpc_model* m, bpmpc_gait** out) {
  if (!m || !out) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_gait_create: null argument");
  return guarded([&] {
    auto g = std::make_unique<bpmpc_gait>();
    g->schedule = std::make_unique<GaitSchedule>(m->rm.initial_mode_schedule, m->rm.default_template, m->rm.phase_transition_stance_time);
    *out = g.release();
    return (int)BPMPC_OK;
  });
}
void bpmpc_gait_destroy(bpmpc_gait* g) { delete g; }

int bpmpc_gait_load_template(const char* path, const char* name, double* switching_times, int* modes, int capacity, int* n_modes) {
  if (!path || !name || !switching_times || !modes || !n_modes) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_gait_load_template: null argument");
  return guarded([&] {
    const ModeTemplate t = load_mode_template(path, name);
    if (t.switching_times.size() != t.modes.size() + 1) throw std::runtime_error("gait template needs one more switching time than modes");
    if ((int)t.modes.size() + 1 > capacity) throw std::length_error("gait template capacity too small");
    std::copy(t.switching_times.begin(), t.switching_times.end(), switching_times);
    std::copy(t.modes.begin(), t.modes.end(), modes);
    *n_modes = (int)t.modes.size();
    return (int)BPMPC_OK;
  });
}

int bpmpc_gait_insert_template(bpmpc_gait* g, const double* switching_times, const int* modes, int n_modes, double start_time, double final_time) {
  if (!g || !switching_times || !modes || n_modes < 0) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_gait_insert_template: bad argument");
  return guarded([&] {
    ModeTemplate t;
    t.switching_times.assign(switching_times, switching_times + n_modes + 1);
    t.modes.assign(modes, modes + n_modes);
    g->schedule->insert_template(t, start_time, final_time);
    return (int)BPMPC_OK;
  });
}

int bpmpc_gait_mode_schedule(bpmpc_gait* g, double lower, double upper, double* event_times, int* modes, int capacity, int* n_events) {
  if (!g || !event_times || !modes || !n_events) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_gait_mode_schedule: null argument");
  return guarded([&] {
    const ModeSchedule& s = g->schedule->mode_schedule(lower, upper);
    if ((int)s.modes.size() > capacity) throw std::length_error("mode schedule capacity too small");
    std::copy(s.event_times.begin(), s.event_times.end(), event_times);
    std::copy(s.modes.begin(), s.modes.end(), modes);
    *n_events = (int)s.event_times.size();
    return (int)BPMPC_OK;
  });
}

int bpmpc_swing_reference(const bpmpc_model* m, const double* event_times, const int* modes, int n_events, const double* t, int n_t, double* z,
                          double* zdot) {
  if (!m || !modes || !t || !z || !zdot || n_events < 0 || (n_events > 0 && !event_times)) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_swing_reference: bad argument");
  return guarded([&] {
    ModeSchedule s;
    s.event_times.assign(event_times, event_times + n_events);
    s.modes.assign(modes, modes + n_events + 1);
    SwingPlanner planner(m->rm.swing);
    planner.update(s);
    for (int i = 0; i < n_t; ++i)
      for (int c = 0; c < kNumContacts; ++c) {
        z[4 * i + c] = planner.z_position(c, t[i]);
        zdot[4 * i + c] = planner.z_velocity(c, t[i]);
      }
    return (int)BPMPC_OK;
  });
}

int bpmpc_time_grid(double t0, double tf, double dt, const double* event_times, int n_events, double* node_times, int* node_events, int capacity,
                    int* n_nodes) {
  if (!node_times || !node_events || !n_nodes || n_events < 0 || (n_events > 0 && !event_times)) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_time_grid: bad argument");
  return guarded([&] {
    const std::vector<double> ev(event_times, event_times + n_events);
    const std::vector<GridNode> grid = shooting_grid(t0, tf, dt, ev);
    if ((int)grid.size() > capacity) throw std::length_error("time grid capacity too small");
    for (size_t i = 0; i < grid.size(); ++i) { node_times[i] = grid[i].time; node_events[i] = grid[i].event; }
    *n_nodes = (int)grid.size();
    return (int)BPMPC_OK;
  });
}

int bpmpc_cmd_vel_to_targets(const bpmpc_model* m, const double cmd_vel[4], double t_now, const double* x_now, double time_to_target, double* times,
                             double* states) {
  if (!m || !cmd_vel || !x_now || !times || !states) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_cmd_vel_to_targets: null argument");
  return guarded([&] { cmd_vel_to_targets(m->rm, cmd_vel, t_now, x_now, time_to_target, times, states); return (int)BPMPC_OK; });
}
int bpmpc_goal_to_targets(const bpmpc_model* m, const double goal[4], double t_now, const double* x_now, double* times, double* states) {
  if (!m || !goal || !x_now || !times || !states) return fail(BPMPC_ERR_INVALID_ARGUMENT, "bpmpc_goal_to_targets: null argument");
  return guarded([&] { goal_to_targets(m->rm, goal, t_now, x_now, times, states); return (int)BPMPC_OK; });
}

}  // extern "C"
