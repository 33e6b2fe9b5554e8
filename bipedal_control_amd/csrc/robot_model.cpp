#include "robot_model.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <set>
#include <stdexcept>

#include "info_tree.h"
#include "urdf_tree.h"

namespace bpmpc {

int mode_from_string(const std::string& s) {
  if (s == "LF") return LF;
  if (s == "RF") return RF;
  if (s == "STANCE") return STANCE;
  return FLY;  // "FLY" and, like std::map::operator[] in string2ModeNumber, anything unknown
}

namespace {

struct Placement { double R[9]; double p[3]; };

Placement identity() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; }
void mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  std::copy(t, t + 9, C);
}
void apply(const double* R, const double* v, double* out) {
  const double a = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], b = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], c = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  out[0] = a; out[1] = b; out[2] = c;
}
Placement compose(const Placement& a, const double rpy[3], const double xyz[3]) {
  Placement r;
  double Rj[9], t[3];
  rpy_to_matrix(rpy, Rj);
  mul(a.R, Rj, r.R);
  apply(a.R, xyz, t);
  for (int i = 0; i < 3; ++i) r.p[i] = a.p[i] + t[i];
  return r;
}

// inertia accumulator of one movable body, expressed in that body's joint frame about its origin
struct BodyAccumulator {
  double m = 0, mc[3] = {0, 0, 0}, Io[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  void add(const UrdfLink& l, const Placement& T) {
    double c[3], t[3];
    apply(T.R, l.com, t);
    for (int i = 0; i < 3; ++i) c[i] = T.p[i] + t[i];
    double RI[9], RIRt[9], Rt[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rt[3 * i + j] = T.R[3 * j + i];
    mul(T.R, l.inertia, RI);
    mul(RI, Rt, RIRt);
    const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Io[3 * i + j] += RIRt[3 * i + j] + l.mass * ((i == j ? cc : 0.0) - c[i] * c[j]);
    m += l.mass;
    for (int i = 0; i < 3; ++i) mc[i] += l.mass * c[i];
  }
};

struct TreeBuilder {
  const UrdfRobot& robot;
  const std::set<std::string>& actuated;
  RobotModel& out;
  std::map<std::string, const UrdfLink*> link_by_name;
  std::map<std::string, std::vector<const UrdfJoint*>> joints_of_parent;  // sorted by joint name (urdfdom stores joints in a std::map)
  std::vector<BodyAccumulator> acc;
  std::map<std::string, std::pair<int, Placement>> frame_of_link;

  void descend(const std::string& link, int body, const Placement& T) {
    auto lit = link_by_name.find(link);
    if (lit == link_by_name.end()) throw std::runtime_error("URDF: joint references unknown link " + link);
    acc[body].add(*lit->second, T);
    frame_of_link[link] = {body, T};
    for (const UrdfJoint* j : joints_of_parent[link]) {
      const Placement Tj = compose(T, j->rpy, j->xyz);
      const bool moves = actuated.count(j->name) > 0;
      if (moves) {
        if (j->type != "revolute" && j->type != "continuous") throw std::runtime_error("joint " + j->name + ": only revolute joints can be actuated");
        if (out.nj >= kMaxJoints) throw std::runtime_error("too many actuated joints");
        const int idx = ++out.nj;
        out.parent[idx] = body;
        std::copy(Tj.R, Tj.R + 9, out.Rfix[idx]);
        std::copy(Tj.p, Tj.p + 3, out.pfix[idx]);
        const double n = std::sqrt(j->axis[0] * j->axis[0] + j->axis[1] * j->axis[1] + j->axis[2] * j->axis[2]);
        for (int i = 0; i < 3; ++i) out.axis[idx][i] = j->axis[i] / n;
        out.joint_names.push_back(j->name);
        acc.emplace_back();
        descend(j->child, idx, identity());
      } else {
        descend(j->child, body, Tj);  // welded at its zero position (fixed joints and joints outside jointNames)
      }
    }
  }
};

}  // namespace

void contact_points(const RobotModel& m, const double* q, double pos[kNumContacts][3], double* jac) {
  double R[kMaxBodies][9], o[kMaxBodies][3], ahat[kMaxBodies][3];
  const double cy = std::cos(q[3]), sy = std::sin(q[3]), cp = std::cos(q[4]), sp = std::sin(q[4]), cr = std::cos(q[5]), sr = std::sin(q[5]);
  const double R0[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr};
  std::copy(R0, R0 + 9, R[0]);
  std::copy(q, q + 3, o[0]);
  for (int j = 1; j <= m.nj; ++j) {
    const double* a = m.axis[j];
    const double c = std::cos(q[5 + j]), s = std::sin(q[5 + j]), v = 1.0 - c;
    const double rot[9] = {c + v * a[0] * a[0], v * a[0] * a[1] - s * a[2], v * a[0] * a[2] + s * a[1],
                           v * a[1] * a[0] + s * a[2], c + v * a[1] * a[1], v * a[1] * a[2] - s * a[0],
                           v * a[2] * a[0] - s * a[1], v * a[2] * a[1] + s * a[0], c + v * a[2] * a[2]};
    double E[9], t[3];
    mul(m.Rfix[j], rot, E);
    mul(R[m.parent[j]], E, R[j]);
    apply(R[m.parent[j]], m.pfix[j], t);
    for (int i = 0; i < 3; ++i) o[j][i] = o[m.parent[j]][i] + t[i];
    apply(R[j], a, ahat[j]);
  }
  for (int c = 0; c < kNumContacts; ++c) {
    const int b = m.contact_body[c];
    double t[3];
    apply(R[b], m.contact_off[c], t);
    for (int i = 0; i < 3; ++i) pos[c][i] = o[b][i] + t[i];
    if (!jac) continue;
    for (int j = 1; j <= m.nj; ++j) {
      bool on_path = false;
      for (int k = b; k > 0; k = m.parent[k]) on_path |= (k == j);
      const double r[3] = {pos[c][0] - o[j][0], pos[c][1] - o[j][1], pos[c][2] - o[j][2]};
      const double* w = ahat[j];
      const double col[3] = {w[1] * r[2] - w[2] * r[1], w[2] * r[0] - w[0] * r[2], w[0] * r[1] - w[1] * r[0]};
      for (int i = 0; i < 3; ++i) jac[(3 * c + i) * m.nj + (j - 1)] = on_path ? col[i] : 0.0;
    }
  }
}

RobotModel load_robot_model(const std::string& urdf_path, const std::string& task_info, const std::string& reference_info) {
  const auto task = read_info_file(task_info);
  const auto ref = read_info_file(reference_info);
  const UrdfRobot robot = read_urdf_file(urdf_path);

  RobotModel m;
  const std::vector<std::string> joint_names = load_string_list(*task, "model_settings.jointNames");
  m.contact_names = load_string_list(*task, "model_settings.contactNames3DoF");
  if (joint_names.empty()) throw std::runtime_error("task.info: model_settings.jointNames is empty");
  if (m.contact_names.size() != kNumContacts) throw std::runtime_error("task.info: exactly 4 contactNames3DoF are supported");
  if (!load_string_list(*task, "model_settings.contactNames6DoF").empty()) throw std::runtime_error("task.info: 6-DoF contacts are not supported");
  const std::set<std::string> actuated(joint_names.begin(), joint_names.end());

  TreeBuilder tb{robot, actuated, m, {}, {}, {}, {}};
  std::set<std::string> children;
  for (const UrdfLink& l : robot.links) tb.link_by_name[l.name] = &l;
  for (const UrdfJoint& j : robot.joints) { tb.joints_of_parent[j.parent].push_back(&j); children.insert(j.child); }
  for (auto& kv : tb.joints_of_parent) std::sort(kv.second.begin(), kv.second.end(), [](const UrdfJoint* a, const UrdfJoint* b) { return a->name < b->name; });
  std::string root;
  for (const UrdfLink& l : robot.links)
    if (!children.count(l.name)) {
      if (!root.empty()) throw std::runtime_error("URDF: more than one root link (" + root + ", " + l.name + ")");
      root = l.name;
    }
  if (root.empty()) throw std::runtime_error("URDF: no root link");
  tb.acc.emplace_back();
  tb.descend(root, 0, identity());
  if (m.nj != static_cast<int>(joint_names.size())) throw std::runtime_error("task.info: some jointNames are not revolute joints of the URDF");
  m.parent[0] = -1;
  m.nx = m.nu = 12 + m.nj;
  for (int b = 0; b <= m.nj; ++b) {
    const BodyAccumulator& a = tb.acc[b];
    m.mass[b] = a.m;
    if (a.m > 0) {
      double c[3] = {a.mc[0] / a.m, a.mc[1] / a.m, a.mc[2] / a.m};
      const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      for (int i = 0; i < 3; ++i) {
        m.com[b][i] = c[i];
        for (int j = 0; j < 3; ++j) m.inertia[b][3 * i + j] = a.Io[3 * i + j] - a.m * ((i == j ? cc : 0.0) - c[i] * c[j]);
      }
    }
  }
  m.robot_mass = 0;
  for (const UrdfLink& l : robot.links) m.robot_mass += l.mass;
  for (int c = 0; c < kNumContacts; ++c) {
    auto it = tb.frame_of_link.find(m.contact_names[c]);
    if (it == tb.frame_of_link.end()) throw std::runtime_error("contact frame " + m.contact_names[c] + " not found in URDF");
    m.contact_body[c] = it->second.first;
    std::copy(it->second.second.p, it->second.second.p + 3, m.contact_off[c]);
  }

  // scalars and vectors
  m.initial_state = load_matrix(*task, "initialState", m.nx, 1);
  m.default_joint_state = load_matrix(*ref, "defaultJointState", m.nj, 1);
  auto need = [](const InfoNode& n, const std::string& key, double* out) { if (!n.get(key, out)) throw std::runtime_error("INFO: missing " + key); };
  need(*ref, "comHeight", &m.com_height);
  need(*ref, "targetDisplacementVelocity", &m.target_displacement_velocity);
  need(*ref, "targetRotationVelocity", &m.target_rotation_velocity);
  need(*task, "frictionConeSoftConstraint.frictionCoefficient", &m.friction_coefficient);
  need(*task, "frictionConeSoftConstraint.mu", &m.barrier_mu);
  need(*task, "frictionConeSoftConstraint.delta", &m.barrier_delta);
  need(*task, "model_settings.positionErrorGain", &m.position_error_gain);
  need(*task, "model_settings.phaseTransitionStanceTime", &m.phase_transition_stance_time);
  need(*task, "swing_trajectory_config.liftOffVelocity", &m.swing.lift_off_velocity);
  need(*task, "swing_trajectory_config.touchDownVelocity", &m.swing.touch_down_velocity);
  need(*task, "swing_trajectory_config.swingHeight", &m.swing.swing_height);
  need(*task, "swing_trajectory_config.swingTimeScale", &m.swing.swing_time_scale);
  need(*task, "sqp.dt", &m.sqp.dt);
  task->get("sqp.sqpIteration", &m.sqp.sqp_iteration);
  task->get("sqp.deltaTol", &m.sqp.delta_tol);
  task->get("sqp.g_max", &m.sqp.g_max);
  task->get("sqp.g_min", &m.sqp.g_min);
  {
    // what the reference hands to SqpMpc as sqp::Settings (BipedalController.cpp:303-306, BipedalRobotInterface.cpp:99) and this engine
    // does not implement is refused, not ignored: a drop-in must not run a different optimiser silently
    std::string v;
    if (task->get("sqp.integratorType", &v) && v != "RK2")
      throw UnsupportedSetting("task.info: sqp.integratorType " + v + " is not implemented (RK2 sensitivities only)");
    if (task->get("sqp.projectStateInputEqualityConstraints", &v) && !(v == "true" || v == "1"))
      throw UnsupportedSetting("task.info: sqp.projectStateInputEqualityConstraints " + v + " is not implemented (the equality constraints are always projected)");
    if (task->get("sqp.useFeedbackPolicy", &v)) m.sqp.use_feedback_policy = (v == "true" || v == "1") ? 1 : 0;
  }
  task->get("sqp.inequalityConstraintMu", &m.sqp_inequality_mu);
  task->get("sqp.inequalityConstraintDelta", &m.sqp_inequality_delta);
  task->get("mpc.timeHorizon", &m.time_horizon);
  task->get("mpc.mrtDesiredFrequency", &m.mrt_frequency);
  task->get("mpc.mpcDesiredFrequency", &m.mpc_frequency);
  // ipm / ddp blocks: optional, as every ocs2 loadSettings (missing entries keep the defaults)
  {
    auto flag = [&](const std::string& path, int* out) { std::string v; if (task->get(path, &v)) *out = (v == "true" || v == "1") ? 1 : 0; };
    auto choice = [&](const std::string& path, std::initializer_list<const char*> names, int* out) {
      std::string v;
      if (!task->get(path, &v)) return;
      int i = 0;
      for (const char* n : names) { if (v == n) { *out = i; return; } ++i; }
      throw std::runtime_error("task.info: unknown value '" + v + "' of " + path);
    };
    IpmConfig& p = m.ipm;
    task->get("ipm.dt", &p.dt); task->get("ipm.ipmIteration", &p.ipm_iteration); task->get("ipm.deltaTol", &p.delta_tol);
    task->get("ipm.g_max", &p.g_max); task->get("ipm.g_min", &p.g_min); task->get("ipm.nThreads", &p.n_threads); task->get("ipm.threadPriority", &p.thread_priority);
    flag("ipm.computeLagrangeMultipliers", &p.compute_lagrange_multipliers); flag("ipm.useFeedbackPolicy", &p.use_feedback_policy);
    task->get("ipm.initialBarrierParameter", &p.initial_barrier_parameter); task->get("ipm.targetBarrierParameter", &p.target_barrier_parameter);
    task->get("ipm.barrierLinearDecreaseFactor", &p.barrier_linear_decrease_factor); task->get("ipm.barrierSuperlinearDecreasePower", &p.barrier_superlinear_decrease_power);
    task->get("ipm.barrierReductionCostTol", &p.barrier_reduction_cost_tol); task->get("ipm.barrierReductionConstraintTol", &p.barrier_reduction_constraint_tol);
    task->get("ipm.fractionToBoundaryMargin", &p.fraction_to_boundary_margin); flag("ipm.usePrimalStepSizeForDual", &p.use_primal_step_size_for_dual);
    task->get("ipm.initialSlackLowerBound", &p.initial_slack_lower_bound); task->get("ipm.initialDualLowerBound", &p.initial_dual_lower_bound);
    task->get("ipm.initialSlackMarginRate", &p.initial_slack_margin_rate); task->get("ipm.initialDualMarginRate", &p.initial_dual_margin_rate);
    DdpConfig& d = m.ddp;
    choice("ddp.algorithm", {"SLQ", "ILQR"}, &d.algorithm);
    task->get("ddp.nThreads", &d.n_threads); task->get("ddp.threadPriority", &d.thread_priority); task->get("ddp.maxNumIterations", &d.max_num_iterations);
    task->get("ddp.minRelCost", &d.min_rel_cost); task->get("ddp.constraintTolerance", &d.constraint_tolerance);
    task->get("ddp.AbsTolODE", &d.abs_tol_ode); task->get("ddp.RelTolODE", &d.rel_tol_ode); task->get("ddp.timeStep", &d.time_step);
    task->get("ddp.maxNumStepsPerSecond", &d.max_num_steps_per_second);
    choice("ddp.backwardPassIntegratorType", {"ODE45", "EULER", "ODE45_OCS2", "ADAMS_BASHFORTH", "BULIRSCH_STOER", "MODIFIED_MIDPOINT", "RK4", "RK5_VARIABLE", "ADAMS_BASHFORTH_MOULTON"},
           &d.backward_pass_integrator);
    task->get("ddp.constraintPenaltyInitialValue", &d.constraint_penalty_initial_value); task->get("ddp.constraintPenaltyIncreaseRate", &d.constraint_penalty_increase_rate);
    flag("ddp.preComputeRiccatiTerms", &d.pre_compute_riccati_terms); flag("ddp.useFeedbackPolicy", &d.use_feedback_policy);
    choice("ddp.strategy", {"LINE_SEARCH", "LEVENBERG_MARQUARDT"}, &d.strategy);
    task->get("ddp.lineSearch.minStepLength", &d.ls_min_step_length); task->get("ddp.lineSearch.maxStepLength", &d.ls_max_step_length);
    choice("ddp.lineSearch.hessianCorrectionStrategy", {"DIAGONAL_SHIFT", "CHOLESKY_MODIFICATION", "EIGENVALUE_MODIFICATION", "GERSHGORIN_MODIFICATION"},
           &d.ls_hessian_correction_strategy);
    task->get("ddp.lineSearch.hessianCorrectionMultiple", &d.ls_hessian_correction_multiple);
  }
  task->get("rollout.AbsTolODE", &m.rollout.abs_tol);
  task->get("rollout.RelTolODE", &m.rollout.rel_tol);
  task->get("rollout.timeStep", &m.rollout.time_step);
  task->get("rollout.maxNumStepsPerSecond", &m.rollout.max_steps_per_second);

  // cost weights: Q as given; R = blkdiag(R_task[forces], J^T R_task[feet] J) with J the contact-point Jacobians
  // w.r.t. the leg joints at initialState (BipedalRobotInterface.cpp:239-271)
  m.Q = load_matrix(*task, "Q", m.nx, m.nx);
  const int tc = 3 * kNumContacts;
  const std::vector<double> Rt = load_matrix(*task, "R", 2 * tc, 2 * tc);
  std::vector<double> J(static_cast<size_t>(tc) * m.nj);
  double pos[kNumContacts][3];
  contact_points(m, m.initial_state.data() + 6, pos, J.data());
  m.R.assign(static_cast<size_t>(m.nu) * m.nu, 0.0);
  for (int i = 0; i < tc; ++i)
    for (int j = 0; j < tc; ++j) m.R[i * m.nu + j] = Rt[i * 2 * tc + j];
  std::vector<double> RJ(static_cast<size_t>(tc) * m.nj);
  for (int i = 0; i < tc; ++i)
    for (int j = 0; j < m.nj; ++j) {
      double t = 0;
      for (int l = 0; l < tc; ++l) t += Rt[(tc + i) * 2 * tc + tc + l] * J[l * m.nj + j];
      RJ[i * m.nj + j] = t;
    }
  for (int i = 0; i < m.nj; ++i)
    for (int j = 0; j < m.nj; ++j) {
      double t = 0;
      for (int l = 0; l < tc; ++l) t += J[l * m.nj + i] * RJ[l * m.nj + j];
      m.R[(tc + i) * m.nu + tc + j] = t;
    }

  // gait bootstrap (BipedalRobotInterface.cpp:209-234, ModeSequenceTemplate.cpp:50-111)
  m.initial_mode_schedule.event_times = load_scalar_list(*ref, "initialModeSchedule.eventTimes");
  for (const std::string& s : load_string_list(*ref, "initialModeSchedule.modeSequence")) m.initial_mode_schedule.modes.push_back(mode_from_string(s));
  if (m.initial_mode_schedule.modes.empty()) throw std::runtime_error("reference.info: failed to load initialModeSchedule");
  m.default_template.switching_times = load_scalar_list(*ref, "defaultModeSequenceTemplate.switchingTimes");
  for (const std::string& s : load_string_list(*ref, "defaultModeSequenceTemplate.modeSequence")) m.default_template.modes.push_back(mode_from_string(s));
  if (m.default_template.switching_times.empty() || m.default_template.modes.empty()) throw std::runtime_error("reference.info: failed to load defaultModeSequenceTemplate");
  return m;
}

ModeTemplate load_mode_template(const std::string& gait_info, const std::string& name) {
  const auto g = read_info_file(gait_info);
  ModeTemplate t;
  t.switching_times = load_scalar_list(*g, name + ".switchingTimes");
  for (const std::string& s : load_string_list(*g, name + ".modeSequence")) t.modes.push_back(mode_from_string(s));
  if (t.switching_times.empty() || t.modes.empty()) throw std::runtime_error("[loadModeSequenceTemplate] failed to load : " + name + " from " + gait_info);
  return t;
}

}  // namespace bpmpc
