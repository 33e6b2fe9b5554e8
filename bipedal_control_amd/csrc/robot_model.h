// Host-side model constants of the MPC problem: what BipedalRobotInterface assembles at construction
// (ocs2_bipedal_robot/src/BipedalRobotInterface.cpp:67-204) reduced to plain arrays the kernels consume.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

namespace bpmpc {

constexpr int kMaxJoints = 12;            // leg joints (H1: 10, G1 / OpenLoong class: 12)
constexpr int kMaxBodies = kMaxJoints + 1;
constexpr int kMaxState = 12 + kMaxJoints;
constexpr int kNumContacts = 4;           // 3-DoF contact points (two per foot), task.info contactNames3DoF
constexpr int kMaxEqRows = 16;            // FLY: 4 x (3 zero-force + 1 normal-velocity)

// mode ids: include/ocs2_bipedal_robot/gait/MotionPhaseDefinition.h:47-52
enum Mode { FLY = 0, LF = 1, RF = 2, STANCE = 3 };
int mode_from_string(const std::string& s);  // unknown names map to 0 like the reference's std::map lookup

struct ModeSchedule {
  std::vector<double> event_times;
  std::vector<int> modes;  // event_times.size() + 1
};
struct ModeTemplate {
  std::vector<double> switching_times;
  std::vector<int> modes;  // switching_times.size() - 1
};

struct SwingConfig { double lift_off_velocity = 0, touch_down_velocity = 0, swing_height = 0.1, swing_time_scale = 0.15; };
// A task.info that asks for a variant of the reference's solver this engine does not implement (load_robot_model): BPMPC_ERR_UNSUPPORTED
struct UnsupportedSetting : std::runtime_error { using std::runtime_error::runtime_error; };

struct SqpConfig { double dt = 0.015; int sqp_iteration = 1; double delta_tol = 1e-4, g_max = 1e-2, g_min = 1e-6;
                   // keys that select the ARITHMETIC of the solver (task.info:76,80,81): the engine implements integratorType RK2 and
                   // projectStateInputEqualityConstraints true only and refuses anything else; both values of useFeedbackPolicy
                   // (LinearController / FeedforwardController of the solution, of the warm start and of the policy rollout)
                   int use_feedback_policy = 1;
                   // [OCS2-upstream] sqp::Settings defaults not present in task.info
                   double alpha_decay = 0.5, alpha_min = 1e-4, gamma_c = 1e-6, armijo_factor = 1e-4, cost_tol = 1e-4; };

// rollout block of task.info (TimeTriggeredRollout, ODE45)
struct RolloutConfig { double abs_tol = 1e-5, rel_tol = 1e-3, time_step = 0.015; int max_steps_per_second = 10000; };

// ipm block of task.info: the reference LOADS these settings (src/BipedalRobotInterface.cpp:100, accessor ipmSettings(),
// include/ocs2_bipedal_robot/BipedalRobotInterface.h:80) and constructs no IPM solver anywhere; the same here - loaded, exposed, unused.
// Defaults [OCS2-upstream, recalled]: ocs2_ipm/IpmSettings.h.
struct IpmConfig {
  double dt = 0.01; int ipm_iteration = 10; double delta_tol = 1e-6, g_max = 1e-2, g_min = 1e-6;
  int compute_lagrange_multipliers = 1, use_feedback_policy = 1, n_threads = 4, thread_priority = 50;
  double initial_barrier_parameter = 1e-2, target_barrier_parameter = 1e-4, barrier_linear_decrease_factor = 0.2,
         barrier_superlinear_decrease_power = 1.5, barrier_reduction_cost_tol = 1e-3, barrier_reduction_constraint_tol = 1e-3;
  double fraction_to_boundary_margin = 0.995; int use_primal_step_size_for_dual = 1;
  double initial_slack_lower_bound = 1e-4, initial_dual_lower_bound = 1e-4, initial_slack_margin_rate = 1e-2, initial_dual_margin_rate = 1e-2;
};
// ddp block of task.info: loaded at src/BipedalRobotInterface.cpp:98 and consumed by the stand-alone DDP node
// (ocs2_bipedal_robot_ros/src/BipedalRobotDdpMpcNode.cpp:70-74, GaussNewtonDDP_MPC) - a solver this engine does not have.
struct DdpConfig {
  int algorithm = 0;                       // 0 SLQ, 1 ILQR
  int n_threads = 1, thread_priority = 99, max_num_iterations = 15;
  double min_rel_cost = 1e-3, constraint_tolerance = 1e-3;
  double abs_tol_ode = 1e-9, rel_tol_ode = 1e-6, time_step = 1e-2; int max_num_steps_per_second = 10000;
  int backward_pass_integrator = 0;        // 0 ODE45 (index into kIntegratorNames)
  double constraint_penalty_initial_value = 2.0, constraint_penalty_increase_rate = 2.0;
  int pre_compute_riccati_terms = 1, use_feedback_policy = 0;
  int strategy = 0;                        // 0 LINE_SEARCH, 1 LEVENBERG_MARQUARDT
  double ls_min_step_length = 0.05, ls_max_step_length = 1.0; int ls_hessian_correction_strategy = 0;   // 0 DIAGONAL_SHIFT, 1 CHOLESKY_MODIFICATION, 2 EIGENVALUE_MODIFICATION, 3 GERSHGORIN_MODIFICATION
  double ls_hessian_correction_multiple = 1e-6;
};

struct RobotModel {
  int nj = 0, nx = 0, nu = 0;
  std::vector<std::string> joint_names, contact_names;
  // kinematic tree: body 0 = floating base (welded links merged), joint j (1..nj) moves body j
  int parent[kMaxBodies] = {};
  double Rfix[kMaxBodies][9] = {}, pfix[kMaxBodies][3] = {}, axis[kMaxBodies][3] = {};
  double mass[kMaxBodies] = {}, com[kMaxBodies][3] = {}, inertia[kMaxBodies][9] = {};
  int contact_body[kNumContacts] = {};
  double contact_off[kNumContacts][3] = {};
  double robot_mass = 0;
  std::vector<double> Q, R, initial_state, default_joint_state;
  double com_height = 0, target_displacement_velocity = 0, target_rotation_velocity = 0;
  double friction_coefficient = 0.7, cone_regularization = 25.0, cone_gripper_force = 0.0, cone_hessian_shift = 1e-6;
  double barrier_mu = 0.1, barrier_delta = 5.0;
  // BipedalRobotInterface(..., useHardFrictionConeConstraint) (BipedalRobotInterface.cpp:68-69,181-182): the cone is an inequality
  // constraint of the problem instead of a soft constraint; the SQP solver penalises it with sqp.inequalityConstraintMu / Delta
  // (task.info:74-75; [OCS2-upstream] defaults 0 - no penalty - and 1e-6)
  bool hard_friction_cone = false;
  double sqp_inequality_mu = 0.0, sqp_inequality_delta = 1e-6;
  double position_error_gain = 0, phase_transition_stance_time = 0;
  double time_horizon = 1.0;
  SwingConfig swing;
  SqpConfig sqp;
  RolloutConfig rollout;
  IpmConfig ipm;
  DdpConfig ddp;
  double mrt_frequency = 400, mpc_frequency = 50;
  ModeSchedule initial_mode_schedule;
  ModeTemplate default_template;
};

// Throws std::runtime_error (bad files, unsupported joints, dimension overflow).
RobotModel load_robot_model(const std::string& urdf_path, const std::string& task_info, const std::string& reference_info);
ModeTemplate load_mode_template(const std::string& gait_info, const std::string& name);

// World positions of the contact points and their 3 x nj Jacobians w.r.t. the leg joints at configuration
// q = [p(3), zyx(3), joints(nj)] (host double precision; used for the input-cost matrix and by tests).
void contact_points(const RobotModel& m, const double* q, double pos[kNumContacts][3], double* jac_joints /*12 x nj or null*/);

}  // namespace bpmpc
