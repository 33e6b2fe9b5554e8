/* bpmpc - C ABI of the MI355X-native batched NMPC engine (libbpmpc.so).
 *
 * Drop-in boundary for the OCS2 SQP hot path of zitongbai/bipedal_control.  Every entry point
 * names the reference interface it replaces (paths relative to the reference tree).  Plain C: pointers, sizes,
 * ints and doubles only; row-major contiguous host arrays owned by the caller unless a function says "device";
 * every call returns 0 on success or a negative bpmpc_status (never throws across the ABI); a textual reason is
 * available from bpmpc_last_error() (thread local).  A solver handle is not re-entrant (one calling thread per
 * handle, like the reference's single MPC thread, bipedal_controllers/src/BipedalController.cpp:332-351);
 * different handles are independent.  There is NO CPU fallback: creating a solver without a HIP device fails.
 *
 * State / input layout (ocs2_centroidal_model convention used by the reference, task.info:181-210,247-278):
 *   x = [h_lin/m (3), h_ang/m (3), base position (3), yaw, pitch, roll, leg joints (nj)]          nx = 12 + nj
 *   u = [F_0, F_1, F_2, F_3 (world frame, 3 each), leg joint velocities (nj)]                      nu = 12 + nj
 * Mode ids: FLY 0, LF 1, RF 2, STANCE 3 (ocs2_bipedal_robot/include/ocs2_bipedal_robot/gait/MotionPhaseDefinition.h:47-52).
 */
#ifndef BPMPC_H
#define BPMPC_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  BPMPC_OK = 0,
  BPMPC_ERR_INVALID_ARGUMENT = -1,
  BPMPC_ERR_IO = -2,            /* file missing / malformed (the reference throws std::invalid_argument / runtime_error) */
  BPMPC_ERR_UNSUPPORTED = -3,
  BPMPC_ERR_NO_DEVICE = -4,     /* no usable HIP device: the engine has no CPU path */
  BPMPC_ERR_DEVICE = -5,        /* HIP runtime error */
  BPMPC_ERR_CAPACITY = -6,      /* caller buffer or solver capacity too small */
  BPMPC_ERR_NUMERICAL = -7      /* non positive-definite stage Hessian etc. */
} bpmpc_status;

const char* bpmpc_last_error(void);
const char* bpmpc_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * Model = what BipedalRobotInterface builds at construction
 *   ocs2_bipedal_robot/src/BipedalRobotInterface.cpp:67-204 (constructor + setupOptimalConrolProblem),
 *   :239-291 (input-cost matrix), :298-315 (friction-cone settings), src/common/ModelSettings.cpp:40-67.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct bpmpc_model bpmpc_model;

int bpmpc_model_create(const char* urdf_path, const char* task_info_path, const char* reference_info_path, bpmpc_model** out);
/* The same with the fourth constructor argument of the reference, useHardFrictionConeConstraint (BipedalRobotInterface.h:66-69, default
 * false; BipedalRobotInterface.cpp:181-182): the friction cone of every stance contact is an INEQUALITY constraint of the problem instead of
 * a soft constraint.  The SQP solver handles it as [OCS2-upstream, recalled] ocs2_sqp does: the relaxed barrier of task.info's
 * sqp.inequalityConstraintMu / sqp.inequalityConstraintDelta (:74-75) on the LINEAR approximation of the constraint, times dt, added to the
 * stage cost before the projection (value, gradient p'(h) dh/du, Gauss-Newton Hessian p''(h) dh dh'; neither the cone's own second
 * derivative nor its hessianDiagonalShift).  Mu = 0 (the upstream default when the key is absent) means no penalty at all. */
int bpmpc_model_create_ex(const char* urdf_path, const char* task_info_path, const char* reference_info_path, int use_hard_friction_cone,
                          bpmpc_model** out);
void bpmpc_model_destroy(bpmpc_model* model);
/* CentroidalModelInfo.stateDim / inputDim / numThreeDofContacts / actuatedDofNum */
int bpmpc_model_dims(const bpmpc_model* model, int* nx, int* nu, int* n_contacts, int* n_joints);
/* Named constant blocks, copied into out[0..capacity); returns the element count or a negative status.
 * Names: "initial_state" (BipedalRobotInterface::getInitialState), "default_joint_state", "Q", "R", "robot_mass",
 * "com_height", "body_mass", "body_com", "body_inertia", "joint_parent", "joint_rotation", "joint_offset", "joint_axis",
 * "contact_body", "contact_offset", "cone" (mu, regularization, gripper force, hessian shift, barrier mu, barrier delta),
 * "swing" (liftOffVelocity, touchDownVelocity, swingHeight, swingTimeScale), "sqp" (dt, sqpIteration, deltaTol, g_max, g_min,
 * useFeedbackPolicy, projectStateInputEqualityConstraints (always 1), integratorType (always 0 = RK2): bpmpc_model_create returns
 * BPMPC_ERR_UNSUPPORTED with the key in bpmpc_last_error() for a task.info that sets the latter two otherwise),
 * "time_horizon", "position_error_gain", "phase_transition_stance_time", "hard_cone" (flag, sqp.inequalityConstraintMu, Delta),
 * "rollout" (AbsTolODE, RelTolODE, timeStep, maxNumStepsPerSecond, mrt frequency, mpc frequency).
 * The other two solver-settings blocks the reference loads beside `sqp` (src/BipedalRobotInterface.cpp:98-100; accessors ddpSettings(),
 * ipmSettings(), include/ocs2_bipedal_robot/BipedalRobotInterface.h:78-80) are loaded and exposed.  The ipm block has no consumer (the
 * reference constructs no IPM solver either); the ddp block is what bpmpc_settings.solver = BPMPC_SOLVER_DDP runs on (the reference's DDP
 * solver lives in one stand-alone node, BipedalRobotDdpMpcNode.cpp:70-74):
 *   "ipm": dt, ipmIteration, deltaTol, g_max, g_min, computeLagrangeMultipliers, useFeedbackPolicy, initialBarrierParameter,
 *          targetBarrierParameter, barrierLinearDecreaseFactor, barrierSuperlinearDecreasePower, barrierReductionCostTol,
 *          barrierReductionConstraintTol, fractionToBoundaryMargin, usePrimalStepSizeForDual, initialSlackLowerBound,
 *          initialDualLowerBound, initialSlackMarginRate, initialDualMarginRate, nThreads, threadPriority          (booleans as 0 / 1)
 *   "ddp": algorithm (0 SLQ, 1 ILQR), maxNumIterations, minRelCost, constraintTolerance, AbsTolODE, RelTolODE, timeStep,
 *          maxNumStepsPerSecond, backwardPassIntegratorType (0 ODE45, 1 EULER, 2 ODE45_OCS2, 3 ADAMS_BASHFORTH, 4 BULIRSCH_STOER,
 *          5 MODIFIED_MIDPOINT, 6 RK4, 7 RK5_VARIABLE, 8 ADAMS_BASHFORTH_MOULTON), constraintPenaltyInitialValue,
 *          constraintPenaltyIncreaseRate, preComputeRiccatiTerms, useFeedbackPolicy, strategy (0 LINE_SEARCH, 1 LEVENBERG_MARQUARDT),
 *          lineSearch.minStepLength, lineSearch.maxStepLength, lineSearch.hessianCorrectionStrategy (0 DIAGONAL_SHIFT,
 *          1 CHOLESKY_MODIFICATION, 2 EIGENVALUE_MODIFICATION, 3 GERSHGORIN_MODIFICATION), lineSearch.hessianCorrectionMultiple,
 *          nThreads, threadPriority */
int bpmpc_model_get(const bpmpc_model* model, const char* name, double* out, int capacity);
/* joint name j (DFS order = state order); returns length or negative status */
int bpmpc_model_joint_name(const bpmpc_model* model, int j, char* out, int capacity);

/* ---------------------------------------------------------------------------------------------------------------
 * Host pre-pass of a solve = SolverBase::preRun hooks of the reference:
 *   GaitSchedule                          ocs2_bipedal_robot/src/gait/GaitSchedule.cpp:40-137
 *   loadModeSequenceTemplate              ocs2_bipedal_robot/src/gait/ModeSequenceTemplate.cpp:50-71
 *   SwitchedModelReferenceManager         ocs2_bipedal_robot/src/reference_manager/SwitchedModelReferenceManager.cpp:55-69
 *   SwingTrajectoryPlanner                ocs2_bipedal_robot/src/foot_planner/SwingTrajectoryPlanner.cpp:50-219
 *   cmdVel / goal -> TargetTrajectories   bipedal_controllers/src/TargetTrajectoriesPublisher.cpp:30-99
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct bpmpc_gait bpmpc_gait;

/* GaitSchedule(initialModeSchedule, defaultModeSequenceTemplate, phaseTransitionStanceTime) from reference.info */
int bpmpc_gait_create(const bpmpc_model* model, bpmpc_gait** out);
void bpmpc_gait_destroy(bpmpc_gait* gait);
/* loadModeSequenceTemplate(gait.info, name): switching_times[n_modes+1], modes[n_modes] */
int bpmpc_gait_load_template(const char* gait_info_path, const char* name, double* switching_times, int* modes, int capacity, int* n_modes);
/* GaitSchedule::insertModeSequenceTemplate */
int bpmpc_gait_insert_template(bpmpc_gait* gait, const double* switching_times, const int* modes, int n_modes, double start_time,
                               double final_time);
/* GaitSchedule::getModeSchedule(lower, upper) (mutating, like the reference): event_times[n_events], modes[n_events+1] */
int bpmpc_gait_mode_schedule(bpmpc_gait* gait, double lower, double upper, double* event_times, int* modes, int capacity, int* n_events);
/* SwingTrajectoryPlanner::update(schedule, terrain 0) then getZpositionConstraint / getZvelocityConstraint at n_t times:
 * z[n_t*4], zdot[n_t*4] */
int bpmpc_swing_reference(const bpmpc_model* model, const double* event_times, const int* modes, int n_events, const double* t, int n_t,
                          double* z, double* zdot);
/* [OCS2-upstream] timeDiscretizationWithEvents(t0, tf, dt, eventTimes): node_times[n], node_events[n] (0 none, 1 pre, 2 post) */
int bpmpc_time_grid(double t0, double tf, double dt, const double* event_times, int n_events, double* node_times, int* node_events,
                    int capacity, int* n_nodes);
/* cmdVelToTargetTrajectories / goalToTargetTrajectories: two-point trajectory times[2], states[2*nx] */
int bpmpc_cmd_vel_to_targets(const bpmpc_model* model, const double cmd_vel[4], double t_now, const double* x_now, double time_to_target,
                             double* times, double* states);
int bpmpc_goal_to_targets(const bpmpc_model* model, const double goal[4], double t_now, const double* x_now, double* times, double* states);

/* ---------------------------------------------------------------------------------------------------------------
 * Solver = SqpMpc / SqpSolver::runImpl for a BATCH of independent MPC problems on one MI355X
 *   construction sites replaced: bipedal_controllers/src/BipedalController.cpp:303-306,
 *                                ocs2_bipedal_robot_ros/src/BipedalRobotSqpMpcNode.cpp:70
 *   run site replaced:           MPC_MRT_Interface::advanceMpc(), BipedalController.cpp:339
 *   settings:                    task.info:66-83 (sqp), :169-179 (mpc)
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct bpmpc_solver bpmpc_solver;

typedef struct {
  int device;           /* HIP device ordinal */
  int max_batch;        /* capacity in problems */
  int max_nodes;        /* capacity in shooting intervals per problem (event nodes included) */
  int sqp_iterations;   /* <= 0: sqp.sqpIteration of task.info */
  double dt;            /* <= 0: sqp.dt of task.info */
  int return_gains;     /* allocate and compute the feedback gains K (sqp.useFeedbackPolicy) */
  int profile;          /* HIP-event timing (bpmpc_solver_kernel_time): 0 off, 1 every kernel class, 2 the linearisation kernel only */
  void* stream;         /* hipStream_t to run on; NULL = a stream owned by the solver */
  int reference_kernels; /* != 0: run the lane-emulation-verified reference kernel bodies instead of the fast variants (debugging) */
  int pipeline_chunks;  /* > 1: sweep the horizon in that many chunks, the Riccati sweep of one chunk overlapping with the
                           linearisation/projection of the earlier stages (fast kernels only, results are bit-identical).
                           0 or 1 = one launch per stage.  Measured on MI355X: no gain (both sides are bound by LDS
                           bandwidth), so it is off by default; see DESIGN.md. */
  int materialize_lq;   /* 0 (default): "fused" solve - the lineariser leaves in HBM only what the rest of the solve reads (rows 3..11
                           of A and B, b, q, r, the active rows of C, D, e, a 320-byte record of the node-dependent part of Q and R);
                           "A", "B", "Q", "R", "c", "C", "D", "e" of bpmpc_solver_read are then incomplete.
                           != 0: the complete per-node LQ approximation of the reference ([OCS2-upstream] LinearQuadraticApproximator
                           output: A, B, b, Q, R, q, r, c, C, D, e) is written - what the parity stages read and what the roofline
                           unit of bench.py is defined on.  The solution (x, u, K) is the same bits in both modes. */
  double reg_prim;      /* [OCS2-upstream] HPIPM's reg_prim (hpipm_catkin sets 1e-12): added to the diagonal of every stage Hessian
                           [R~ P~'; P~ Q~] of the projected QP, terminal stage included, before the Riccati factorisation.  0 (default) =
                           exact recursion; 1e-12 moves the H1 input step by ~5e-9 relative (tests/test_recalled_behaviours.py). */
  int solver;           /* BPMPC_SOLVER_SQP (0, default): SqpMpc, the solver of BipedalController.cpp:303-306 / BipedalRobotSqpMpcNode.cpp:70.
                           BPMPC_SOLVER_DDP (1): GaussNewtonDDP_MPC of ocs2_bipedal_robot_ros/src/BipedalRobotDdpMpcNode.cpp:70-71 with the ddp block of
                           task.info:115-156 - ONE ILQR iteration per run (ddp.algorithm ILQR, maxNumIterations 1, strategy LINE_SEARCH, hessian
                           correction DIAGONAL_SHIFT; anything else: BPMPC_ERR_UNSUPPORTED): Euler-discretised LQ model on the time grid of the
                           nominal trajectories, equality-constrained Riccati recursion, line search over TimeTriggeredRollout roll-outs of the
                           policy (rollout block of task.info).  The solution is the accepted roll-out ON ITS OWN TIME POINTS: bpmpc_solver_fetch
                           returns them in out_t, stats.n_nodes = points - 1 (at most max_nodes), stats.step_size = the accepted step length,
                           merit_before / merit_after = performance index of the baseline / accepted roll-out; the controller is a
                           FeedforwardController (ddp.useFeedbackPolicy false) - out_K of a DDP solve holds the gains of the policy on the
                           nominal grid, for inspection.  The backward pass runs on the kernels of the SQP path (reference_kernels = 1: the
                           lane-emulated bodies), all step lengths of the line search are rolled out in one launch.  A DDP solution is NOT on the
                           shooting grid: bpmpc_solver_rollout, bpmpc_solver_constraint_values and a second bpmpc_solver_run without a new setup /
                           reset return BPMPC_ERR_UNSUPPORTED.  SLQ, later iterations on the roll-out's grid and the continuous-time backward pass
                           are not implemented. */
  int feedback_policy;  /* 0 (default): sqp.useFeedbackPolicy of task.info decides for the warm start, the policy rollout AND the controller a caller
                           builds from the solution; 1: LinearController, 2: FeedforwardController - one value for all three, as the single
                           sqp::Settings of the reference */
} bpmpc_settings;
#define BPMPC_SOLVER_SQP 0
#define BPMPC_SOLVER_DDP 1

typedef struct {
  int n_events;
  const double* event_times;  /* [n_events] */
  const int* modes;           /* [n_events + 1] */
} bpmpc_mode_schedule;

typedef struct {
  int n_points;
  const double* times;        /* [n_points] */
  const double* states;       /* [n_points * nx] */
} bpmpc_target;

typedef struct {
  int n_nodes;                /* shooting intervals of this problem */
  int iterations;             /* SQP iterations performed */
  int status;                 /* 0 ok, 1 line search took no step, 2 numerical failure (non-positive pivot in the Riccati sweep), 3 (DDP solver) the baseline
                                 roll-out failed or recorded more time points than max_nodes + 1: the nominal trajectories stay */
  int reserved;
  double merit_before, dynamics_sse_before, equality_sse_before;   /* PerformanceIndex of the last linearisation */
  double merit_after, dynamics_sse_after, equality_sse_after;      /* after the accepted step */
  double step_size;           /* alpha of the last iteration (0 = rejected) */
  double armijo_descent;
  double dx_norm, du_norm;
} bpmpc_stats;

int bpmpc_solver_create(const bpmpc_model* model, const bpmpc_settings* settings, bpmpc_solver** out);
void bpmpc_solver_destroy(bpmpc_solver* solver);

/* One call = host pre-pass + upload + SQP iteration(s) + download, for `batch` problems with horizon [t0, t0 + horizon].
 *   t0[batch], x0[batch*nx] (measured state)
 *   schedules: n_schedules == 1 (shared; all t0 must be equal) or == batch
 *   targets[batch]
 *   warm_x[batch*(max_nodes+1)*nx], warm_u[batch*max_nodes*nu]: initial iterate on this solve's grid, or NULL for the
 *     cold start of BipedalRobotInitializer::compute (src/initialization/BipedalRobotInitializer.cpp:56-63)
 * outputs (strides use max_nodes; entries beyond n_nodes are untouched):
 *   out_t[batch*(max_nodes+1)], out_x[batch*(max_nodes+1)*nx], out_u[batch*max_nodes*nu],
 *   out_K[batch*max_nodes*nu*nx] (nullable; needs return_gains), stats[batch] */
int bpmpc_solve_batch(bpmpc_solver* solver, int batch, double horizon, const double* t0, const double* x0,
                      const bpmpc_mode_schedule* schedules, int n_schedules, const bpmpc_target* targets, const double* warm_x,
                      const double* warm_u, double* out_t, double* out_x, double* out_u, double* out_K, bpmpc_stats* stats);

/* The same work split into stages so that a caller can keep everything resident in HBM between solves
 * (bench.py times bpmpc_solver_run only; inputs are already on the device when the timed region starts). */
int bpmpc_solver_setup(bpmpc_solver* solver, int batch, double horizon, const double* t0, const double* x0,
                       const bpmpc_mode_schedule* schedules, int n_schedules, const bpmpc_target* targets, const double* warm_x,
                       const double* warm_u);
/* Receding-horizon step: like bpmpc_solver_setup, but the initial iterate is taken from the previous solve of this handle
 * (same batch), entirely on the device: inside the time span of the previous solution u_k = uff(t_k) + K(t_k) x_k and x_{k+1} is
 * interpolated, beyond it the initializer guess is used ([OCS2-upstream] SqpSolver::initializeStateInputTrajectories with
 * mpc.coldStart false, task.info:173, and sqp.useFeedbackPolicy true, task.info:80 - the MPC loop of
 * bipedal_controllers/src/BipedalController.cpp:332-350).  Needs a completed bpmpc_solver_run on the handle. */
int bpmpc_solver_setup_from_previous(bpmpc_solver* solver, int batch, double horizon, const double* t0, const double* x0,
                                     const bpmpc_mode_schedule* schedules, int n_schedules, const bpmpc_target* targets);
/* Device-side reference generation: the whole pre-pass of a solve on the GPU.  Problem b follows
 * gait template gaits[gait_of_problem[b]] (a ModeSequenceTemplate of gait.info; < 0 or n_gaits == 0: the initial schedule of
 * reference.info only), inserted at gait_start[b] into GaitSchedule(initialModeSchedule, defaultModeSequenceTemplate) and asked for
 * [t0 - horizon, t0 + 2 horizon] (GaitSchedule.cpp:40-137 as called from SwitchedModelReferenceManager.cpp:55-69); swing-height
 * splines, shooting grid and node tables are derived from it per distinct (t0, gait, start), and the target trajectory is
 * cmdVelToTargetTrajectories(cmd_vel[b], t0[b], x0[b]) reaching time_to_target (<= 0: horizon) ahead
 * (TargetTrajectoriesPublisher.cpp:40-62); with command_kind = 1 the four numbers are a goal pose (x, y, unused, yaw) and the
 * target is goalToTargetTrajectories (TargetTrajectoriesPublisher.cpp:64-99, reach time from targetDisplacementVelocity /
 * targetRotationVelocity of reference.info).  Tables are bit-identical to those bpmpc_solver_setup builds on the host from the same
 * schedule; errors (undefined take-off / touch-down, grid longer than max_nodes) are reported the same way.
 * x0 == NULL: the end states of the last bpmpc_solver_rollout on this handle (closed loop without leaving the device).
 * from_previous != 0: initial iterate shifted from the previous solve as in bpmpc_solver_setup_from_previous, else cold start. */
typedef struct {
  int n_modes;
  const double* switching_times; /* n_modes + 1 */
  const int* modes;              /* n_modes */
} bpmpc_gait_template;
int bpmpc_solver_setup_commands(bpmpc_solver* solver, int batch, double horizon, const double* t0, const double* x0,
                                const bpmpc_gait_template* gaits, int n_gaits, const int* gait_of_problem, const double* gait_start,
                                const double* cmd_vel /* [batch][4]: vx, vy, vz, yaw rate */, int command_kind, double time_to_target,
                                int from_previous);
/* MRT side: MRT_BASE::rolloutPolicy for every problem of the batch - TimeTriggeredRollout::run from
 * (t_start[b], x_start[b]) over `duration` under the LinearController of the last solve (u = uff(t) + K(t) x), ODE45 with the
 * rollout block of task.info (AbsTolODE, RelTolODE, timeStep, maxNumStepsPerSecond), restarted at the mode-schedule events inside
 * the window; what MRT_ROS_Dummy_Loop (ocs2_bipedal_robot_ros/src/BipedalRobotDummyNode.cpp:72-86) and BipedalController.cpp:322
 * obtain through initRollout.  t_start / x_start NULL: initial time / measured state of the last solve.  Outputs (nullable):
 * x_end[batch*nx], u_end[batch*nu] (the policy at the end point), steps[batch*2] (accepted, rejected integrator steps).  The end
 * states also stay on the device: bpmpc_solver_setup_commands(x0 = NULL) starts the next solve from them; with all three outputs
 * NULL the call only enqueues and integrator failures are reported by that next setup. */
int bpmpc_solver_rollout(bpmpc_solver* solver, const double* t_start, const double* x_start, double duration, double* x_end,
                         double* u_end, int* steps);
int bpmpc_solver_reset(bpmpc_solver* solver);   /* restore the initial iterate of the last setup (device-side copy, async) */
int bpmpc_solver_run(bpmpc_solver* solver);     /* enqueue the SQP iteration(s) on the solver's stream */
int bpmpc_solver_sync(bpmpc_solver* solver);
int bpmpc_solver_fetch(bpmpc_solver* solver, double* out_t, double* out_x, double* out_u, double* out_K, bpmpc_stats* stats);
/* Run one stage of an iteration on the current iterate (parity tests, roofline measurement):
 * "linearize", "project", "riccati", "linesearch". */
int bpmpc_solver_stage(bpmpc_solver* solver, const char* stage);
/* Copy a named device buffer to the host (tests): "x","u","xref","A","B","b","Q","R","P","q","r","c","C","D","e","nc","perf",
 * "Px","Pu","Pe","nut","dx","du","K","Acl","summary","stats","g_kind","g_mode","g_nodes","g_dt","g_start","g_zref","g_zdref","g_time",
 * "p_grid","x0".  The projected LQ model depends on the kernel set:
 *   settings.reference_kernels = 1:  plain matrices "At","Bt","bt","Qt","Rt","Pt","qt","rt" and the gain scratch "Kt","kt";
 *   fast kernels (default):          the packed layout "Wt" = [At | bt | Bt] (nx rows of WP columns; on the default structured path only
 *                                    its rows 0..11 are written: the JOINT rows 12..nx-1 read as zeros or as the leftovers of an earlier
 *                                    run - they are [I | b | 0] + dt * "Vt" (row j of Vt = joint row 12 + j of [Px | Pe | Pu], dt = g_dt of
 *                                    the node) and every sweep completes them itself; they are written with BPMPC_WT_JOINT_ROWS=1,
 *                                    BPMPC_DENSE_PROJECT=1 or reference_kernels), "Qp" = [Qt | qt] (nx rows of 32
 *                                    columns), "Mt" = [Pt | rt | Rt] (nu rows of WP columns), WP = 16 * ceil((nx + 1 + nu) / 16); block
 *                                    columns beyond nx + 1 + nut and rows >= nut of Mt are not written (kernels/project_node.h PackedLq);
 *                                    the plain names return "unknown buffer".
 * The elimination outputs depend on the path as well: on the default fast path (structured elimination, input weight without force /
 * joint-velocity cross terms) only "Vt" (the joint rows of [Px | Pe | Pu], nj rows of WP columns; columns at and beyond
 * 16 * ceil((nx + 1 + nut) / 16) are not written and hold whatever an earlier, wider node left there), "Pe" and "nut" are written -
 * "Px" and "Pu" are NOT (they read as zeros or as the leftovers of an earlier run on another path); they are written with
 * BPMPC_DENSE_PROJECT=1 (environment, read when the solver is created) and by the reference kernels.
 * Integer buffers are converted to double.  Returns the element count or a negative status; out == NULL only queries the element
 * count. */
int bpmpc_solver_read(bpmpc_solver* solver, const char* name, double* out, long capacity);
/* Device pointers of the iterate, for zero-copy hand-off (e.g. an RCCL gather through torch.distributed):
 * x: batch*(max_nodes+1)*nx doubles, u: batch*max_nodes*nu doubles.  Valid until the next setup with a warm start from the previous
 * solve (the solution buffers then trade places with the kept copy); query again after such a setup. */
int bpmpc_solver_device_trajectories(bpmpc_solver* solver, double** x_dev, double** u_dev);
/* Asynchronous device-to-device copy of the iterate into caller-owned device buffers (same shapes as above) on the
 * solver's stream - e.g. torch tensors that are then all-gathered over RCCL. */
int bpmpc_solver_export_trajectories(bpmpc_solver* solver, double* x_dst_dev, double* u_dst_dev);
/* Solution metrics for solver observers (the reference adds SolverObserver::ConstraintTermObserver on "<foot>_zeroVelocity",
 * ocs2_bipedal_robot_ros/src/BipedalRobotSqpMpcNode.cpp:74-86): the values of the active state-input equality rows at the CURRENT iterate
 * (after a solve: the solution) of every intermediate node, values[b][k][0..rows[b][k]) in registration order per contact i = 0..3:
 * zeroForce_i (3 rows, swing), zeroVelocity_i (3 rows, stance), normalVelocity_i (1 row, swing) (BipedalRobotInterface.cpp:187-191).
 * values: [batch][max_nodes][16]; rows, modes (optional): [batch][max_nodes], 0 / -1 for event nodes and beyond the grid; mode bit 0 =
 * left foot in stance, bit 1 = right foot (MotionPhaseDefinition.h:57-76).  A debugging path: synchronises. */
int bpmpc_solver_constraint_values(bpmpc_solver* solver, double* values, int* rows, int* modes);
/* Change settings.materialize_lq of a live solver. */
int bpmpc_solver_set_materialize(bpmpc_solver* solver, int materialize_lq);
/* Change settings.profile of a live solver (0 / 1 / 2 as above). */
int bpmpc_solver_set_profile(bpmpc_solver* solver, int level);
/* Accumulated HIP-event time of one kernel class since the last call with reset != 0 (needs settings.profile):
 * "prepare","linearize","project","riccati","linesearch". */
int bpmpc_solver_kernel_time(bpmpc_solver* solver, const char* kernel, int reset, double* total_ms, int* launches);
/* Sizes chosen by the last setup: nodes per problem (max over the batch) and number of distinct grids. */
int bpmpc_solver_layout(const bpmpc_solver* solver, int* batch, int* n_nodes_max, int* n_grids, int* nx, int* nu);

/* ---------------------------------------------------------------------------------------------------------------
 * Whole-body controller = WeightedWbc for a BATCH of robots (first slice)
 *   construction replaced:  bipedal_controllers/src/BipedalController.cpp:97-100 (WeightedWbc + loadTasksSetting(taskFile))
 *   update site replaced:   bipedal_controllers/src/BipedalController.cpp:229   wbc_->update(optimizedState, optimizedInput, measuredRbdState_, plannedMode, period)
 *   tasks:                  bipedal_wbc/src/WbcBase.cpp:162-403, QP: bipedal_wbc/src/WeightedWbc.cpp:20-84 (qpOASES, nWSR 20)
 * Decision vector per robot: [generalised accelerations (6 + nj), contact forces (12), joint torques (nj)].
 * rbd_state_measured per robot: [ZYX Euler (3), base position (3), joints (nj), world angular velocity (3), linear velocity (3), joint velocities (nj)]
 * (CentroidalModelRbdConversions layout, WbcBase.cpp:58-77).  A QP that cannot be solved (inconsistent constraints, more than 20 working-set
 * changes) leaves the robot's previous solution in place - lastQpSol_, WeightedWbc.cpp:68-81 - and sets status 1; the handle keeps the last
 * solutions between calls (zero after creation / bpmpc_wbc_reset).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct bpmpc_wbc bpmpc_wbc;
int bpmpc_wbc_create(const bpmpc_model* model, const char* task_info_path, int device, int max_batch, bpmpc_wbc** out);
void bpmpc_wbc_destroy(bpmpc_wbc* wbc);
int bpmpc_wbc_dims(const bpmpc_wbc* wbc, int* n_decision_variables, int* n_generalized_coordinates);
/* state_desired[batch*nx], input_desired[batch*nu], rbd_state_measured[batch*2*(6+nj)], mode[batch] -> solution[batch*n], status[batch] (nullable);
 * debug (nullable, tests): batch*1024 doubles - M, nle, J, Jdot v, base-task right-hand side, rank / iterations / working set */
int bpmpc_wbc_update(bpmpc_wbc* wbc, int batch, const double* state_desired, const double* input_desired, const double* rbd_state_measured,
                     const int* mode, double period, double* solution, int* status, double* debug);
int bpmpc_wbc_reset(bpmpc_wbc* wbc);

#ifdef __cplusplus
}
#endif
#endif /* BPMPC_H */
