/* ORACLE — test infrastructure, NOT product code.
 *
 * Single-thread CPU restatement (IEEE double) of one OCS2 multiple-shooting SQP solve of the
 * ocs2_bipedal_robot problem, i.e. rows a1-a13 of SURVEY.md section 8.  It is the checker the HIP path is
 * compared against and the timed `cpu_baseline` ("port") of bench.py.  Nothing under bipedal_control_amd/
 * may include, link or call it.
 *
 * PARITY STATUS: UNPINNED.  The reference (zitongbai/bipedal_control @ 2024-10-08) cannot be built here
 * (needs ROS/OCS2/Pinocchio/CppAD/HPIPM/Eigen/Boost, none present) and holds no golden vectors for this
 * path.  The arithmetic itself lives in un-vendored, un-pinned third-party code
 * (leggedrobotics/ocs2 main >= the ocs2_ipm commit, Pinocchio, HPIPM, Eigen FullPivLU); functions tagged
 * [OCS2-upstream] restate the published algorithm of those libraries from knowledge of their sources.
 * What pins this oracle instead: finite differences, hand-derived known answers and invariants
 * (tests/test_oracle_*.py) and an independent numpy implementation (oracle/ingest.py, oracle/reference_py.py).
 *
 * Derivatives are produced by forward-mode dual numbers over all nx+nu directions (the reference uses
 * CppAD); the HIP kernels use hand-derived analytic Jacobians, so the two derivations are independent.
 *
 * Model blob layout (doubles; produced by oracle/ingest.py:model_blob), nj = number of leg joints,
 * nx = nu = 12 + nj, bodies 0..nj (0 = floating base with welded links merged), joints 1..nj:
 *   [0]                nj
 *   parent[nj]         parent body of joint j+1
 *   Rfix[nj][9]        rotation of the joint frame in the parent body frame (row major)
 *   pfix[nj][3]        origin of the joint frame in the parent body frame
 *   axis[nj][3]        unit rotation axis in the joint frame
 *   mass[nj+1], com[nj+1][3] (body frame), inertia[nj+1][9] (about com, body frame)
 *   contact_body[4], contact_off[4][3] (contact point in its body frame)
 *   Q[nx*nx], R[nu*nu]
 *   mu_friction, cone_regularization, cone_gripper_force, cone_hessian_shift,
 *   barrier_mu, barrier_delta, position_error_gain, robot_mass
 */
#ifndef BPMPC_ORACLE_H
#define BPMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_model oracle_model;

oracle_model* oracle_model_create(const double* blob, int n);
void oracle_model_destroy(oracle_model*);
int oracle_model_nx(const oracle_model*);

/* a1: flow map and its dense Jacobians (row major A[nx*nx], B[nx*nu]); A/B may be NULL. */
int oracle_flow_map(const oracle_model*, const double* x, const double* u, double* f, double* A, double* B);

/* a6: contact kinematics: pos[4*3], vel[4*3]; optional Jacobians dpdx[12*nx], dvdx[12*nx], dvdu[12*nu]. */
int oracle_ee_kinematics(const oracle_model*, const double* x, const double* u, double* pos, double* vel, double* dpdx,
                         double* dvdx, double* dvdu);

/* centroidal momentum matrix A(q) (6 x (6+nj), row major) and CoM (3). */
int oracle_cmm(const oracle_model*, const double* q, double* A, double* com);

/* One node of the transcription (a13 ii-iii + a2,a3,a5,a7,a8,a9): kind 0 = intermediate, 1 = event.
 * Outputs (row major): A[nx*nx] B[nx*nu] b[nx] Q[nx*nx] R[nu*nu] P[nu*nx] q[nx] r[nu] c[1]
 * C[16*nx] D[16*nu] e[16] nc[1]; perf[3] = {cost, dynamicsViolationSSE, equalityConstraintsSSE}. */
int oracle_node_lq(const oracle_model*, int kind, double dt, const double* x, const double* u, const double* xnext, const double* xref,
                   int mode, const double* zref4, const double* zdref4, double* A, double* B, double* b, double* Q, double* R, double* P,
                   double* q, double* r, double* c, double* C, double* D, double* e, int* nc, double* perf);

/* Value-only node metrics for the line search: perf[3] as above. */
/* operation counts of the restatement (liboracle_count.so, built with -DORACLE_COUNT_FLOPS; returns -1 in liboracle.so):
 * out[5] = flow map value, end-effector kinematics value, flow map with all forward-mode directions, end-effector kinematics with
 * all directions, one complete node linearisation */
int oracle_flop_counts(const oracle_model*, const double* x, const double* u, int mode, double* out);
int oracle_node_perf(const oracle_model*, int kind, double dt, const double* x, const double* u, const double* xnext, const double* xref,
                     int mode, const double* zref4, const double* zdref4, double* perf);

/* Eigen::FullPivLU restatement + OCS2 luConstraintProjection: D is nc x nu.
 * Outputs Px[nu*nx], Pe[nu], Pu[nu*nu] (first nut columns used, row major with stride nu), rank, nut = nu - rank. */
int oracle_lu_projection(int nc, int nx, int nu, const double* C, const double* D, const double* e, double* Px, double* Pu, double* Pe,
                         int* rank);

/* Full solve.  Node arrays have N entries (intervals); x_init (N+1)*nx, u_init N*nu.
 * opts[9] = {sqp_iterations, g_max, g_min, alpha_decay, alpha_min, gamma_c, armijo_factor, delta_tol}
 * Outputs: x_out (N+1)*nx, u_out N*nu, K_out N*nu*nx (nullable), stats[16 * iterations]:
 *   per iteration {merit0, dyn0, eq0, alpha, merit1, dyn1, eq1, armijo_descent, dx_norm, du_norm, n_trials, 0...}. */
int oracle_solve(const oracle_model*, int N, const int* kind, const double* dt, const int* mode, const double* zref, const double* zdref,
                 const double* xref, const double* x0, const double* x_init, const double* u_init, const double* opts, double* x_out,
                 double* u_out, double* K_out, double* stats);

/* The equality-constrained QP of one SQP iteration (projection + Riccati + remap), for KKT-residual tests.
 * Outputs dx (N+1)*nx, du N*nu. */
int oracle_qp_step(const oracle_model*, int N, const int* kind, const double* dt, const int* mode, const double* zref, const double* zdref,
                   const double* xref, const double* x0, const double* x, const double* u, double* dx, double* du, double* K);
/* the same with HPIPM's reg_prim added to the diagonal of every stage Hessian (0 = oracle_qp_step) */
int oracle_qp_step_reg(const oracle_model*, int N, const int* kind, const double* dt, const int* mode, const double* zref, const double* zdref,
                   const double* xref, const double* x0, const double* x, const double* u, double reg_prim, double* dx, double* du, double* K);

#ifdef __cplusplus
}
#endif
#endif
