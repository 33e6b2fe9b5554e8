// Batched weighted whole-body controller (first slice): for every robot of a batch, the QP of
//   WeightedWbc::update                bipedal_wbc/src/WeightedWbc.cpp:20-84
// assembled from the tasks of
//   WbcBase::updateMeasured / updateDesired                                 bipedal_wbc/src/WbcBase.cpp:58-155
//   formulateFloatingBaseEomTask, TorqueLimits, FrictionCone, NoContactMotion (constraints)   :162-198, 346-403
//   formulateSwingLegTask, BaseAccelPDTask, ContactForceTask (weighted cost)                  :234-343
// and solved on the GPU.  HIP only; the CPU restatement (oracle/wbc_py.py) derives the same quantities differently (Lagrangian +
// complex-step derivatives, generic active-set QP) and is the parity reference.
//
// One wavefront per robot, everything in LDS.  Rigid-body dynamics in WORLD coordinates about the world origin with one-DoF
// coordinates (3 base translations, 3 Euler ZYX rotations, the leg joints - the convention of centroidal_model::
// createPinocchioInterface [OCS2-upstream]: v = [base linear velocity, Euler-angle rates, joint rates]):
//   spatial axis of coordinate g:   S_g = [a_g ; o_g x a_g] (revolute about a_g through o_g),  [0 ; e_g] (translation)
//   velocity / bias acceleration of a body: chain walk  A += crm(V) S_g v_g,  V += S_g v_g   (A starts at [0 ; +g e_z]: gravity)
//   body wrench about the origin:   f = m a_c,  n = c x f + I alpha + omega x (I omega)
//   nonlinear effects  nle_g = S_g . (sum of the wrenches of the subtree moved by g)                      (RNEA with zero accelerations)
//   mass matrix        M_hg  = S_h . (I^c_g S_g)  for h an ancestor of g, I^c_g = composite inertia of that subtree    (CRBA)
//   contact Jacobian column g = S_g evaluated at the contact point; Jdot v = its classical bias acceleration
//   centroidal momentum matrix column g = I^c_g S_g shifted to the centre of mass; Adot v = total bias wrench about the com
// The QP is reduced exactly before it is solved: the swing forces are zero, the joint torques are given by the joint rows of the
// equations of motion (they only appear in the torque limits), the pairs of opposite rows of the no-contact-motion task are the
// equalities they amount to (J a + Jdot v = tolerance, see oracle/wbc_py.py on this reference quirk).  What is left - 6 + 3 n_stance
// equalities on y = [vdot, F_stance] - is eliminated by an LU with complete pivoting (rank deficient: two points per rigid foot),
// the reduced problem (<= 12 unknowns, torque limits and friction pyramids as inequalities) by a primal active-set iteration on
// small dense KKT systems, capped like the reference's nWSR = 20.  Inconsistent equalities or an exhausted iteration budget leave
// the previous solution in place (WeightedWbc.cpp:68-81) and report status 1.
#pragma once
#include <hip/hip_runtime.h>

#include "../device_model.h"
#include "riccati_fast.h"   // lds_wave_sync

namespace bpmpc {

struct WbcSettings {
  double torque_limits[kMaxJoints / 2];
  double friction, swing_kp, swing_kd, base_kp[6], base_kd[6], contact_tolerance, w_swing, w_base, w_force;
  int max_working_set_changes;         // qpOASES nWSR of the reference (20)
};

constexpr double kWbcGravity = 9.81;
constexpr double kWbcFeasTol = 1e-8;   // relative: equality rows that cannot all hold (as the oracle)
constexpr double kWbcActiveTol = 1e-9;

__device__ __forceinline__ void w_cross(const double* a, const double* b, double* c) {
  const double c0 = a[1] * b[2] - a[2] * b[1], c1 = a[2] * b[0] - a[0] * b[2], c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
__device__ __forceinline__ double w_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void w_mat3_vec(const double* R, const double* v, double* out) {
  const double a = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], b = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], c = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  out[0] = a; out[1] = b; out[2] = c;
}
__device__ __forceinline__ void w_mat3_mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  for (int i = 0; i < 9; ++i) C[i] = t[i];
}
// symmetric 3x3 stored as xx, xy, xz, yy, yz, zz
__device__ __forceinline__ void w_sym3_vec(const double* S, const double* v, double* out) {
  const double a = S[0] * v[0] + S[1] * v[1] + S[2] * v[2], b = S[1] * v[0] + S[3] * v[1] + S[4] * v[2], c = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
  out[0] = a; out[1] = b; out[2] = c;
}
__device__ __forceinline__ double wave_max(double x) {
  for (int m = 1; m < kWave; m <<= 1) x = fmax(x, __shfl_xor(x, m));
  return x;
}
__device__ __forceinline__ int wave_min_int(int x) {
  for (int m = 1; m < kWave; m <<= 1) { const int o = __shfl_xor(x, m); x = o < x ? o : x; }
  return x;
}

// ---------------------------------------------------------------------------------------------------------------- rigid-body pass
template <int NJ>
struct WbcRbd {
  static constexpr int NV = 6 + NJ, NB = NJ + 1;
  double q[NV], v[NV];
  double R[NB][9], o[NB][3];             // body frames (after the joint rotation) and their origins
  double S[NV][6];                       // spatial axes [w ; v_O]
  double cw[NB][3], Iw[NB][6];           // centre of mass (world), inertia about it in world axes
  double Vw[NB][3], Vo[NB][3], Aw[NB][3], Ao[NB][3];
  double comp[NB][10];                   // m, m c, inertia about the origin
  double wr[NB][6];                      // bias wrench about the origin (n, f)
  double cs[NV][10];                     // composite of the subtree moved by a coordinate
  double ws[NV][6];                      // bias wrench of that subtree
  double fc[NV][6];                      // I^c_g S_g  (n, f)
  double cp[kNumContacts][3], cv[kNumContacts][3], ca[kNumContacts][3];   // contact point, its velocity, its bias acceleration (no gravity)
  double com[3], mass;
};

// body moved by coordinate g (base coordinates: body 0 and, through it, every body)
__device__ __forceinline__ int wbc_body_of(int g) { return g < 6 ? 0 : g - 5; }
// does coordinate h lie on the chain of coordinate g (h == g included)?
__device__ __forceinline__ bool wbc_on_chain(const DeviceModel& md, int h, int g) {
  if (h < 6) return g >= 6 || h <= g;
  if (g < 6) return false;
  return (md.subtree[h - 5] >> (g - 5)) & 1u;
}

// q, v must be in r.q / r.v.  gravity: bias accelerations start from [0 ; +g e_z] (nonlinear effects) or from zero (Adot v).
template <int NJ>
__device__ void wbc_rbd_pass(const DeviceModel& md, WbcRbd<NJ>& r, bool gravity, int l) {
  constexpr int NV = 6 + NJ, NB = NJ + 1;
  lds_wave_sync();
  // ---- body frames by a chain walk per body
  if (l < NB) {
    const double cy = cos(r.q[3]), sy = sin(r.q[3]), cp = cos(r.q[4]), sp = sin(r.q[4]), cr = cos(r.q[5]), sr = sin(r.q[5]);
    double R[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr};
    double o[3] = {r.q[0], r.q[1], r.q[2]};
    const int depth = md.depth[l];
    for (int d = 0; d < depth; ++d) {
      const int j = md.path[l][d];
      const double* a = md.axis[j];
      const double c = cos(r.q[5 + j]), s = sin(r.q[5 + j]), vv = 1.0 - c;
      const double rot[9] = {c + vv * a[0] * a[0], vv * a[0] * a[1] - s * a[2], vv * a[0] * a[2] + s * a[1],
                             vv * a[1] * a[0] + s * a[2], c + vv * a[1] * a[1], vv * a[1] * a[2] - s * a[0],
                             vv * a[2] * a[0] - s * a[1], vv * a[2] * a[1] + s * a[0], c + vv * a[2] * a[2]};
      double t[3], E[9];
      w_mat3_vec(R, md.pfix[j], t);
      for (int i = 0; i < 3; ++i) o[i] += t[i];
      w_mat3_mul(md.Rfix[j], rot, E);
      w_mat3_mul(R, E, R);
    }
    for (int i = 0; i < 9; ++i) r.R[l][i] = R[i];
    for (int i = 0; i < 3; ++i) r.o[l][i] = o[i];
    // spatial axis of the coordinate that moves this body
    if (l >= 1) {
      double ah[3], sv[3];
      w_mat3_vec(R, md.axis[l], ah);
      w_cross(o, ah, sv);
      for (int i = 0; i < 3; ++i) { r.S[5 + l][i] = ah[i]; r.S[5 + l][3 + i] = sv[i]; }
    } else {
      const double ax[3][3] = {{0.0, 0.0, 1.0}, {-sy, cy, 0.0}, {cy * cp, sy * cp, -sp}};     // world axes of the z, y', x'' rotations
      for (int g = 0; g < 3; ++g) {
        for (int i = 0; i < 3; ++i) { r.S[g][i] = 0.0; r.S[g][3 + i] = (i == g) ? 1.0 : 0.0; }
        double sv[3];
        w_cross(o, ax[g], sv);
        for (int i = 0; i < 3; ++i) { r.S[3 + g][i] = ax[g][i]; r.S[3 + g][3 + i] = sv[i]; }
      }
    }
    // inertial quantities of the body in the world frame
    double cb[3];
    w_mat3_vec(R, md.com[l], cb);
    const double c[3] = {o[0] + cb[0], o[1] + cb[1], o[2] + cb[2]};
    const double* I = md.inertia[l];
    const double Ib[9] = {I[0], I[1], I[2], I[1], I[3], I[4], I[2], I[4], I[5]};
    double T[9];
    w_mat3_mul(R, Ib, T);
    double Iw[6];
    Iw[0] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];
    Iw[1] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
    Iw[2] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
    Iw[3] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];
    Iw[4] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
    Iw[5] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
    const double m = md.mass[l], cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    for (int i = 0; i < 3; ++i) r.cw[l][i] = c[i];
    for (int i = 0; i < 6; ++i) r.Iw[l][i] = Iw[i];
    double* cm = r.comp[l];
    cm[0] = m; cm[1] = m * c[0]; cm[2] = m * c[1]; cm[3] = m * c[2];
    cm[4] = Iw[0] + m * (cc - c[0] * c[0]); cm[5] = Iw[1] - m * c[0] * c[1]; cm[6] = Iw[2] - m * c[0] * c[2];
    cm[7] = Iw[3] + m * (cc - c[1] * c[1]); cm[8] = Iw[4] - m * c[1] * c[2]; cm[9] = Iw[5] + m * (cc - c[2] * c[2]);
  }
  lds_wave_sync();
  // ---- velocity and bias acceleration of every body: chain walk over the coordinates that move it
  if (l < NB) {
    double Vw[3] = {0, 0, 0}, Vo[3] = {0, 0, 0}, Aw[3] = {0, 0, 0}, Ao[3] = {0, 0, gravity ? kWbcGravity : 0.0};
    const int depth = md.depth[l];
    for (int step = 0; step < 6 + depth; ++step) {
      const int g = step < 6 ? step : 5 + md.path[l][step - 6];
      const double* s = r.S[g];
      const double vg = r.v[g];
      double t1[3], t2[3], t3[3];
      w_cross(Vw, s, t1);            // w x s_w
      w_cross(Vw, s + 3, t2);        // w x s_v
      w_cross(Vo, s, t3);            // v_O x s_w
      for (int i = 0; i < 3; ++i) { Aw[i] += t1[i] * vg; Ao[i] += (t2[i] + t3[i]) * vg; }
      for (int i = 0; i < 3; ++i) { Vw[i] += s[i] * vg; Vo[i] += s[3 + i] * vg; }
    }
    for (int i = 0; i < 3; ++i) { r.Vw[l][i] = Vw[i]; r.Vo[l][i] = Vo[i]; r.Aw[l][i] = Aw[i]; r.Ao[l][i] = Ao[i]; }
    // bias wrench of the body about the origin
    const double* c = r.cw[l];
    double t[3], vc[3], ac[3], f[3], n[3], Iw_w[3], Ia[3], t4[3];
    w_cross(Vw, c, t);
    for (int i = 0; i < 3; ++i) vc[i] = Vo[i] + t[i];
    w_cross(Aw, c, t);
    w_cross(Vw, vc, t4);
    for (int i = 0; i < 3; ++i) ac[i] = Ao[i] + t[i] + t4[i];
    for (int i = 0; i < 3; ++i) f[i] = md.mass[l] * ac[i];
    w_cross(c, f, n);
    w_sym3_vec(r.Iw[l], Aw, Ia);
    w_sym3_vec(r.Iw[l], Vw, Iw_w);
    w_cross(Vw, Iw_w, t);
    for (int i = 0; i < 3; ++i) { r.wr[l][i] = n[i] + Ia[i] + t[i]; r.wr[l][3 + i] = f[i]; }
  }
  lds_wave_sync();
  // ---- per coordinate: composite inertia and bias wrench of the subtree it moves, I^c S
  if (l < NV) {
    const unsigned mask = l < 6 ? md.subtree[0] : md.subtree[l - 5];
    double c10[10], w6[6];
    for (int i = 0; i < 10; ++i) c10[i] = 0.0;
    for (int i = 0; i < 6; ++i) w6[i] = 0.0;
    for (int b = 0; b < NB; ++b)
      if ((mask >> b) & 1u) {
        for (int i = 0; i < 10; ++i) c10[i] += r.comp[b][i];
        for (int i = 0; i < 6; ++i) w6[i] += r.wr[b][i];
      }
    for (int i = 0; i < 10; ++i) r.cs[l][i] = c10[i];
    for (int i = 0; i < 6; ++i) r.ws[l][i] = w6[i];
    const double* s = r.S[l];
    const double h[3] = {c10[1], c10[2], c10[3]};
    double Ia[3], t1[3], t2[3];
    w_sym3_vec(&c10[4], s, Ia);
    w_cross(h, s + 3, t1);           // h x a_O
    w_cross(s, h, t2);               // alpha x h
    for (int i = 0; i < 3; ++i) { r.fc[l][i] = Ia[i] + t1[i]; r.fc[l][3 + i] = c10[0] * s[3 + i] + t2[i]; }
    if (l == 0) { r.mass = c10[0]; for (int i = 0; i < 3; ++i) r.com[i] = c10[1 + i] / c10[0]; }
  }
  // ---- contact points: position, velocity, bias acceleration without gravity
  if (l < kNumContacts) {
    const int b = md.contact_body[l];
    double t[3], p[3], vp[3], t2[3], t3[3];
    w_mat3_vec(r.R[b], md.contact_off[l], t);
    for (int i = 0; i < 3; ++i) p[i] = r.o[b][i] + t[i];
    w_cross(r.Vw[b], p, t);
    for (int i = 0; i < 3; ++i) vp[i] = r.Vo[b][i] + t[i];
    w_cross(r.Aw[b], p, t2);
    w_cross(r.Vw[b], vp, t3);
    for (int i = 0; i < 3; ++i) {
      r.cp[l][i] = p[i]; r.cv[l][i] = vp[i];
      r.ca[l][i] = r.Ao[b][i] - ((gravity && i == 2) ? kWbcGravity : 0.0) + t2[i] + t3[i];
    }
  }
  lds_wave_sync();
}

// column g of the contact Jacobian of contact i (3 entries)
template <int NJ>
__device__ __forceinline__ void wbc_contact_jac_col(const DeviceModel& md, const WbcRbd<NJ>& r, int i, int g, double* col) {
  const bool moves = g < 6 || ((md.contact_path[i] >> (g - 5)) & 1u);
  if (!moves) { col[0] = col[1] = col[2] = 0.0; return; }
  double t[3];
  w_cross(r.S[g], r.cp[i], t);
  for (int a = 0; a < 3; ++a) col[a] = r.S[g][3 + a] + t[a];
}

// ---------------------------------------------------------------------------------------------------------------- workspace
template <int NJ>
struct WbcLds {
  static constexpr int NV = 6 + NJ, NX = 12 + NJ, NY = NV + 12, NE = 18, NI = 2 * NJ + 20, NZ = NY, NK = 32;
  WbcRbd<NJ> rbd;
  double M[NV][NV], nle[NV], J[12][NV], djv[12];
  double jb[3][NV], dbv[3];              // angular base Jacobian (rows 3..5 of baseJ_), baseDj v
  double xd[NX], ud[NX];
  double pos_d[kNumContacts][3], vel_d[kNumContacts][3], pos_m[kNumContacts][3], vel_m[kNumContacts][3];
  double base_b[6];                      // right-hand side of the base acceleration task
  int flag[kNumContacts], ycol[kNumContacts], nst, ny;
  double C[NE][NY + 1];                  // equalities on y = [vdot, F_stance], right-hand side in the last column
  double H[NY][NY], gy[NY];
  double D[NI][NY], fi[NI];              // inequalities on y
  double yp[NY], Z[NY][NY];              // particular solution and null-space basis of the equalities
  double Hz[NZ][NZ], gz[NZ], Dz[NI][NZ], fz[NI];
  double K[NK][NK + 1];                  // KKT system of the active-set iteration
  double z[NZ], mu[NI];
  int colperm[NY], work[NI], nwork, rank, status, iters;
  double y[NY];
};

// ---------------------------------------------------------------------------------------------------------------- the controller
struct WbcArgs {
  int batch, nx;
  const double *state_des, *input_des, *rbd_meas;   // [B][nx], [B][nx], [B][2 nv]
  const int* mode;                                  // [B]
  double* sol;                                      // [B][n]: in = last solution, out = new one (or unchanged on failure)
  int* status;                                      // [B]
  double* debug;                                    // optional [B][kWbcDebugStride]
};
constexpr int kWbcDebugStride = 1024;

template <int NJ>
__device__ void wbc_robot(const DeviceModel& md, const WbcSettings& st, WbcLds<NJ>& w, const WbcArgs& a, int b, int l) {
  using W = WbcLds<NJ>;
  constexpr int NV = W::NV, NX = W::NX, NY = W::NY, NE = W::NE, NI = W::NI;
  const int n = NV + 12 + NJ;
  WbcRbd<NJ>& r = w.rbd;
  // ======================================================= desired state (WbcBase::updateDesired, computeBaseKinematicsFromCentroidalModel)
  for (int i = l; i < NX; i += kWave) { w.xd[i] = a.state_des[(size_t)b * NX + i]; w.ud[i] = a.input_des[(size_t)b * NX + i]; }
  if (l < kNumContacts) {
    const int mode = a.mode[b];
    w.flag[l] = (l < 2) ? ((mode == 1 || mode == 3) ? 1 : 0) : ((mode == 2 || mode == 3) ? 1 : 0);     // MotionPhaseDefinition.h:57-76
  }
  lds_wave_sync();
  if (l < NV) { r.q[l] = w.xd[6 + l]; r.v[l] = l >= 6 ? w.ud[12 + l - 6] : 0.0; }
  wbc_rbd_pass<NJ>(md, r, false, l);      // first with zero base velocity: only the composite quantities are used
  double pose_d[6], vel_d[6], acc_d[6];   // lane 0
  if (l == 0) {
    // centroidal momentum matrix columns K_g = [F ; N - c x F] of I^c_g S_g; v_base = Ab^-1 (m hbar - Aj vj)
    double rhs[6];
    for (int i = 0; i < 6; ++i) rhs[i] = md.robot_mass * w.xd[i];
    for (int g = 6; g < NV; ++g) {
      double t[3];
      w_cross(r.com, &r.fc[g][3], t);
      for (int i = 0; i < 3; ++i) { rhs[i] -= r.fc[g][3 + i] * r.v[g]; rhs[3 + i] -= (r.fc[g][i] - t[i]) * r.v[g]; }
    }
    // Ab = [[m I, A12], [0, A22]] (translations carry no angular momentum about the com)
    double A12[9], A22[9];
    for (int g = 3; g < 6; ++g) {
      double t[3];
      w_cross(r.com, &r.fc[g][3], t);
      for (int i = 0; i < 3; ++i) { A12[3 * i + (g - 3)] = r.fc[g][3 + i]; A22[3 * i + (g - 3)] = r.fc[g][i] - t[i]; }
    }
    const double* Mx = A22;
    const double c00 = Mx[4] * Mx[8] - Mx[5] * Mx[7], c01 = Mx[5] * Mx[6] - Mx[3] * Mx[8], c02 = Mx[3] * Mx[7] - Mx[4] * Mx[6];
    const double idet = 1.0 / (Mx[0] * c00 + Mx[1] * c01 + Mx[2] * c02);
    double X22[9] = {c00 * idet, (Mx[2] * Mx[7] - Mx[1] * Mx[8]) * idet, (Mx[1] * Mx[5] - Mx[2] * Mx[4]) * idet,
                     c01 * idet, (Mx[0] * Mx[8] - Mx[2] * Mx[6]) * idet, (Mx[2] * Mx[3] - Mx[0] * Mx[5]) * idet,
                     c02 * idet, (Mx[1] * Mx[6] - Mx[0] * Mx[7]) * idet, (Mx[0] * Mx[4] - Mx[1] * Mx[3]) * idet};
    double th[3], t[3];
    w_mat3_vec(X22, &rhs[3], th);
    w_mat3_vec(A12, th, t);
    const double im = 1.0 / r.mass;
    for (int i = 0; i < 3; ++i) { r.v[i] = (rhs[i] - t[i]) * im; r.v[3 + i] = th[i]; }
    // keep what the second pass overwrites: A12, X22 live in registers of this lane
    w.K[0][0] = im;
    for (int i = 0; i < 9; ++i) { w.K[1][i] = A12[i]; w.K[2][i] = X22[i]; }
  }
  wbc_rbd_pass<NJ>(md, r, false, l);      // now with the full desired velocity: contact velocities, total bias wrench = Adot v
  if (l < kNumContacts)
    for (int i = 0; i < 3; ++i) { w.pos_d[l][i] = r.cp[l][i]; w.vel_d[l][i] = r.cv[l][i]; }
  if (l == 0) {
    const double im = w.K[0][0];
    const double* A12 = w.K[1];
    const double* X22 = w.K[2];
    // momentum rate of the planned contact forces minus Adot v (joint accelerations are zero, WbcBase.cpp:243)
    double hd[6] = {0.0, 0.0, -kWbcGravity * md.robot_mass, 0.0, 0.0, 0.0};
    for (int i = 0; i < kNumContacts; ++i) {
      const double* F = &w.ud[3 * i];
      const double rr[3] = {r.cp[i][0] - r.com[0], r.cp[i][1] - r.com[1], r.cp[i][2] - r.com[2]};
      double t[3];
      w_cross(rr, F, t);
      for (int k = 0; k < 3; ++k) { hd[k] += F[k]; hd[3 + k] += t[k]; }
    }
    double t[3];
    w_cross(r.com, &r.ws[0][3], t);
    for (int k = 0; k < 3; ++k) { hd[k] -= r.ws[0][3 + k]; hd[3 + k] -= r.ws[0][k] - t[k]; }
    double thdd[3], t2[3], pdd[3];
    w_mat3_vec(X22, &hd[3], thdd);
    w_mat3_vec(A12, thdd, t2);
    for (int k = 0; k < 3; ++k) pdd[k] = (hd[k] - t2[k]) * im;
    for (int k = 0; k < 6; ++k) pose_d[k] = r.q[k];
    for (int k = 0; k < 3; ++k) {
      vel_d[k] = r.v[k];
      vel_d[3 + k] = r.S[3][k] * r.v[3] + r.S[4][k] * r.v[4] + r.S[5][k] * r.v[5];
      acc_d[k] = pdd[k];
      // E thetaddot + Edot thetadot; the latter is the angular bias acceleration of the base body
      acc_d[3 + k] = r.S[3][k] * thdd[0] + r.S[4][k] * thdd[1] + r.S[5][k] * thdd[2] + r.Aw[0][k];
    }
  }
  // ======================================================= measured state (WbcBase::updateMeasured)
  lds_wave_sync();
  if (l < NV) {
    const double* rb = a.rbd_meas + (size_t)b * 2 * NV;
    double qv;
    if (l < 3) qv = rb[3 + l]; else if (l < 6) qv = rb[l - 3]; else qv = rb[l];
    r.q[l] = qv;
  }
  lds_wave_sync();
  if (l == 0) {
    const double* rb = a.rbd_meas + (size_t)b * 2 * NV;
    for (int i = 0; i < 3; ++i) r.v[i] = rb[NV + 3 + i];
    for (int j = 0; j < NJ; ++j) r.v[6 + j] = rb[NV + 6 + j];
    // Euler-angle rates from the world angular velocity: solve E thetadot = omega, E = [e_z, Rz e_y, Rz Ry e_x]
    const double cy = cos(r.q[3]), sy = sin(r.q[3]), cp = cos(r.q[4]), sp = sin(r.q[4]);
    const double wx = rb[NV], wy = rb[NV + 1], wz = rb[NV + 2];
    const double rr = (cy * wx + sy * wy) / cp;          // roll rate
    const double pr = -sy * wx + cy * wy;                // pitch rate
    r.v[3] = wz + sp * rr; r.v[4] = pr; r.v[5] = rr;
  }
  wbc_rbd_pass<NJ>(md, r, true, l);
  // mass matrix, nonlinear effects, contact Jacobian and its bias, angular base Jacobian
  for (int idx = l; idx < NV * NV; idx += kWave) {
    const int h = idx / NV, g = idx % NV;
    double val = 0.0;
    if (wbc_on_chain(md, h, g)) val = w_dot(r.S[h], r.fc[g]) + w_dot(r.S[h] + 3, r.fc[g] + 3);
    else if (wbc_on_chain(md, g, h)) val = w_dot(r.S[g], r.fc[h]) + w_dot(r.S[g] + 3, r.fc[h] + 3);
    w.M[h][g] = val;
  }
  if (l < NV) w.nle[l] = w_dot(r.S[l], r.ws[l]) + w_dot(r.S[l] + 3, r.ws[l] + 3);
  for (int idx = l; idx < kNumContacts * NV; idx += kWave) {
    const int i = idx / NV, g = idx % NV;
    double col[3];
    wbc_contact_jac_col<NJ>(md, r, i, g, col);
    for (int k = 0; k < 3; ++k) w.J[3 * i + k][g] = col[k];
  }
  if (l < kNumContacts)
    for (int k = 0; k < 3; ++k) { w.djv[3 * l + k] = r.ca[l][k]; w.pos_m[l][k] = r.cp[l][k]; w.vel_m[l][k] = r.cv[l][k]; }
  if (l < NV)
    for (int k = 0; k < 3; ++k) w.jb[k][l] = (l >= 3 && l < 6) ? r.S[l][k] : 0.0;
  if (l == 0) {
    for (int k = 0; k < 3; ++k) w.dbv[k] = r.Aw[0][k];
    // ---- base acceleration PD task, right-hand side (WbcBase.cpp:234-287)
    double Rd[9], Rm[9];
    {
      const double z = pose_d[3], y = pose_d[4], x = pose_d[5];
      const double cz = cos(z), sz = sin(z), cyy = cos(y), syy = sin(y), cx = cos(x), sx = sin(x);
      const double T[9] = {cz * cyy, cz * syy * sx - sz * cx, cz * syy * cx + sz * sx, sz * cyy, sz * syy * sx + cz * cx, sz * syy * cx - cz * sx, -syy, cyy * sx, cyy * cx};
      for (int i = 0; i < 9; ++i) Rd[i] = T[i];
      for (int i = 0; i < 9; ++i) Rm[i] = r.R[0][i];
    }
    // rotationErrorInWorld(Rd, Rm) = rotation vector of Rd Rm'
    double E[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) E[3 * i + j] = Rd[3 * i] * Rm[3 * j] + Rd[3 * i + 1] * Rm[3 * j + 1] + Rd[3 * i + 2] * Rm[3 * j + 2];
    const double tr = E[0] + E[4] + E[8];
    const double skew[3] = {E[7] - E[5], E[2] - E[6], E[3] - E[1]};
    const double tmp = 0.5 * (tr - 3.0);
    double e_rot[3];
    if (-tmp < 1e-8) { for (int k = 0; k < 3; ++k) e_rot[k] = (0.5 - tmp / 6.0) * skew[k]; }
    else { const double th = acos(0.5 * (tr - 1.0)); const double f = th / (2.0 * sin(th)); for (int k = 0; k < 3; ++k) e_rot[k] = f * skew[k]; }
    for (int k = 0; k < 3; ++k) {
      const double e_pos = pose_d[k] - r.q[k], e_lin = vel_d[k] - r.v[k];
      w.base_b[k] = acc_d[k] + st.base_kp[k] * e_pos + st.base_kd[k] * e_lin;
      // the reference's "angular velocity error" is desiredBaseVelocity.head<3>(3) - measured.head<3>(3): the linear one (WbcBase.cpp:274)
      w.base_b[3 + k] = acc_d[3 + k] + st.base_kp[3 + k] * e_rot[k] + st.base_kd[3 + k] * e_lin - w.dbv[k];
    }
    int ns = 0;
    for (int i = 0; i < kNumContacts; ++i) { w.ycol[i] = w.flag[i] ? NV + 3 * ns : -1; ns += w.flag[i]; }
    w.nst = ns; w.ny = NV + 3 * ns;
  }
  lds_wave_sync();
  const int ny = w.ny, nst = w.nst, ne = 6 + 3 * nst, ni = 2 * NJ + 5 * nst;
  // ======================================================= reduced QP on y = [vdot, F_stance]
  // equalities: base rows of the equations of motion, contact accelerations
  for (int idx = l; idx < ne * (NY + 1); idx += kWave) {
    const int row = idx / (NY + 1), c = idx % (NY + 1);
    double val = 0.0;
    if (row < 6) {
      if (c < NV) val = w.M[row][c];
      else if (c < ny) { int ci = 0; for (int i = 0; i < kNumContacts; ++i) if (w.ycol[i] >= 0 && c >= w.ycol[i] && c < w.ycol[i] + 3) ci = i; val = -w.J[3 * ci + (c - w.ycol[ci])][row]; }
      else if (c == NY) val = -w.nle[row];
    } else {
      const int k = (row - 6) / 3, ax = (row - 6) % 3;
      int ci = 0, seen = 0;
      for (int i = 0; i < kNumContacts; ++i) if (w.flag[i]) { if (seen == k) ci = i; ++seen; }
      if (c < NV) val = w.J[3 * ci + ax][c];
      else if (c == NY) val = -w.djv[3 * ci + ax] + st.contact_tolerance;
    }
    w.C[row][c] = val;
  }
  // cost: H = W'W, g = -W'b over the weighted task rows (swing leg, base acceleration, contact forces)
  for (int idx = l; idx < ny * (ny + 1); idx += kWave) {
    const int i = idx / (ny + 1), j = idx % (ny + 1);
    double acc = 0.0;
    auto col = [&](int rowkind, int rr, int c) -> double {   // entry (row, column c) of a task matrix; c == ny: right-hand side
      if (rowkind == 0) {                                    // swing leg rows of contact rr / 3, axis rr % 3
        const int ci = rr / 3, ax = rr % 3;
        if (c < NV) return st.w_swing * w.J[3 * ci + ax][c];
        if (c == ny) return st.w_swing * (st.swing_kp * (w.pos_d[ci][ax] - w.pos_m[ci][ax]) + st.swing_kd * (w.vel_d[ci][ax] - w.vel_m[ci][ax]) - w.djv[3 * ci + ax]);
        return 0.0;
      }
      if (rowkind == 1) {                                    // base acceleration rows
        if (c < NV) return st.w_base * (rr < 3 ? (c == rr ? 1.0 : 0.0) : w.jb[rr - 3][c]);
        if (c == ny) return st.w_base * w.base_b[rr];
        return 0.0;
      }
      const int ci = rr / 3, ax = rr % 3;                    // contact force rows (stance contacts only: swing forces are zero)
      if (c == ny) return st.w_force * w.ud[3 * ci + ax];
      return (w.ycol[ci] >= 0 && c == w.ycol[ci] + ax) ? st.w_force : 0.0;
    };
    for (int ci = 0; ci < kNumContacts; ++ci)
      if (!w.flag[ci]) for (int ax = 0; ax < 3; ++ax) acc += col(0, 3 * ci + ax, i) * col(0, 3 * ci + ax, j);
    for (int rr = 0; rr < 6; ++rr) acc += col(1, rr, i) * col(1, rr, j);
    for (int ci = 0; ci < kNumContacts; ++ci)
      if (w.flag[ci]) for (int ax = 0; ax < 3; ++ax) acc += col(2, 3 * ci + ax, i) * col(2, 3 * ci + ax, j);
    if (j < ny) w.H[i][j] = acc; else w.gy[i] = -acc;
  }
  // inequalities: torque limits through the joint rows of the equations of motion, friction pyramids
  for (int idx = l; idx < ni * (NY + 1); idx += kWave) {
    const int row = idx / (NY + 1), c = idx % (NY + 1);
    double val = 0.0;
    if (row < 2 * NJ) {
      const int j = row % NJ;
      const double sgn = row < NJ ? 1.0 : -1.0;
      if (c < NV) val = sgn * w.M[6 + j][c];
      else if (c < ny) { int ci = 0; for (int i = 0; i < kNumContacts; ++i) if (w.ycol[i] >= 0 && c >= w.ycol[i] && c < w.ycol[i] + 3) ci = i; val = -sgn * w.J[3 * ci + (c - w.ycol[ci])][6 + j]; }
      else if (c == NY) val = st.torque_limits[j % (NJ / 2)] - sgn * w.nle[6 + j];
    } else {
      const int k = (row - 2 * NJ) / 5, pr = (row - 2 * NJ) % 5;
      int ci = 0, seen = 0;
      for (int i = 0; i < kNumContacts; ++i) if (w.flag[i]) { if (seen == k) ci = i; ++seen; }
      const double mu = st.friction;
      const double pyr[5][3] = {{0, 0, -1}, {1, 0, -mu}, {-1, 0, -mu}, {0, 1, -mu}, {0, -1, -mu}};
      if (c >= w.ycol[ci] && c < w.ycol[ci] + 3) val = pyr[pr][c - w.ycol[ci]];
    }
    if (c < NY) w.D[row][c] = val; else w.fi[row] = val;
  }
  lds_wave_sync();
  // ======================================================= elimination of the equalities: LU with complete pivoting of C (ne x ny | rhs)
  if (l < NY) w.colperm[l] = l;
  lds_wave_sync();
  int rank = 0;
  double maxpiv = 0.0;
  for (int k = 0; k < ne && k < ny; ++k) {
    // largest entry of the trailing block (ties: smallest packed index)
    double best = -1.0;
    int where = 0x7fffffff;
    for (int idx = l; idx < (ne - k) * (ny - k); idx += kWave) {
      const int i = k + idx / (ny - k), j = k + idx % (ny - k);
      const double v = fabs(w.C[i][j]);
      if (v > best) { best = v; where = (j << 8) | i; }      // column-major order of the first maximum, like the oracle's / Eigen's rule
      else if (v == best) { const int key = (j << 8) | i; where = key < where ? key : where; }
    }
    const double gmax = wave_max(best);
    const int key = wave_min_int(best == gmax ? where : 0x7fffffff);
    if (!(gmax > 0.0)) break;
    if (k == 0) maxpiv = gmax;
    if (gmax <= maxpiv * 2.220446049250313e-16 * (ne < ny ? ne : ny)) break;     // below the rank threshold of FullPivLU
    maxpiv = fmax(maxpiv, gmax);
    const int pi = key & 255, pj = key >> 8;
    lds_wave_sync();
    // swap rows k <-> pi (all columns incl. rhs), columns k <-> pj (all rows)
    for (int c = l; c <= NY; c += kWave) { const double t = w.C[k][c]; w.C[k][c] = w.C[pi][c]; w.C[pi][c] = t; }
    lds_wave_sync();
    for (int i = l; i < ne; i += kWave) { const double t = w.C[i][k]; w.C[i][k] = w.C[i][pj]; w.C[i][pj] = t; }
    if (l == 0) { const int t = w.colperm[k]; w.colperm[k] = w.colperm[pj]; w.colperm[pj] = t; }
    lds_wave_sync();
    const double inv = 1.0 / w.C[k][k];
    for (int idx = l; idx < (ne - k - 1) * (ny - k + 1); idx += kWave) {       // columns k+1 .. ny-1 and the rhs (index NY)
      const int i = k + 1 + idx / (ny - k + 1), jj = idx % (ny - k + 1);
      if (jj == 0) continue;
      const int j = (jj == ny - k) ? NY : k + jj;
      w.C[i][j] -= w.C[i][k] * inv * w.C[k][j];
    }
    lds_wave_sync();
    for (int i = k + 1 + l; i < ne; i += kWave) w.C[i][k] = 0.0;
    lds_wave_sync();
    rank = k + 1;
  }
  // consistency of the rows that were not pivoted
  double bad = 0.0, scale = 1.0;
  for (int i = l; i < ne; i += kWave) { scale = fmax(scale, fabs(i < rank ? w.C[i][NY] : 0.0)); if (i >= rank) bad = fmax(bad, fabs(w.C[i][NY])); }
  // the scale of the comparison is the original right-hand side; the eliminated one is what is at hand (same order of magnitude)
  bad = wave_max(bad); scale = wave_max(scale);
  int status = bad > 10.0 * kWbcFeasTol * scale ? 1 : 0;
  // back substitution: U11 Y = [rhs | U12], one right-hand side per lane
  const int nz = ny - rank;
  {
    for (int cidx = l; cidx <= nz; cidx += kWave) {           // cidx == 0: the particular solution, 1..nz: null-space columns
      double sol[NE];
      for (int i = rank - 1; i >= 0; --i) {
        double t = cidx == 0 ? w.C[i][NY] : -w.C[i][rank + cidx - 1];
        for (int j = i + 1; j < rank; ++j) t -= w.C[i][j] * sol[j];
        sol[i] = t / w.C[i][i];
      }
      if (cidx == 0) {
        for (int i = 0; i < ny; ++i) w.yp[w.colperm[i]] = i < rank ? sol[i] : 0.0;
      } else {
        for (int i = 0; i < ny; ++i) w.Z[w.colperm[i]][cidx - 1] = i < rank ? sol[i] : (i == rank + cidx - 1 ? 1.0 : 0.0);
      }
    }
  }
  lds_wave_sync();
  // reduced problem: Hz = Z'HZ, gz = Z'(H yp + gy), Dz = D Z, fz = fi - D yp
  {
    // t = H yp + gy  (kept in w.y)
    if (l < ny) { double t = w.gy[l]; for (int j = 0; j < ny; ++j) t += w.H[l][j] * w.yp[j]; w.y[l] = t; }
    // HZ into K (ny x nz fits: NK >= NY is not guaranteed, so Hz is formed as sum over k of Z[k][i] * (H Z)[k][j] with (H Z) recomputed)
    lds_wave_sync();
    for (int idx = l; idx < nz * (nz + 1); idx += kWave) {
      const int i = idx / (nz + 1), j = idx % (nz + 1);
      double acc = 0.0;
      if (j < nz) {
        for (int p = 0; p < ny; ++p) {
          double hz = 0.0;
          for (int qq = 0; qq < ny; ++qq) hz += w.H[p][qq] * w.Z[qq][j];
          acc += w.Z[p][i] * hz;
        }
        w.Hz[i][j] = acc;
      } else {
        for (int p = 0; p < ny; ++p) acc += w.Z[p][i] * w.y[p];
        w.gz[i] = acc;
      }
    }
    for (int idx = l; idx < ni * (nz + 1); idx += kWave) {
      const int i = idx / (nz + 1), j = idx % (nz + 1);
      double acc = 0.0;
      if (j < nz) { for (int p = 0; p < ny; ++p) acc += w.D[i][p] * w.Z[p][j]; w.Dz[i][j] = acc; }
      else { for (int p = 0; p < ny; ++p) acc += w.D[i][p] * w.yp[p]; w.fz[i] = w.fi[i] - acc; }
    }
  }
  if (l == 0) w.nwork = 0;
  lds_wave_sync();
  // ======================================================= active-set iteration on the reduced problem (dense KKT solves)
  int iters = 0;
  bool done = status != 0;
  while (!done) {
    ++iters;
    const int na = w.nwork, nk = nz + na;
    if (nk > W::NK) { status = 1; break; }
    // K = [Hz Dw'; Dw 0 | -gz; fw]
    for (int idx = l; idx < nk * (nk + 1); idx += kWave) {
      const int i = idx / (nk + 1), j = idx % (nk + 1);
      double val;
      if (i < nz) val = j < nz ? w.Hz[i][j] : (j < nk ? w.Dz[w.work[j - nz]][i] : -w.gz[i]);
      else val = j < nz ? w.Dz[w.work[i - nz]][j] : (j < nk ? 0.0 : w.fz[w.work[i - nz]]);
      w.K[i][j] = val;
    }
    lds_wave_sync();
    // Gaussian elimination with partial pivoting (the KKT matrix is indefinite)
    bool singular = false;
    for (int k = 0; k < nk; ++k) {
      double best = -1.0; int where = 0x7fffffff;
      for (int i = k + l; i < nk; i += kWave) { const double v = fabs(w.K[i][k]); if (v > best) { best = v; where = i; } }
      const double gmax = wave_max(best);
      const int pi = wave_min_int(best == gmax ? where : 0x7fffffff);
      if (!(gmax > 1e-14)) { singular = true; break; }
      lds_wave_sync();
      if (pi != k) for (int c = l; c <= nk; c += kWave) { const double t = w.K[k][c]; w.K[k][c] = w.K[pi][c]; w.K[pi][c] = t; }
      lds_wave_sync();
      const double inv = 1.0 / w.K[k][k];
      for (int idx = l; idx < (nk - k - 1) * (nk - k); idx += kWave) {
        const int i = k + 1 + idx / (nk - k), j = k + 1 + idx % (nk - k);
        w.K[i][j] -= w.K[i][k] * inv * w.K[k][j];
      }
      lds_wave_sync();
    }
    if (singular) { status = 1; break; }
    if (l == 0) {
      double sol[W::NK];
      for (int i = nk - 1; i >= 0; --i) {
        double t = w.K[i][nk];
        for (int j = i + 1; j < nk; ++j) t -= w.K[i][j] * sol[j];
        sol[i] = t / w.K[i][i];
      }
      for (int i = 0; i < nz; ++i) w.z[i] = sol[i];
      for (int i = 0; i < na; ++i) w.mu[i] = sol[nz + i];
    }
    lds_wave_sync();
    // most violated inactive inequality
    double worst = -1e300; int wi = 0x7fffffff;
    for (int i = l; i < ni; i += kWave) {
      bool active = false;
      for (int k = 0; k < na; ++k) active |= (w.work[k] == i);
      if (active) continue;
      double t = -w.fz[i];
      for (int j = 0; j < nz; ++j) t += w.Dz[i][j] * w.z[j];
      if (t > worst) { worst = t; wi = i; }
    }
    const double gworst = wave_max(worst);
    const int add = wave_min_int(worst == gworst ? wi : 0x7fffffff);
    if (gworst > kWbcActiveTol) {
      if (iters > st.max_working_set_changes) { status = 1; break; }
      if (l == 0) { w.work[w.nwork] = add; w.nwork = na + 1; }
      lds_wave_sync();
      continue;
    }
    // most negative multiplier of an active inequality
    int drop = -1; double mneg = -kWbcActiveTol;
    for (int k = 0; k < na; ++k) if (w.mu[k] < mneg) { mneg = w.mu[k]; drop = k; }
    if (drop >= 0) {
      if (iters > st.max_working_set_changes) { status = 1; break; }
      lds_wave_sync();
      if (l == 0) { for (int k = drop; k + 1 < na; ++k) w.work[k] = w.work[k + 1]; w.nwork = na - 1; }
      lds_wave_sync();
      continue;
    }
    done = true;
  }
  lds_wave_sync();
  // ======================================================= recover [vdot, F, tau]
  if (status == 0) {
    if (l < ny) { double t = w.yp[l]; for (int j = 0; j < nz; ++j) t += w.Z[l][j] * w.z[j]; w.y[l] = t; }
    lds_wave_sync();
    double* out = a.sol + (size_t)b * n;
    if (l < NV) out[l] = w.y[l];
    if (l < 12) { const int ci = l / 3; out[NV + l] = w.flag[ci] ? w.y[w.ycol[ci] + l % 3] : 0.0; }
    if (l < NJ) {
      double t = w.nle[6 + l];
      for (int g = 0; g < NV; ++g) t += w.M[6 + l][g] * w.y[g];
      for (int ci = 0; ci < kNumContacts; ++ci)
        if (w.flag[ci]) for (int ax = 0; ax < 3; ++ax) t -= w.J[3 * ci + ax][6 + l] * w.y[w.ycol[ci] + ax];
      out[NV + 12 + l] = t;
    }
  }
  if (l == 0) a.status[b] = status;
  if (a.debug) {
    double* dbg = a.debug + (size_t)b * kWbcDebugStride;
    for (int idx = l; idx < NV * NV; idx += kWave) dbg[idx] = w.M[idx / NV][idx % NV];
    if (l < NV) dbg[NV * NV + l] = w.nle[l];
    for (int idx = l; idx < 12 * NV; idx += kWave) dbg[NV * NV + NV + idx] = w.J[idx / NV][idx % NV];
    if (l < 12) dbg[NV * NV + NV + 12 * NV + l] = w.djv[l];
    if (l < 6) dbg[NV * NV + NV + 12 * NV + 12 + l] = w.base_b[l];
    if (l == 0) { dbg[NV * NV + NV + 12 * NV + 18] = (double)rank; dbg[NV * NV + NV + 12 * NV + 19] = (double)iters; dbg[NV * NV + NV + 12 * NV + 20] = (double)w.nwork; }
    if (l < w.nwork) dbg[NV * NV + NV + 12 * NV + 24 + l] = (double)w.work[l];
  }
}

}  // namespace bpmpc
