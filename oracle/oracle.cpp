// ORACLE — test infrastructure, NOT product code.  See oracle.h for scope, provenance and parity status
// (UNPINNED: no reference golden vectors exist for this path; pinned by FD / known answers / invariants).
//
// Citations: paths are relative to /root/reference; [OCS2-upstream] marks behaviour of the un-vendored
// dependencies (leggedrobotics/ocs2 main 2023, Pinocchio, Eigen 3.3/3.4 FullPivLU, HPIPM) restated from their
// published algorithms.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int MAXJ = 12;
constexpr int MAXB = MAXJ + 1;
constexpr int MAXG = 6 + MAXJ;
constexpr int MAXX = 12 + MAXJ;
constexpr int NCMAX = 16;
constexpr double GRAVITY = 9.81;  // [OCS2-upstream] ModelHelperFunctions: gravity (0,0,-9.81); utils.h:67 uses 9.81 too

struct Model {
  int nj, nx, nu;
  int parent[MAXB];
  double Rfix[MAXB][9], pfix[MAXB][3], axis[MAXB][3];
  double mass[MAXB], com[MAXB][3], inertia[MAXB][9];
  int cbody[4];
  double coff[4][3];
  std::vector<double> Q, R;
  double mu, reg, grip, shift, bmu, bdelta, gain, robot_mass;
  // useHardFrictionConeConstraint (src/BipedalRobotInterface.cpp:68-69,181-182): the cone is an inequality constraint of the problem;
  // [OCS2-upstream, recalled] the SQP solver turns inequality constraints into a relaxed-barrier penalty of their LINEAR approximation
  // with sqp.inequalityConstraintMu / Delta (task.info:74-75) - see node_lq
  int hard = 0;
  double imu = 0.0, idelta = 1e-6;
  bool anc[MAXB][MAXB];  // anc[j][b]: joint j (1..nj) lies on the path base -> body b (inclusive)
};

// ------------------------------------------------------------------------------------------------
// Operation counter (SURVEY.md section 8(d) "algorithmic flops ... instrument the templated scalar").  Only the separate build
// liboracle_count.so (-DORACLE_COUNT_FLOPS) counts; in liboracle.so - the checker and the timed cpu_baseline - FL() is nothing.
// One unit = one double-precision add, multiply, divide, square root or sin / cos evaluation of THIS restatement (forward-mode AD
// over all nx + nu directions); an analytic or sparse differentiation needs far fewer, see DESIGN.md.
// ------------------------------------------------------------------------------------------------
#ifdef ORACLE_COUNT_FLOPS
static unsigned long long g_flops = 0;
#define FL(n) (g_flops += static_cast<unsigned long long>(n))
#else
#define FL(n) ((void)0)
#endif

// counting scalar for value-only evaluations of the templated functions
struct CD {
  double v;
  CD() : v(0.0) {}
  CD(double a) : v(a) {}  // NOLINT
};
inline CD operator+(CD a, CD b) { FL(1); return CD(a.v + b.v); }
inline CD operator-(CD a, CD b) { FL(1); return CD(a.v - b.v); }
inline CD operator-(CD a) { return CD(-a.v); }
inline CD operator*(CD a, CD b) { FL(1); return CD(a.v * b.v); }
inline CD operator/(CD a, CD b) { FL(1); return CD(a.v / b.v); }
inline CD& operator+=(CD& a, CD b) { a = a + b; return a; }
inline CD& operator-=(CD& a, CD b) { a = a - b; return a; }
inline CD sin(CD a) { FL(1); return CD(std::sin(a.v)); }
inline CD cos(CD a) { FL(1); return CD(std::cos(a.v)); }
inline double val(CD a) { return a.v; }

// ------------------------------------------------------------------------------------------------
// forward-mode dual numbers
// ------------------------------------------------------------------------------------------------
template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  Dual(double a) : v(a) { for (int i = 0; i < N; ++i) d[i] = 0.0; }  // NOLINT
};
template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { FL(1 + N); Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { FL(1 + N); Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { FL(1 + 3 * N); Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { FL(2 + 3 * N); Dual<N> r; const double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { FL(1); Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double b, const Dual<N>& a) { return a + b; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { FL(1); Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double b, const Dual<N>& a) { return (-a) + b; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) { FL(1 + N); Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> inline Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { FL(1); return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> inline Dual<N> sin(const Dual<N>& a) { FL(2 + N); Dual<N> r; r.v = std::sin(a.v); const double c = std::cos(a.v); for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i]; return r; }
template <int N> inline Dual<N> cos(const Dual<N>& a) { FL(2 + N); Dual<N> r; r.v = std::cos(a.v); const double s = -std::sin(a.v); for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
inline double val(double a) { return a; }
template <int N> inline double val(const Dual<N>& a) { return a.v; }

using std::cos;
using std::sin;

// ------------------------------------------------------------------------------------------------
// small templated vector helpers
// ------------------------------------------------------------------------------------------------
template <class S> inline void cross(const S* a, const S* b, S* c) {
  S c0 = a[1] * b[2] - a[2] * b[1];
  S c1 = a[2] * b[0] - a[0] * b[2];
  S c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
template <class S> inline void matvec3(const S* R, const S* v, S* out) {  // out = R v
  S o0 = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  S o1 = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  S o2 = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  out[0] = o0; out[1] = o1; out[2] = o2;
}
template <class S> inline void matTvec3(const S* R, const S* v, S* out) {  // out = R^T v
  S o0 = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  S o1 = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  S o2 = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  out[0] = o0; out[1] = o1; out[2] = o2;
}
template <class S> inline void matmul3(const S* A, const S* B, S* C) {
  S t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  for (int i = 0; i < 9; ++i) C[i] = t[i];
}

// ------------------------------------------------------------------------------------------------
// kinematics  ([OCS2-upstream] createPinocchioInterface / Pinocchio conventions)
// ------------------------------------------------------------------------------------------------
template <class S>
struct Kin {
  S R[MAXB][9], o[MAXB][3], ahat[MAXB][3], cw[MAXB][3];
  S Sz[9];  // body-frame ZYX rate map S(theta): omega_body = S thetadot   
  S com[3];
  S cpos[4][3];
  S A[6][MAXG];  // centroidal momentum matrix
};

template <class S>
void kinematics(const Model& m, const S* q, Kin<S>& k) {
  const S sy = sin(q[3]), cy = cos(q[3]), sp = sin(q[4]), cp = cos(q[4]), sr = sin(q[5]), cr = cos(q[5]);
  S* R0 = k.R[0];
  R0[0] = cy * cp; R0[1] = cy * sp * sr - sy * cr; R0[2] = cy * sp * cr + sy * sr;
  R0[3] = sy * cp; R0[4] = sy * sp * sr + cy * cr; R0[5] = sy * sp * cr - cy * sr;
  R0[6] = -sp;     R0[7] = cp * sr;                R0[8] = cp * cr;
  k.Sz[0] = -sp;     k.Sz[1] = S(0.0); k.Sz[2] = S(1.0);
  k.Sz[3] = cp * sr; k.Sz[4] = cr;     k.Sz[5] = S(0.0);
  k.Sz[6] = cp * cr; k.Sz[7] = -sr;    k.Sz[8] = S(0.0);
  for (int i = 0; i < 3; ++i) k.o[0][i] = q[i];
  for (int j = 1; j <= m.nj; ++j) {
    const int lam = m.parent[j];
    const double* a = m.axis[j];
    const S s = sin(q[5 + j]), c = cos(q[5 + j]);
    const S omc = S(1.0) - c;
    S Rot[9];  // Rodrigues
    Rot[0] = c + omc * (a[0] * a[0]);        Rot[1] = omc * (a[0] * a[1]) - s * a[2]; Rot[2] = omc * (a[0] * a[2]) + s * a[1];
    Rot[3] = omc * (a[1] * a[0]) + s * a[2]; Rot[4] = c + omc * (a[1] * a[1]);        Rot[5] = omc * (a[1] * a[2]) - s * a[0];
    Rot[6] = omc * (a[2] * a[0]) - s * a[1]; Rot[7] = omc * (a[2] * a[1]) + s * a[0]; Rot[8] = c + omc * (a[2] * a[2]);
    S Rf[9], E[9], pf[3], av[3];
    for (int i = 0; i < 9; ++i) Rf[i] = S(m.Rfix[j][i]);
    for (int i = 0; i < 3; ++i) { pf[i] = S(m.pfix[j][i]); av[i] = S(a[i]); }
    matmul3(Rf, Rot, E);
    matmul3(k.R[lam], E, k.R[j]);
    S t[3];
    matvec3(k.R[lam], pf, t);
    for (int i = 0; i < 3; ++i) k.o[j][i] = k.o[lam][i] + t[i];
    matvec3(k.R[j], av, k.ahat[j]);
  }
  S msum(0.0), mc[3] = {S(0.0), S(0.0), S(0.0)};
  for (int b = 0; b <= m.nj; ++b) {
    S cb[3] = {S(m.com[b][0]), S(m.com[b][1]), S(m.com[b][2])}, t[3];
    matvec3(k.R[b], cb, t);
    for (int i = 0; i < 3; ++i) { k.cw[b][i] = k.o[b][i] + t[i]; mc[i] += k.cw[b][i] * m.mass[b]; }
    msum += S(m.mass[b]);
  }
  for (int i = 0; i < 3; ++i) k.com[i] = mc[i] / msum;
  for (int c = 0; c < 4; ++c) {
    const int b = m.cbody[c];
    S off[3] = {S(m.coff[c][0]), S(m.coff[c][1]), S(m.coff[c][2])}, t[3];
    matvec3(k.R[b], off, t);
    for (int i = 0; i < 3; ++i) k.cpos[c][i] = k.o[b][i] + t[i];
  }
}

// Centroidal momentum matrix by its definition: column g = momentum (world aligned, about the CoM) produced by a
// unit generalised velocity g.  [OCS2-upstream] pinocchio::computeCentroidalMap yields the same matrix (data.Ag).
template <class S>
void cmm(const Model& m, Kin<S>& k) {
  const int G = 6 + m.nj;
  for (int g = 0; g < G; ++g) {
    S w[3] = {S(0.0), S(0.0), S(0.0)};
    const S* org = k.o[0];
    int joint = 0;
    if (g >= 3 && g < 6) {
      S col[3] = {k.Sz[g - 3], k.Sz[3 + g - 3], k.Sz[6 + g - 3]};
      matvec3(k.R[0], col, w);
    } else if (g >= 6) {
      joint = g - 5;
      for (int i = 0; i < 3; ++i) w[i] = k.ahat[joint][i];
      org = k.o[joint];
    }
    S lin[3] = {S(0.0), S(0.0), S(0.0)}, ang[3] = {S(0.0), S(0.0), S(0.0)};
    for (int b = 0; b <= m.nj; ++b) {
      if (joint > 0 && !m.anc[joint][b]) continue;
      S cdot[3];
      if (g < 3) {
        cdot[0] = S(g == 0 ? 1.0 : 0.0); cdot[1] = S(g == 1 ? 1.0 : 0.0); cdot[2] = S(g == 2 ? 1.0 : 0.0);
      } else {
        S r[3] = {k.cw[b][0] - org[0], k.cw[b][1] - org[1], k.cw[b][2] - org[2]};
        cross(w, r, cdot);
      }
      S d[3] = {k.cw[b][0] - k.com[0], k.cw[b][1] - k.com[1], k.cw[b][2] - k.com[2]}, t[3];
      cross(d, cdot, t);
      for (int i = 0; i < 3; ++i) { lin[i] += cdot[i] * m.mass[b]; ang[i] += t[i] * m.mass[b]; }
      if (g >= 3) {  // I_b^world w = R (I (R^T w))
        S wl[3], Iw[3], Ib[9];
        for (int i = 0; i < 9; ++i) Ib[i] = S(m.inertia[b][i]);
        matTvec3(k.R[b], w, wl);
        matvec3(Ib, wl, Iw);
        matvec3(k.R[b], Iw, t);
        for (int i = 0; i < 3; ++i) ang[i] += t[i];
      }
    }
    for (int i = 0; i < 3; ++i) { k.A[i][g] = lin[i]; k.A[3 + i][g] = ang[i]; }
  }
}

template <class S>
void inverse3(const S* M, S* inv) {  // cofactor formula (what Eigen uses for fixed 3x3)
  const S c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const S det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  const S id = S(1.0) / det;
  inv[0] = c00 * id; inv[1] = (M[2] * M[7] - M[1] * M[8]) * id; inv[2] = (M[1] * M[5] - M[2] * M[4]) * id;
  inv[3] = c01 * id; inv[4] = (M[0] * M[8] - M[2] * M[6]) * id; inv[5] = (M[2] * M[3] - M[0] * M[5]) * id;
  inv[6] = c02 * id; inv[7] = (M[1] * M[6] - M[0] * M[7]) * id; inv[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

// base velocity from normalised momentum and joint velocities:
// [OCS2-upstream] CentroidalModelPinocchioMapping::getPinocchioJointVelocity +
// computeFloatingBaseCentroidalMomentumMatrixInverse  
template <class S>
void base_velocity(const Model& m, const Kin<S>& k, const S* x, const S* u, S* vb) {
  S rhs[6];
  for (int i = 0; i < 6; ++i) {
    S acc = x[i] * m.robot_mass;
    for (int j = 0; j < m.nj; ++j) acc -= k.A[i][6 + j] * u[12 + j];
    rhs[i] = acc;
  }
  S Ab22[9], inv[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ab22[3 * i + j] = k.A[3 + i][3 + j];
  inverse3(Ab22, inv);
  S th[3];
  matvec3(inv, rhs + 3, th);
  const S mass = k.A[0][0];
  for (int i = 0; i < 3; ++i) {
    S t = rhs[i];
    for (int j = 0; j < 3; ++j) t -= k.A[i][3 + j] * th[j];
    vb[i] = t / mass;
    vb[3 + i] = th[i];
  }
}

// a1: [OCS2-upstream] PinocchioCentroidalDynamicsAD::getValueCppAd (wrapper: src/dynamics/BipedalRobotDynamicsAD.cpp:46-56)
template <class S>
void flow_map(const Model& m, const S* x, const S* u, S* f) {
  Kin<S> k;
  kinematics(m, x + 6, k);
  cmm(m, k);
  S lin[3] = {S(0.0), S(0.0), S(-GRAVITY * m.robot_mass)}, ang[3] = {S(0.0), S(0.0), S(0.0)};
  for (int c = 0; c < 4; ++c) {
    const S* F = u + 3 * c;
    S r[3] = {k.cpos[c][0] - k.com[0], k.cpos[c][1] - k.com[1], k.cpos[c][2] - k.com[2]}, t[3];
    cross(r, F, t);
    for (int i = 0; i < 3; ++i) { lin[i] += F[i]; ang[i] += t[i]; }
  }
  for (int i = 0; i < 3; ++i) { f[i] = lin[i] / m.robot_mass; f[3 + i] = ang[i] / m.robot_mass; }
  base_velocity(m, k, x, u, f + 6);
  for (int j = 0; j < m.nj; ++j) f[12 + j] = u[12 + j];
}

// a6: [OCS2-upstream] PinocchioEndEffectorKinematicsCppAd::{getPositionCppAd,getVelocityCppAd}: world position and
// LOCAL_WORLD_ALIGNED linear velocity of the contact frames, v = mapping.getPinocchioJointVelocity(x,u)
// (built at src/BipedalRobotInterface.cpp:169-178).  Computed here by a body-twist forward pass.
template <class S>
void ee_kinematics(const Model& m, const S* x, const S* u, S pos[4][3], S vel[4][3]) {
  Kin<S> k;
  kinematics(m, x + 6, k);
  cmm(m, k);
  S vb[6];
  base_velocity(m, k, x, u, vb);
  S om[MAXB][3], vo[MAXB][3];
  S wb[3];
  matvec3(k.Sz, vb + 3, wb);
  matvec3(k.R[0], wb, om[0]);
  for (int i = 0; i < 3; ++i) vo[0][i] = vb[i];
  for (int j = 1; j <= m.nj; ++j) {
    const int lam = m.parent[j];
    S r[3] = {k.o[j][0] - k.o[lam][0], k.o[j][1] - k.o[lam][1], k.o[j][2] - k.o[lam][2]}, t[3];
    cross(om[lam], r, t);
    for (int i = 0; i < 3; ++i) { vo[j][i] = vo[lam][i] + t[i]; om[j][i] = om[lam][i] + k.ahat[j][i] * u[11 + j]; }
  }
  for (int c = 0; c < 4; ++c) {
    const int b = m.cbody[c];
    S r[3] = {k.cpos[c][0] - k.o[b][0], k.cpos[c][1] - k.o[b][1], k.cpos[c][2] - k.o[b][2]}, t[3];
    cross(om[b], r, t);
    for (int i = 0; i < 3; ++i) { pos[c][i] = k.cpos[c][i]; vel[c][i] = vo[b][i] + t[i]; }
  }
}

template <int N>
void seed(const Model& m, const double* x, const double* u, Dual<N>* xd, Dual<N>* ud) {
  for (int i = 0; i < m.nx; ++i) { xd[i] = Dual<N>(x[i]); xd[i].d[i] = 1.0; }
  for (int i = 0; i < m.nu; ++i) { ud[i] = Dual<N>(u[i]); ud[i].d[m.nx + i] = 1.0; }
}

template <int N>
void flow_map_lin_t(const Model& m, const double* x, const double* u, double* f, double* A, double* B) {
  Dual<N> xd[MAXX], ud[MAXX], fd[MAXX];
  seed<N>(m, x, u, xd, ud);
  flow_map(m, xd, ud, fd);
  for (int i = 0; i < m.nx; ++i) {
    f[i] = fd[i].v;
    for (int j = 0; j < m.nx; ++j) A[i * m.nx + j] = fd[i].d[j];
    for (int j = 0; j < m.nu; ++j) B[i * m.nu + j] = fd[i].d[m.nx + j];
  }
}
void flow_map_lin(const Model& m, const double* x, const double* u, double* f, double* A, double* B) {
  if (m.nj == 10) flow_map_lin_t<44>(m, x, u, f, A, B); else flow_map_lin_t<48>(m, x, u, f, A, B);
}

template <int N>
void ee_lin_t(const Model& m, const double* x, const double* u, double* pos, double* vel, double* dpdx, double* dvdx, double* dvdu) {
  Dual<N> xd[MAXX], ud[MAXX], p[4][3], v[4][3];
  seed<N>(m, x, u, xd, ud);
  ee_kinematics(m, xd, ud, p, v);
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 3; ++i) {
      const int r = 3 * c + i;
      pos[r] = p[c][i].v;
      vel[r] = v[c][i].v;
      for (int j = 0; j < m.nx; ++j) { if (dpdx) dpdx[r * m.nx + j] = p[c][i].d[j]; if (dvdx) dvdx[r * m.nx + j] = v[c][i].d[j]; }
      for (int j = 0; j < m.nu; ++j) if (dvdu) dvdu[r * m.nu + j] = v[c][i].d[m.nx + j];
    }
}
void ee_lin(const Model& m, const double* x, const double* u, double* pos, double* vel, double* dpdx, double* dvdx, double* dvdu) {
  if (m.nj == 10) ee_lin_t<44>(m, x, u, pos, vel, dpdx, dvdx, dvdu); else ee_lin_t<48>(m, x, u, pos, vel, dpdx, dvdx, dvdu);
}

// ------------------------------------------------------------------------------------------------
// per-node terms
// ------------------------------------------------------------------------------------------------
// include/ocs2_bipedal_robot/gait/MotionPhaseDefinition.h:57-76
inline void mode_flags(int mode, bool* f) {
  f[0] = f[1] = (mode == 1 || mode == 3);
  f[2] = f[3] = (mode == 2 || mode == 3);
}
// include/ocs2_bipedal_robot/common/utils.h:49-76
void weight_compensating_input(const Model& m, const bool* flags, double* u) {
  int n = 0;
  for (int i = 0; i < 4; ++i) n += flags[i] ? 1 : 0;
  for (int i = 0; i < m.nu; ++i) u[i] = 0.0;
  if (n > 0) {
    const double totalWeight = m.robot_mass * 9.81;
    for (int i = 0; i < 4; ++i)
      if (flags[i]) u[3 * i + 2] = totalWeight / n;
  }
}
// [OCS2-upstream] RelaxedBarrierPenalty
inline void relaxed_barrier(double mu, double delta, double h, double* p, double* dp, double* ddp) {
  if (!(mu > 0.0)) { *p = 0.0; *dp = 0.0; *ddp = 0.0; return; }      // no penalty configured ([OCS2-upstream] sqp.inequalityConstraintMu defaults to 0)
  if (h > delta) {
    *p = -mu * std::log(h); *dp = -mu / h; *ddp = mu / (h * h);
  } else {
    const double t = (h - 2.0 * delta) / delta;
    *p = mu * (-std::log(delta) + 0.5 * t * t - 0.5); *dp = mu * ((h - 2.0 * delta) / (delta * delta)); *ddp = mu / (delta * delta);
  }
}
// src/constraint/FrictionConeConstraint.cpp:129-160 (t_R_w = I)
inline double cone_value(const Model& m, const double* F) {
  return m.mu * (F[2] + m.grip) - std::sqrt(F[0] * F[0] + F[1] * F[1] + m.reg);
}

// RK2 value-only discretisation [OCS2-upstream ocs2_core/integration/SensitivityIntegratorImpl / Integrator rk2]
void rk2_value(const Model& m, double dt, const double* x, const double* u, double* xn) {
  double f1[MAXX], f2[MAXX], x2[MAXX];
  flow_map<double>(m, x, u, f1);
  for (int i = 0; i < m.nx; ++i) x2[i] = x[i] + dt * f1[i];
  flow_map<double>(m, x2, u, f2);
  for (int i = 0; i < m.nx; ++i) xn[i] = x[i] + 0.5 * dt * f1[i] + 0.5 * dt * f2[i];
}

struct NodeLQ {
  int nc = 0;
  std::vector<double> A, B, b, Q, R, P, q, r, C, D, e;
  double c = 0.0;
  double perf[3] = {0, 0, 0};
  void resize(int nx, int nu) {
    A.assign(nx * nx, 0.0); B.assign(nx * nu, 0.0); b.assign(nx, 0.0); Q.assign(nx * nx, 0.0); R.assign(nu * nu, 0.0);
    P.assign(nu * nx, 0.0); q.assign(nx, 0.0); r.assign(nu, 0.0); C.assign(NCMAX * nx, 0.0); D.assign(NCMAX * nu, 0.0); e.assign(NCMAX, 0.0);
    c = 0.0; nc = 0;
  }
};

// cost value / constraint values shared by the LQ and the value-only paths
double node_cost_value(const Model& m, const double* x, const double* u, const double* xref, const bool* flags) {
  const int nx = m.nx, nu = m.nu;
  double dx[MAXX], du[MAXX], unom[MAXX];
  weight_compensating_input(m, flags, unom);
  for (int i = 0; i < nx; ++i) dx[i] = x[i] - xref[i];
  for (int i = 0; i < nu; ++i) du[i] = u[i] - unom[i];
  double c = 0.0;
  for (int i = 0; i < nx; ++i) { double t = 0; for (int j = 0; j < nx; ++j) t += m.Q[i * nx + j] * dx[j]; c += 0.5 * dx[i] * t; }
  for (int i = 0; i < nu; ++i) { double t = 0; for (int j = 0; j < nu; ++j) t += m.R[i * nu + j] * du[j]; c += 0.5 * du[i] * t; }
  for (int k = 0; k < 4; ++k)
    if (flags[k]) {
      double p, dp, ddp;
      relaxed_barrier(m.hard ? m.imu : m.bmu, m.hard ? m.idelta : m.bdelta, cone_value(m, u + 3 * k), &p, &dp, &ddp);
      c += p;
    }
  return c;
}

void node_perf(const Model& m, int kind, double dt, const double* x, const double* u, const double* xnext, const double* xref, int mode,
               const double* zref, const double* zdref, double* perf) {
  const int nx = m.nx;
  if (kind == 1) {  // [OCS2-upstream] multiple_shooting::computeEventPerformance: identity jump map
    double s = 0;
    for (int i = 0; i < nx; ++i) { const double d = x[i] - xnext[i]; s += d * d; }
    perf[0] = 0; perf[1] = s; perf[2] = 0;
    return;
  }
  bool flags[4];
  mode_flags(mode, flags);
  double xn[MAXX];
  rk2_value(m, dt, x, u, xn);
  double dyn = 0;
  for (int i = 0; i < nx; ++i) { const double d = xn[i] - xnext[i]; dyn += d * d; }
  double pos[4][3], vel[4][3];
  ee_kinematics<double>(m, x, u, pos, vel);
  double eq = 0;
  for (int k = 0; k < 4; ++k) {
    if (!flags[k]) {
      for (int i = 0; i < 3; ++i) eq += u[3 * k + i] * u[3 * k + i];
      double g = vel[k][2] - zdref[k];
      if (m.gain != 0.0) g += m.gain * (pos[k][2] - zref[k]);
      eq += g * g;
    } else {
      for (int i = 0; i < 3; ++i) {
        double g = vel[k][i];
        if (m.gain != 0.0 && i == 2) g += m.gain * pos[k][2];
        eq += g * g;
      }
    }
  }
  perf[0] = dt * node_cost_value(m, x, u, xref, flags);
  perf[1] = dt * dyn;
  perf[2] = dt * eq;
}

// [OCS2-upstream] multiple_shooting::setupIntermediateNode / setupEventNode / computeIntermediatePerformance
void node_lq(const Model& m, int kind, double dt, const double* x, const double* u, const double* xnext, const double* xref, int mode,
             const double* zref, const double* zdref, NodeLQ& o) {
  const int nx = m.nx, nu = m.nu;
  o.resize(nx, nu);
  if (kind == 1) {
    for (int i = 0; i < nx; ++i) { o.A[i * nx + i] = 1.0; o.b[i] = x[i] - xnext[i]; }
    double s = 0;
    for (int i = 0; i < nx; ++i) s += o.b[i] * o.b[i];
    o.perf[0] = 0; o.perf[1] = s; o.perf[2] = 0;
    return;
  }
  bool flags[4];
  mode_flags(mode, flags);
  // --- dynamics: RK2 sensitivity discretiser
  std::vector<double> A1(nx * nx), B1(nx * nu), A2(nx * nx), B2(nx * nu);
  double f1[MAXX], f2[MAXX], x2[MAXX];
  flow_map_lin(m, x, u, f1, A1.data(), B1.data());
  for (int i = 0; i < nx; ++i) x2[i] = x[i] + dt * f1[i];
  flow_map_lin(m, x2, u, f2, A2.data(), B2.data());
  const double h = 0.5 * dt;
  for (int i = 0; i < nx; ++i) {
    for (int j = 0; j < nx; ++j) {
      double t = 0;
      for (int l = 0; l < nx; ++l) t += A2[i * nx + l] * A1[l * nx + j];
      FL(2 * nx + 5);
      o.A[i * nx + j] = (i == j ? 1.0 : 0.0) + h * (A1[i * nx + j] + A2[i * nx + j] + dt * t);
    }
    for (int j = 0; j < nu; ++j) {
      double t = 0;
      for (int l = 0; l < nx; ++l) t += A2[i * nx + l] * B1[l * nu + j];
      FL(2 * nx + 4);
      o.B[i * nu + j] = h * (B1[i * nu + j] + B2[i * nu + j] + dt * t);
    }
    o.b[i] = x[i] + h * f1[i] + h * f2[i] - xnext[i];
  }
  // --- cost: tracking (include/.../cost/BipedalRobotQuadraticTrackingCost.h:57-63, [OCS2-upstream] QuadraticStateInputCost)
  double dx[MAXX], du[MAXX], unom[MAXX];
  weight_compensating_input(m, flags, unom);
  for (int i = 0; i < nx; ++i) dx[i] = x[i] - xref[i];
  for (int i = 0; i < nu; ++i) du[i] = u[i] - unom[i];
  o.Q = m.Q; o.R = m.R;
  FL(2 * nx + 2 * nu + (2 * nx + 3) * nx + (2 * nu + 3) * nu + 4 * 60 + nx * nx + nu * nu + nx + nu);   // b, dx, du, two mat-vecs + cost, four cones, dt scaling
  double c = 0;
  for (int i = 0; i < nx; ++i) { double t = 0; for (int j = 0; j < nx; ++j) t += m.Q[i * nx + j] * dx[j]; o.q[i] = t; c += 0.5 * dx[i] * t; }
  for (int i = 0; i < nu; ++i) { double t = 0; for (int j = 0; j < nu; ++j) t += m.R[i * nu + j] * du[j]; o.r[i] = t; c += 0.5 * du[i] * t; }
  // --- soft friction cones (src/constraint/FrictionConeConstraint.cpp:96-206 through [OCS2-upstream]
  //     StateInputSoftConstraint + MultidimensionalPenalty::getQuadraticApproximation + RelaxedBarrierPenalty)
  for (int k = 0; k < 4; ++k) {
    if (!flags[k]) continue;
    const double* F = u + 3 * k;
    const double Fx2 = F[0] * F[0], Fy2 = F[1] * F[1];
    const double T2 = Fx2 + Fy2 + m.reg, T = std::sqrt(T2), T32 = T * T2;
    const double hval = m.mu * (F[2] + m.grip) - T;
    const double g[3] = {-F[0] / T, -F[1] / T, m.mu};
    double H[9] = {-(Fy2 + m.reg) / T32, F[0] * F[1] / T32, 0, F[0] * F[1] / T32, -(Fx2 + m.reg) / T32, 0, 0, 0, 0};
    double p, dp, ddp;
    if (m.hard) {
      // Hard cone = inequality constraint of the OCP (BipedalRobotInterface.cpp:181-182).  [OCS2-upstream, recalled] ocs2_sqp handles
      // state-input inequality constraints as a penalty: multiple_shooting::setupIntermediateNode takes their LINEAR approximation
      // h + dh/du du >= 0 and adds RelaxedBarrierPenalty(sqp.inequalityConstraintMu, sqp.inequalityConstraintDelta) of it to the cost
      // (penaltyCostQuadraticApproximation: value p(h), gradient p'(h) dh, Gauss-Newton Hessian p''(h) dh dh'), times dt like the rest of
      // the intermediate cost and BEFORE the projection.  The constraint's own second derivative and its hessianDiagonalShift
      // (FrictionConeConstraint.cpp:163-206) belong to getQuadraticApproximation, which this path never calls.
      relaxed_barrier(m.imu, m.idelta, hval, &p, &dp, &ddp);
      c += p;
      for (int i = 0; i < 3; ++i) {
        o.r[3 * k + i] += dp * g[i];
        for (int j = 0; j < 3; ++j) o.R[(3 * k + i) * nu + 3 * k + j] += ddp * g[i] * g[j];
      }
      continue;
    }
    relaxed_barrier(m.bmu, m.bdelta, hval, &p, &dp, &ddp);
    c += p;
    for (int i = 0; i < 3; ++i) {
      o.r[3 * k + i] += dp * g[i];
      for (int j = 0; j < 3; ++j) o.R[(3 * k + i) * nu + 3 * k + j] += ddp * g[i] * g[j] + dp * H[3 * i + j];
    }
    for (int i = 0; i < nu; ++i) o.R[i * nu + i] += dp * (-m.shift);  // FrictionConeConstraint.cpp:195 (all nu diagonals)
    for (int i = 0; i < nx; ++i) o.Q[i * nx + i] += dp * (-m.shift);  // :202 (all nx diagonals)
  }
  o.c = c * dt;
  for (auto& v : o.Q) v *= dt;
  for (auto& v : o.R) v *= dt;
  for (auto& v : o.q) v *= dt;
  for (auto& v : o.r) v *= dt;
  // --- equality constraints, registration order zeroForce_i, zeroVelocity_i, normalVelocity_i
  //     (src/BipedalRobotInterface.cpp:187-191)
  std::vector<double> dpdx(12 * nx), dvdx(12 * nx), dvdu(12 * nu);
  double pos[12], vel[12];
  ee_lin(m, x, u, pos, vel, dpdx.data(), dvdx.data(), dvdu.data());
  int row = 0;
  for (int k = 0; k < 4; ++k) {
    if (!flags[k]) {  // ZeroForceConstraint.cpp:58-72
      for (int i = 0; i < 3; ++i) { o.D[row * nu + 3 * k + i] = 1.0; o.e[row] = u[3 * k + i]; ++row; }
    }
    if (flags[k]) {  // ZeroVelocityConstraintCppAd + EndEffectorLinearConstraint.cpp:74-111, config BipedalRobotInterface.cpp:350-359
      for (int i = 0; i < 3; ++i) {
        const int s = 3 * k + i;
        o.e[row] = vel[s];
        for (int j = 0; j < nx; ++j) o.C[row * nx + j] = dvdx[s * nx + j];
        for (int j = 0; j < nu; ++j) o.D[row * nu + j] = dvdu[s * nu + j];
        if (m.gain != 0.0 && i == 2) {
          o.e[row] += m.gain * pos[s];
          for (int j = 0; j < nx; ++j) o.C[row * nx + j] += m.gain * dpdx[s * nx + j];
        }
        ++row;
      }
    } else {  // NormalVelocityConstraintCppAd + BipedalRobotPreComputation.cpp:71-80
      const int s = 3 * k + 2;
      o.e[row] = vel[s] - zdref[k];
      for (int j = 0; j < nx; ++j) o.C[row * nx + j] = dvdx[s * nx + j];
      for (int j = 0; j < nu; ++j) o.D[row * nu + j] = dvdu[s * nu + j];
      if (m.gain != 0.0) {
        o.e[row] += m.gain * (pos[s] - zref[k]);
        for (int j = 0; j < nx; ++j) o.C[row * nx + j] += m.gain * dpdx[s * nx + j];
      }
      ++row;
    }
  }
  o.nc = row;
  FL(2 * nx + 2 * row + (m.gain != 0.0 ? 2 * row * nx : 0));
  double dyn = 0, eq = 0;
  for (int i = 0; i < nx; ++i) dyn += o.b[i] * o.b[i];
  for (int i = 0; i < row; ++i) eq += o.e[i] * o.e[i];
  o.perf[0] = o.c; o.perf[1] = dt * dyn; o.perf[2] = dt * eq;
}

// ------------------------------------------------------------------------------------------------
// [OCS2-upstream] LinearAlgebra::luConstraintProjection over Eigen::FullPivLU (restated: complete pivoting with
// first-in-column-major tie break, rank threshold eps*min(rows,cols)*|maxpivot|, solve() with free variables = 0,
// kernel() = Q [-U11^{-1} U12; I]).
// ------------------------------------------------------------------------------------------------
struct Projection {
  int rank = 0, nut = 0;
  std::vector<double> Px, Pu, Pe;
};

void lu_projection(int nc, int nx, int nu, const double* C, const double* D, const double* e, Projection& pr) {
  pr.Px.assign(nu * nx, 0.0); pr.Pe.assign(nu, 0.0); pr.Pu.assign(nu * nu, 0.0);
  if (nc == 0) {
    pr.rank = 0; pr.nut = nu;
    for (int i = 0; i < nu; ++i) pr.Pu[i * nu + i] = 1.0;
    return;
  }
  const int rows = nc, cols = nu, size = std::min(rows, cols);
  std::vector<double> lu(D, D + rows * cols);
  std::vector<int> rowT(size), colT(size);
  int nonzero = size;
  double maxpivot = 0.0;
  for (int k = 0; k < size; ++k) {
    double best = -1.0; int br = k, bc = k;
    for (int j = k; j < cols; ++j)
      for (int i = k; i < rows; ++i) {
        const double a = std::fabs(lu[i * cols + j]);
        if (a > best) { best = a; br = i; bc = j; }
      }
    if (best == 0.0) {
      nonzero = k;
      for (int i = k; i < size; ++i) { rowT[i] = i; colT[i] = i; }
      break;
    }
    if (best > maxpivot) maxpivot = best;
    rowT[k] = br; colT[k] = bc;
    if (br != k) for (int j = 0; j < cols; ++j) std::swap(lu[k * cols + j], lu[br * cols + j]);
    if (bc != k) for (int i = 0; i < rows; ++i) std::swap(lu[i * cols + k], lu[i * cols + bc]);
    if (k < rows - 1) for (int i = k + 1; i < rows; ++i) lu[i * cols + k] /= lu[k * cols + k];
    if (k < size - 1)
      for (int i = k + 1; i < rows; ++i)
        for (int j = k + 1; j < cols; ++j) lu[i * cols + j] -= lu[i * cols + k] * lu[k * cols + j];
  }
  // permutations: P = T_{size-1} ... T_0 (rows), Q = T_0 ... T_{size-1} (columns)
  std::vector<int> p(rows), qidx(cols);
  for (int i = 0; i < rows; ++i) p[i] = i;
  for (int k = size - 1; k >= 0; --k) std::swap(p[k], p[rowT[k]]);  // Eigen: m_p.applyTranspositionOnTheRight(k, rowT[k]) for k descending
  for (int i = 0; i < cols; ++i) qidx[i] = i;
  for (int k = 0; k < size; ++k) std::swap(qidx[k], qidx[colT[k]]);
  // Eigen permutation semantics: (P*b).row(p.indices[i]) = b.row(i);  dst.row(q.indices[i]) = c.row(i)
  const double thr = std::fabs(maxpivot) * (2.220446049250313e-16 * size);
  int rank = 0;
  for (int i = 0; i < nonzero; ++i) rank += (std::fabs(lu[i * cols + i]) > thr) ? 1 : 0;
  pr.rank = rank; pr.nut = nu - rank;
  // solve for rhs = [C | e]  (nc x (nx+1)), result y (nu x (nx+1)), then Px = -y[:, :nx], Pe = -y[:, nx]
  const int nr = nx + 1;
  std::vector<double> c(rows * nr, 0.0);
  for (int i = 0; i < rows; ++i) {
    for (int j = 0; j < nx; ++j) c[p[i] * nr + j] = C[i * nx + j];
    c[p[i] * nr + nx] = e[i];
  }
  if (rank > 0) {
    for (int i = 0; i < size; ++i)  // unit lower solve on top-left size x size
      for (int l = 0; l < i; ++l)
        for (int j = 0; j < nr; ++j) c[i * nr + j] -= lu[i * cols + l] * c[l * nr + j];
    if (rows > cols)
      for (int i = cols; i < rows; ++i)
        for (int l = 0; l < cols; ++l)
          for (int j = 0; j < nr; ++j) c[i * nr + j] -= lu[i * cols + l] * c[l * nr + j];
    for (int i = rank - 1; i >= 0; --i)  // upper solve rank x rank
      for (int j = 0; j < nr; ++j) {
        double t = c[i * nr + j];
        for (int l = i + 1; l < rank; ++l) t -= lu[i * cols + l] * c[l * nr + j];
        c[i * nr + j] = t / lu[i * cols + i];
      }
    for (int i = 0; i < rank; ++i) {
      for (int j = 0; j < nx; ++j) pr.Px[qidx[i] * nx + j] = -c[i * nr + j];
      pr.Pe[qidx[i]] = -c[i * nr + nx];
    }
  }
  // kernel: Q [-U11^{-1} U12; I]   (pivots above threshold are the leading `rank` ones under complete pivoting)
  const int dimker = cols - rank;
  std::vector<double> X(std::max(1, rank * dimker), 0.0);
  for (int i = rank - 1; i >= 0; --i)
    for (int j = 0; j < dimker; ++j) {
      double t = lu[i * cols + rank + j];
      for (int l = i + 1; l < rank; ++l) t -= lu[i * cols + l] * X[l * dimker + j];
      X[i * dimker + j] = t / lu[i * cols + i];
    }
  for (int i = 0; i < rank; ++i)
    for (int j = 0; j < dimker; ++j) pr.Pu[qidx[i] * nu + j] = -X[i * dimker + j];
  for (int j = 0; j < dimker; ++j) pr.Pu[qidx[rank + j] * nu + j] = 1.0;
}

// [OCS2-upstream] changeOfInputVariables (dynamics and cost), u = Px x + Pu utilde + Pe
struct ProjectedLQ {
  int nut = 0;
  std::vector<double> A, B, b, Q, R, P, q, r;  // B nx*nut, R nut*nut, P nut*nx, r nut (dense, stride nut)
  double c = 0;
};

void project_lq(int nx, int nu, const NodeLQ& lq, const Projection& pr, ProjectedLQ& o) {
  const int nt = pr.nut;
  o.nut = nt;
  o.A = lq.A; o.b = lq.b; o.Q = lq.Q; o.q = lq.q; o.c = lq.c;
  o.B.assign(nx * std::max(nt, 1), 0.0); o.R.assign(std::max(nt * nt, 1), 0.0); o.P.assign(std::max(nt, 1) * nx, 0.0); o.r.assign(std::max(nt, 1), 0.0);
  // dynamics
  for (int i = 0; i < nx; ++i) {
    for (int j = 0; j < nx; ++j) { double t = 0; for (int l = 0; l < nu; ++l) t += lq.B[i * nu + l] * pr.Px[l * nx + j]; o.A[i * nx + j] += t; }
    for (int j = 0; j < nt; ++j) { double t = 0; for (int l = 0; l < nu; ++l) t += lq.B[i * nu + l] * pr.Pu[l * nu + j]; o.B[i * nt + j] = t; }
    double t = 0; for (int l = 0; l < nu; ++l) t += lq.B[i * nu + l] * pr.Pe[l]; o.b[i] += t;
  }
  // cost
  std::vector<double> Ru0(nu), rr(nu), RPx(nu * nx), PRP(nu * nx);
  for (int i = 0; i < nu; ++i) { double t = 0; for (int l = 0; l < nu; ++l) t += lq.R[i * nu + l] * pr.Pe[l]; Ru0[i] = t; }
  for (int i = 0; i < nu; ++i) o.c += pr.Pe[i] * (lq.r[i] + 0.5 * Ru0[i]);
  for (int j = 0; j < nx; ++j) { double t = 0; for (int l = 0; l < nu; ++l) t += lq.P[l * nx + j] * pr.Pe[l]; o.q[j] += t; }
  for (int i = 0; i < nu; ++i) rr[i] = lq.r[i] + Ru0[i];
  for (int j = 0; j < nx; ++j) { double t = 0; for (int l = 0; l < nu; ++l) t += pr.Px[l * nx + j] * rr[l]; o.q[j] += t; }
  for (int i = 0; i < nu; ++i)
    for (int j = 0; j < nx; ++j) { double t = 0; for (int l = 0; l < nu; ++l) t += lq.R[i * nu + l] * pr.Px[l * nx + j]; RPx[i * nx + j] = t; }
  for (int i = 0; i < nu; ++i)
    for (int j = 0; j < nx; ++j) PRP[i * nx + j] = lq.P[i * nx + j] + RPx[i * nx + j];  // P + R Px
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < nx; ++j) {
      double t = 0;
      for (int l = 0; l < nu; ++l) t += pr.Px[l * nx + i] * lq.P[l * nx + j] + lq.P[l * nx + i] * pr.Px[l * nx + j] + pr.Px[l * nx + i] * RPx[l * nx + j];
      o.Q[i * nx + j] += t;
    }
  for (int i = 0; i < nt; ++i) {
    double t = 0; for (int l = 0; l < nu; ++l) t += pr.Pu[l * nu + i] * rr[l]; o.r[i] = t;
    for (int j = 0; j < nx; ++j) { double s = 0; for (int l = 0; l < nu; ++l) s += pr.Pu[l * nu + i] * PRP[l * nx + j]; o.P[i * nx + j] = s; }
  }
  std::vector<double> RPu(nu * std::max(nt, 1));
  for (int i = 0; i < nu; ++i)
    for (int j = 0; j < nt; ++j) { double t = 0; for (int l = 0; l < nu; ++l) t += lq.R[i * nu + l] * pr.Pu[l * nu + j]; RPu[i * nt + j] = t; }
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < nt; ++j) { double t = 0; for (int l = 0; l < nu; ++l) t += pr.Pu[l * nu + i] * RPu[l * nt + j]; o.R[i * nt + j] = t; }
}

// ------------------------------------------------------------------------------------------------
// Riccati recursion for the projected equality-free QP.  [OCS2-upstream] the reference hands this QP to HPIPM
// (hpipm_catkin), which for a QP without inequality rows performs one Riccati factorise+solve; the minimiser and
// the feedback gains are unique, so any exact method gives the same result up to round-off.
// Terminal cost: none registered (only src/BipedalRobotInterface.cpp:151 adds a cost) -> S_N = 0, s_N = 0.
// ------------------------------------------------------------------------------------------------
struct QPSolution {
  std::vector<double> dx, dut;         // (N+1)*nx, N*nu (utilde, first nut entries)
  std::vector<double> Kt, kt;          // N*nu*nx (rows nut), N*nu
};

bool cholesky(int n, std::vector<double>& H) {  // in place lower
  for (int j = 0; j < n; ++j) {
    double d = H[j * n + j];
    for (int l = 0; l < j; ++l) d -= H[j * n + l] * H[j * n + l];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    H[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double t = H[i * n + j];
      for (int l = 0; l < j; ++l) t -= H[i * n + l] * H[j * n + l];
      H[i * n + j] = t / d;
    }
  }
  return true;
}
void chol_solve(int n, const std::vector<double>& L, double* rhs, int nrhs, int stride) {  // rhs is n x nrhs
  for (int c = 0; c < nrhs; ++c) {
    for (int i = 0; i < n; ++i) { double t = rhs[i * stride + c]; for (int l = 0; l < i; ++l) t -= L[i * n + l] * rhs[l * stride + c]; rhs[i * stride + c] = t / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double t = rhs[i * stride + c]; for (int l = i + 1; l < n; ++l) t -= L[l * n + i] * rhs[l * stride + c]; rhs[i * stride + c] = t / L[i * n + i]; }
  }
}

// reg: HPIPM's reg_prim ([OCS2-upstream] hpipm_catkin: 1e-12), added to the diagonal of every stage Hessian [R~ P~; P~' Q~] (and of the
// terminal one, which is otherwise zero here) before the factorisation.  0 = exact recursion (the default of this restatement).
int riccati(int nx, int nu, int N, const std::vector<ProjectedLQ>& lq, const double* dx0, QPSolution& sol, double reg = 0.0) {
  sol.dx.assign((N + 1) * nx, 0.0); sol.dut.assign(N * nu, 0.0); sol.Kt.assign(N * nu * nx, 0.0); sol.kt.assign(N * nu, 0.0);
  std::vector<double> S(nx * nx, 0.0), s(nx, 0.0), SA(nx * nx), Sb(nx), Sn(nx * nx), sn(nx);
  for (int i = 0; i < nx; ++i) S[i * nx + i] = reg;
  for (int k = N - 1; k >= 0; --k) {
    const ProjectedLQ& n = lq[k];
    const int nt = n.nut;
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nx; ++j) { double t = 0; for (int l = 0; l < nx; ++l) t += S[i * nx + l] * n.A[l * nx + j]; SA[i * nx + j] = t; }
      double t = s[i]; for (int l = 0; l < nx; ++l) t += S[i * nx + l] * n.b[l]; Sb[i] = t;
    }
    std::vector<double> H(std::max(1, nt * nt)), G(std::max(1, nt) * nx), g(std::max(1, nt));
    for (int i = 0; i < nt; ++i) {
      for (int j = 0; j < nx; ++j) { double t = n.P[i * nx + j]; for (int l = 0; l < nx; ++l) t += n.B[l * nt + i] * SA[l * nx + j]; G[i * nx + j] = t; }
      double t = n.r[i]; for (int l = 0; l < nx; ++l) t += n.B[l * nt + i] * Sb[l]; g[i] = t;
    }
    std::vector<double> SB(nx * std::max(1, nt));
    for (int i = 0; i < nx; ++i)
      for (int j = 0; j < nt; ++j) { double t = 0; for (int l = 0; l < nx; ++l) t += S[i * nx + l] * n.B[l * nt + j]; SB[i * nt + j] = t; }
    for (int i = 0; i < nt; ++i)
      for (int j = 0; j < nt; ++j) { double t = n.R[i * nt + j] + (i == j ? reg : 0.0); for (int l = 0; l < nx; ++l) t += n.B[l * nt + i] * SB[l * nt + j]; H[i * nt + j] = t; }
    double* Kt = &sol.Kt[k * nu * nx];
    double* kt = &sol.kt[k * nu];
    if (nt > 0) {
      if (!cholesky(nt, H)) return -1;
      std::vector<double> rhs(nt * (nx + 1));
      for (int i = 0; i < nt; ++i) { for (int j = 0; j < nx; ++j) rhs[i * (nx + 1) + j] = -G[i * nx + j]; rhs[i * (nx + 1) + nx] = -g[i]; }
      chol_solve(nt, H, rhs.data(), nx + 1, nx + 1);
      for (int i = 0; i < nt; ++i) { for (int j = 0; j < nx; ++j) Kt[i * nx + j] = rhs[i * (nx + 1) + j]; kt[i] = rhs[i * (nx + 1) + nx]; }
    }
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nx; ++j) {
        double t = n.Q[i * nx + j] + (i == j ? reg : 0.0);
        for (int l = 0; l < nx; ++l) t += n.A[l * nx + i] * SA[l * nx + j];
        for (int l = 0; l < nt; ++l) t += G[l * nx + i] * Kt[l * nx + j];
        Sn[i * nx + j] = t;
      }
      double t = n.q[i];
      for (int l = 0; l < nx; ++l) t += n.A[l * nx + i] * Sb[l];
      for (int l = 0; l < nt; ++l) t += G[l * nx + i] * kt[l];
      sn[i] = t;
    }
    for (int i = 0; i < nx; ++i)
      for (int j = 0; j < nx; ++j) S[i * nx + j] = 0.5 * (Sn[i * nx + j] + Sn[j * nx + i]);
    s = sn;
  }
  for (int i = 0; i < nx; ++i) sol.dx[i] = dx0[i];
  for (int k = 0; k < N; ++k) {
    const ProjectedLQ& n = lq[k];
    const int nt = n.nut;
    const double* x = &sol.dx[k * nx];
    double* ut = &sol.dut[k * nu];
    for (int i = 0; i < nt; ++i) { double t = sol.kt[k * nu + i]; for (int j = 0; j < nx; ++j) t += sol.Kt[k * nu * nx + i * nx + j] * x[j]; ut[i] = t; }
    double* xn = &sol.dx[(k + 1) * nx];
    for (int i = 0; i < nx; ++i) {
      double t = n.b[i];
      for (int j = 0; j < nx; ++j) t += n.A[i * nx + j] * x[j];
      for (int j = 0; j < nt; ++j) t += n.B[i * nt + j] * ut[j];
      xn[i] = t;
    }
  }
  return 0;
}

struct Problem {
  int N;
  const int* kind; const double* dt; const int* mode; const double* zref; const double* zdref; const double* xref;
};

struct StepOut {
  std::vector<double> dx, du, K;
  double armijo = 0;
  double base[3] = {0, 0, 0};
};

// one QP: [OCS2-upstream] SqpSolver::setupQuadraticSubproblem + getOCPSolution
int qp_step(const Model& m, const Problem& pb, const double* x0, const double* x, const double* u, StepOut& out, double reg_prim = 0.0) {
  const int nx = m.nx, nu = m.nu, N = pb.N;
  std::vector<ProjectedLQ> plq(N);
  std::vector<Projection> proj(N);
  NodeLQ lq;
  out.base[0] = out.base[1] = out.base[2] = 0;
  for (int k = 0; k < N; ++k) {
    node_lq(m, pb.kind[k], pb.dt[k], x + k * nx, u + k * nu, x + (k + 1) * nx, pb.xref + k * nx, pb.mode[k], pb.zref + 4 * k, pb.zdref + 4 * k, lq);
    for (int i = 0; i < 3; ++i) out.base[i] += lq.perf[i];
    if (pb.kind[k] == 1) {
      proj[k].rank = nu; proj[k].nut = 0;
      proj[k].Px.assign(nu * nx, 0.0); proj[k].Pu.assign(nu * nu, 0.0); proj[k].Pe.assign(nu, 0.0);
      plq[k].nut = 0; plq[k].A = lq.A; plq[k].b = lq.b; plq[k].Q = lq.Q; plq[k].q = lq.q; plq[k].c = 0;
      plq[k].B.assign(nx, 0.0); plq[k].R.assign(1, 0.0); plq[k].P.assign(nx, 0.0); plq[k].r.assign(1, 0.0);
    } else {
      lu_projection(lq.nc, nx, nu, lq.C.data(), lq.D.data(), lq.e.data(), proj[k]);
      project_lq(nx, nu, lq, proj[k], plq[k]);
    }
  }
  double dx0[MAXX], d0 = 0;
  for (int i = 0; i < nx; ++i) { dx0[i] = x0[i] - x[i]; d0 += dx0[i] * dx0[i]; }
  out.base[1] += d0;  // account for the initial state in the performance
  QPSolution sol;
  if (riccati(nx, nu, N, plq, dx0, sol, reg_prim) != 0) return -1;
  // armijo descent metric with the projected cost and utilde ([OCS2-upstream] SqpSolver::getOCPSolution order)
  double metric = 0;
  for (int k = 0; k < N; ++k) {
    for (int i = 0; i < nx; ++i) metric += plq[k].q[i] * sol.dx[k * nx + i];
    for (int i = 0; i < plq[k].nut; ++i) metric += plq[k].r[i] * sol.dut[k * nu + i];
  }
  out.armijo = metric;
  out.dx = sol.dx;
  out.du.assign(N * nu, 0.0);
  out.K.assign(N * nu * nx, 0.0);
  for (int k = 0; k < N; ++k) {
    if (pb.kind[k] == 1) continue;
    const Projection& pr = proj[k];
    for (int i = 0; i < nu; ++i) {
      double t = pr.Pe[i];
      for (int j = 0; j < nx; ++j) t += pr.Px[i * nx + j] * sol.dx[k * nx + j];
      for (int j = 0; j < pr.nut; ++j) t += pr.Pu[i * nu + j] * sol.dut[k * nu + j];
      out.du[k * nu + i] = t;
      for (int j = 0; j < nx; ++j) {
        double g = pr.Px[i * nx + j];
        for (int l = 0; l < pr.nut; ++l) g += pr.Pu[i * nu + l] * sol.Kt[k * nu * nx + l * nx + j];
        out.K[k * nu * nx + i * nx + j] = g;
      }
    }
  }
  return 0;
}

void performance(const Model& m, const Problem& pb, const double* x0, const double* x, const double* u, double* perf) {
  const int nx = m.nx, nu = m.nu;
  perf[0] = perf[1] = perf[2] = 0;
  for (int k = 0; k < pb.N; ++k) {
    double p[3];
    node_perf(m, pb.kind[k], pb.dt[k], x + k * nx, u + k * nu, x + (k + 1) * nx, pb.xref + k * nx, pb.mode[k], pb.zref + 4 * k, pb.zdref + 4 * k, p);
    for (int i = 0; i < 3; ++i) perf[i] += p[i];
  }
  double d0 = 0;
  for (int i = 0; i < nx; ++i) { const double d = x0[i] - x[i]; d0 += d * d; }
  perf[1] += d0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C API
// ------------------------------------------------------------------------------------------------
struct oracle_model { Model m; };

extern "C" {

oracle_model* oracle_model_create(const double* blob, int n) {
  if (n < 1) return nullptr;
  const int nj = static_cast<int>(blob[0]);
  if (nj < 1 || nj > MAXJ) return nullptr;
  const int nx = 12 + nj;
  const int expect = 1 + nj + 9 * nj + 3 * nj + 3 * nj + (nj + 1) * 13 + 4 + 12 + 2 * nx * nx + 8;
  if (n != expect && n != expect + 3) return nullptr;      // optional trailer: hard friction cone flag, sqp.inequalityConstraintMu, Delta
  oracle_model* om = new oracle_model;
  Model& m = om->m;
  m.nj = nj; m.nx = nx; m.nu = nx;
  const double* p = blob + 1;
  m.parent[0] = -1;
  for (int j = 1; j <= nj; ++j) m.parent[j] = static_cast<int>(*p++);
  for (int j = 1; j <= nj; ++j) for (int i = 0; i < 9; ++i) m.Rfix[j][i] = *p++;
  for (int j = 1; j <= nj; ++j) for (int i = 0; i < 3; ++i) m.pfix[j][i] = *p++;
  for (int j = 1; j <= nj; ++j) for (int i = 0; i < 3; ++i) m.axis[j][i] = *p++;
  for (int b = 0; b <= nj; ++b) m.mass[b] = *p++;
  for (int b = 0; b <= nj; ++b) for (int i = 0; i < 3; ++i) m.com[b][i] = *p++;
  for (int b = 0; b <= nj; ++b) for (int i = 0; i < 9; ++i) m.inertia[b][i] = *p++;
  for (int c = 0; c < 4; ++c) m.cbody[c] = static_cast<int>(*p++);
  for (int c = 0; c < 4; ++c) for (int i = 0; i < 3; ++i) m.coff[c][i] = *p++;
  m.Q.assign(p, p + nx * nx); p += nx * nx;
  m.R.assign(p, p + nx * nx); p += nx * nx;
  m.mu = *p++; m.reg = *p++; m.grip = *p++; m.shift = *p++; m.bmu = *p++; m.bdelta = *p++; m.gain = *p++; m.robot_mass = *p++;
  if (n == expect + 3) { m.hard = *p++ != 0.0 ? 1 : 0; m.imu = *p++; m.idelta = *p++; }
  for (int j = 0; j <= nj; ++j)
    for (int b = 0; b <= nj; ++b) {
      bool a = false;
      int c = b;
      while (c > 0) { if (c == j) { a = true; break; } c = m.parent[c]; }
      m.anc[j][b] = a;
    }
  return om;
}
void oracle_model_destroy(oracle_model* m) { delete m; }
int oracle_model_nx(const oracle_model* m) { return m->m.nx; }

int oracle_flow_map(const oracle_model* om, const double* x, const double* u, double* f, double* A, double* B) {
  const Model& m = om->m;
  if (A && B) {
    flow_map_lin(m, x, u, f, A, B);
  } else {
    flow_map<double>(m, x, u, f);
  }
  return 0;
}

int oracle_ee_kinematics(const oracle_model* om, const double* x, const double* u, double* pos, double* vel, double* dpdx, double* dvdx,
                         double* dvdu) {
  const Model& m = om->m;
  if (dpdx || dvdx || dvdu) {
    ee_lin(m, x, u, pos, vel, dpdx, dvdx, dvdu);
  } else {
    double p[4][3], v[4][3];
    ee_kinematics<double>(m, x, u, p, v);
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 3; ++i) { pos[3 * c + i] = p[c][i]; vel[3 * c + i] = v[c][i]; }
  }
  return 0;
}

int oracle_cmm(const oracle_model* om, const double* q, double* A, double* com) {
  const Model& m = om->m;
  Kin<double> k;
  kinematics<double>(m, q, k);
  cmm<double>(m, k);
  const int G = 6 + m.nj;
  for (int i = 0; i < 6; ++i) for (int g = 0; g < G; ++g) A[i * G + g] = k.A[i][g];
  for (int i = 0; i < 3; ++i) com[i] = k.com[i];
  return 0;
}

int oracle_node_lq(const oracle_model* om, int kind, double dt, const double* x, const double* u, const double* xnext, const double* xref,
                   int mode, const double* zref4, const double* zdref4, double* A, double* B, double* b, double* Q, double* R, double* P,
                   double* q, double* r, double* c, double* C, double* D, double* e, int* nc, double* perf) {
  const Model& m = om->m;
  NodeLQ o;
  node_lq(m, kind, dt, x, u, xnext, xref, mode, zref4, zdref4, o);
  auto cp = [](const std::vector<double>& s, double* d) { if (d) std::memcpy(d, s.data(), s.size() * sizeof(double)); };
  cp(o.A, A); cp(o.B, B); cp(o.b, b); cp(o.Q, Q); cp(o.R, R); cp(o.P, P); cp(o.q, q); cp(o.r, r); cp(o.C, C); cp(o.D, D); cp(o.e, e);
  if (c) *c = o.c;
  if (nc) *nc = o.nc;
  if (perf) for (int i = 0; i < 3; ++i) perf[i] = o.perf[i];
  return 0;
}

// Operation counts of this restatement at (x, u) in mode `mode` (liboracle_count.so only; -1 otherwise):
//   out[0] flow map, value only          out[1] end-effector kinematics (4 contacts), value only
//   out[2] flow map with all nx + nu forward-mode directions            out[3] end-effector kinematics with all directions
//   out[4] one complete node linearisation (two flow-map Jacobians, RK2 combination, cost, soft cones, end-effector rows)
int oracle_flop_counts(const oracle_model* om, const double* x, const double* u, int mode, double* out) {
#ifdef ORACLE_COUNT_FLOPS
  const Model& m = om->m;
  CD xc[MAXX], uc[MAXX], fc[MAXX], pc[4][3], vc[4][3];
  for (int i = 0; i < m.nx; ++i) { xc[i] = CD(x[i]); uc[i] = CD(u[i]); }
  g_flops = 0; flow_map(m, xc, uc, fc); out[0] = static_cast<double>(g_flops);
  g_flops = 0; ee_kinematics(m, xc, uc, pc, vc); out[1] = static_cast<double>(g_flops);
  std::vector<double> f(m.nx), A(m.nx * m.nx), B(m.nx * m.nu), pos(12), vel(12), dpdx(12 * m.nx), dvdx(12 * m.nx), dvdu(12 * m.nu);
  g_flops = 0; flow_map_lin(m, x, u, f.data(), A.data(), B.data()); out[2] = static_cast<double>(g_flops);
  g_flops = 0; ee_lin(m, x, u, pos.data(), vel.data(), dpdx.data(), dvdx.data(), dvdu.data()); out[3] = static_cast<double>(g_flops);
  NodeLQ lq;
  const double z[4] = {0, 0, 0, 0};
  g_flops = 0; node_lq(m, 0, 0.015, x, u, x, x, mode, z, z, lq); out[4] = static_cast<double>(g_flops);
  return 0;
#else
  (void)om; (void)x; (void)u; (void)mode; (void)out;
  return -1;
#endif
}

int oracle_node_perf(const oracle_model* om, int kind, double dt, const double* x, const double* u, const double* xnext, const double* xref,
                     int mode, const double* zref4, const double* zdref4, double* perf) {
  node_perf(om->m, kind, dt, x, u, xnext, xref, mode, zref4, zdref4, perf);
  return 0;
}

int oracle_lu_projection(int nc, int nx, int nu, const double* C, const double* D, const double* e, double* Px, double* Pu, double* Pe,
                         int* rank) {
  Projection pr;
  lu_projection(nc, nx, nu, C, D, e, pr);
  std::memcpy(Px, pr.Px.data(), sizeof(double) * nu * nx);
  std::memcpy(Pu, pr.Pu.data(), sizeof(double) * nu * nu);
  std::memcpy(Pe, pr.Pe.data(), sizeof(double) * nu);
  *rank = pr.rank;
  return 0;
}

int oracle_qp_step(const oracle_model* om, int N, const int* kind, const double* dt, const int* mode, const double* zref, const double* zdref,
                   const double* xref, const double* x0, const double* x, const double* u, double* dx, double* du, double* K) {
  return oracle_qp_step_reg(om, N, kind, dt, mode, zref, zdref, xref, x0, x, u, 0.0, dx, du, K);
}
int oracle_qp_step_reg(const oracle_model* om, int N, const int* kind, const double* dt, const int* mode, const double* zref, const double* zdref,
                       const double* xref, const double* x0, const double* x, const double* u, double reg_prim, double* dx, double* du, double* K) {
  const Model& m = om->m;
  Problem pb{N, kind, dt, mode, zref, zdref, xref};
  StepOut so;
  if (qp_step(m, pb, x0, x, u, so, reg_prim) != 0) return -1;
  std::memcpy(dx, so.dx.data(), sizeof(double) * so.dx.size());
  std::memcpy(du, so.du.data(), sizeof(double) * so.du.size());
  if (K) std::memcpy(K, so.K.data(), sizeof(double) * so.K.size());
  return 0;
}

// [OCS2-upstream] SqpSolver::runImpl / takeStep / FilterLinesearch::acceptStep / checkConvergence
int oracle_solve(const oracle_model* om, int N, const int* kind, const double* dt, const int* mode, const double* zref, const double* zdref,
                 const double* xref, const double* x0, const double* x_init, const double* u_init, const double* opts, double* x_out,
                 double* u_out, double* K_out, double* stats) {
  const Model& m = om->m;
  const int nx = m.nx, nu = m.nu;
  const int iters = static_cast<int>(opts[0]);
  const double g_max = opts[1], g_min = opts[2], alpha_decay = opts[3], alpha_min = opts[4], gamma_c = opts[5], armijo = opts[6],
               delta_tol = opts[7], reg_prim = opts[8];
  const double cost_tol = 1e-4;  // [OCS2-upstream] sqp::Settings default costTol
  Problem pb{N, kind, dt, mode, zref, zdref, xref};
  std::vector<double> x(x_init, x_init + (N + 1) * nx), u(u_init, u_init + N * nu), xn((N + 1) * nx), un(N * nu);
  StepOut so;
  for (int it = 0; it < iters; ++it) {
    double* st = stats + 16 * it;
    for (int i = 0; i < 16; ++i) st[i] = 0;
    if (qp_step(m, pb, x0, x.data(), u.data(), so, reg_prim) != 0) return -1;
    const double merit0 = so.base[0];
    const double viol0 = std::sqrt(so.base[1] + so.base[2]);
    st[0] = merit0; st[1] = so.base[1]; st[2] = so.base[2]; st[7] = so.armijo;
    double dxn = 0, dun = 0;
    for (double v : so.dx) dxn += v * v;
    for (double v : so.du) dun += v * v;
    dxn = std::sqrt(dxn); dun = std::sqrt(dun);
    double alpha = 1.0;
    bool accepted = false;
    double perf[3] = {0, 0, 0};
    int trials = 0;
    do {
      for (size_t i = 0; i < x.size(); ++i) xn[i] = x[i] + alpha * so.dx[i];
      for (size_t i = 0; i < u.size(); ++i) un[i] = u[i] + alpha * so.du[i];
      performance(m, pb, x0, xn.data(), un.data(), perf);
      ++trials;
      const double viol = std::sqrt(perf[1] + perf[2]);
      const double descent = alpha * so.armijo;
      if (viol > g_max) {
        accepted = viol < (1.0 - gamma_c) * viol0;
      } else if (viol < g_min && viol0 < g_min && descent < 0.0) {
        accepted = perf[0] < merit0 + armijo * descent;
      } else {
        accepted = perf[0] < (merit0 - gamma_c * viol0) || viol < (1.0 - gamma_c) * viol0;
      }
      if (accepted) break;
      alpha *= alpha_decay;
      // [OCS2-upstream] SqpSolver::takeStep: "detect too small step size during back-tracking to escape early" - once the
      // next trial step would be below deltaTol in both norms the search stops and no step is taken
      if (alpha * dun < delta_tol && alpha * dxn < delta_tol) break;
    } while (alpha >= alpha_min);
    st[10] = trials;
    if (K_out) std::memcpy(K_out, so.K.data(), sizeof(double) * so.K.size());
    if (accepted) {
      x = xn; u = un;
      st[3] = alpha; st[4] = perf[0]; st[5] = perf[1]; st[6] = perf[2]; st[8] = alpha * dxn; st[9] = alpha * dun;
    } else {
      st[3] = 0.0; st[4] = merit0; st[5] = so.base[1]; st[6] = so.base[2];
    }
    // convergence
    if (it + 1 >= iters) break;
    if (st[3] < alpha_min) break;
    if (std::fabs(st[4] - merit0) < cost_tol && std::sqrt(st[5] + st[6]) < g_min) break;
    if (st[8] < delta_tol && st[9] < delta_tol) break;
  }
  std::memcpy(x_out, x.data(), sizeof(double) * x.size());
  std::memcpy(u_out, u.data(), sizeof(double) * u.size());
  return 0;
}

}  // extern "C"
