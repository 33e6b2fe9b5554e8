// Centroidal dynamics of the floating-base biped and its analytic Jacobians, one wavefront per evaluation.
//
// Replaces, for one (x,u):
//   a1  BipedalRobotDynamicsAD::{computeFlowMap,linearApproximation}   ocs2_bipedal_robot/src/dynamics/BipedalRobotDynamicsAD.cpp:46-56
//       -> [OCS2-upstream] PinocchioCentroidalDynamicsAD (CppAD generated code in the reference)
//   a6  PinocchioEndEffectorKinematicsCppAd position / velocity models built at src/BipedalRobotInterface.cpp:169-178
// The reference differentiates with CppAD; here the Jacobians are hand-derived:
//   * the model is a tree of 1-DoF joints: 3 prismatic (world x,y,z), 3 revolute (Euler Z,Y,X at the base origin),
//     then the leg joints; generalised coordinate g: 0-2 translation, 3-5 yaw/pitch/roll, 6.. leg joints;
//   * centroidal momentum matrix by the composite-rigid-body recursion;
//   * d(A v)/dq_k (v fixed) = crf(s_k) h_sub(k) - Ic_k crm(s_k) V_k  (spatial-algebra identity, evaluated per column);
//   * d(J_i v)/dq_k (v fixed) = a_k x (v_i - v_ok) + (w_k x a_k) x (p_i - o_k) for joints k that move contact i.
// Lane roles: "lane b" = body b (0 = base, 1..NJ leg links), "lane g" = generalised coordinate g, and for the final
// assembly "lane c" = column c of [df/dx | df/du].
#pragma once
#include "../device_model.h"
#include "lane_model.h"

namespace bpmpc {

#if defined(BPMPC_EVAL_PROFILE) && !defined(BPMPC_HOST_EMULATION)
__device__ long long g_evprof[16];
#define EVPROF_BEGIN() long long ev_prev_ = clock64()
#define EVPROF(slot) do { const long long tn_ = clock64(); if (blockIdx.x == 40 && threadIdx.x == 0) g_evprof[slot] += tn_ - ev_prev_; ev_prev_ = tn_; } while (0)
#else
#define EVPROF_BEGIN() ((void)0)
#define EVPROF(slot) ((void)0)
#endif

template <int NJ>
struct CentroidalWorkspace {
  static constexpr int NB = NJ + 1, G = 6 + NJ, NX = 12 + NJ, NU = 12 + NJ;
  // per-body model constants cached once per kernel (cache_model): no phase touches global memory afterwards
  int path[NB][NJ], depth[NB], cbody[kNumContacts], maxdepth;
  unsigned subtree[NB], cpath[kNumContacts];
  double m_Rfix[NB][9], pfix[NB][3], m_axis[NB][3], m_com[NB][3], m_inertia[NB][6], m_mass[NB], m_coff[kNumContacts][3];
  // evaluation point
  double x[NX], u[NU];
  double eul[6];                        // sin/cos of yaw, pitch, roll: sy cy sp cp sr cr
  // LDS diet: buffers with disjoint lifetimes share storage.
  //   kinematic scratch (dead once phase J is done)  <->  the dense rows Ar/Br written by the last phase
  //   contact Jacobians (dead after the constraint rows are written)  <->  the saved k1 rows Ar1/Br1 of the RK2 scheme
  union {
    struct {
      double E[NB][9];                  // joint-local rotation Rfix * Rot(axis, q)
      double R[NB][9];                  // world rotation of the body frames
      double comp[NB][10];              // per body: mass, first moment about o0 (3), inertia about o0 (6)
      double Iw[NB][6];                 // world inertia (about own com) of each body
      double hb[NB][6];                 // per body momentum: linear, angular about o0
    };
    struct {
      double Ar[9][NX], Br[9][NU];      // rows 3..11 of df/dx and df/du (the other rows are structural constants)
    };
  };
  union {
    struct {
      double J[3 * kNumContacts][G];    // contact-point Jacobians
      double DJv[3 * kNumContacts][G];  // d(J_i v)/dq at fixed v
    };
    struct {
      double Ar1[9][NX], Br1[9][NU];    // k1 rows kept across the second evaluation
    };
  };
  double o[NB][3];                      // world position of the body frames
  double cw[NB][3];                     // world com of each body
  double Mc[NB], Cc[NB][3], Ic[NB][6];  // composite (subtree) mass, com, inertia about the composite com
  double ah[G][3], og[G][3];            // world axis and a point on the axis of generalised coordinate g
  double A[6][G];                       // centroidal momentum matrix
  double cpos[kNumContacts][3], cvel[kNumContacts][3];
  double rhs[6];
  double X12[9], X22[9];                // blocks of A_b^{-1}: [[I/m, X12],[0, X22]]
  double v[G];                          // generalised velocity [v_base; v_joints]
  double omg[G][3], vog[G][3];          // twist (angular velocity, velocity of og) of the body moved by coordinate g>=3
  double hs[NB][6];                     // subtree momentum
  double f[NX];
};

template <int NJ>
BP_DEVICE void cache_model(const DeviceModel& md, CentroidalWorkspace<NJ>& w) {
  constexpr int NB = NJ + 1;
  BP_LANES(tid, kWave) {
    for (int idx = tid; idx < NB * NJ; idx += kWave) w.path[idx / NJ][idx % NJ] = md.path[idx / NJ][idx % NJ];
    for (int idx = tid; idx < NB * 9; idx += kWave) w.m_Rfix[idx / 9][idx % 9] = md.Rfix[idx / 9][idx % 9];
    for (int idx = tid; idx < NB * 6; idx += kWave) w.m_inertia[idx / 6][idx % 6] = md.inertia[idx / 6][idx % 6];
    if (tid < NB) {
      w.depth[tid] = md.depth[tid];
      w.subtree[tid] = md.subtree[tid];
      w.m_mass[tid] = md.mass[tid];
      for (int i = 0; i < 3; ++i) { w.pfix[tid][i] = md.pfix[tid][i]; w.m_axis[tid][i] = md.axis[tid][i]; w.m_com[tid][i] = md.com[tid][i]; }
    } else if (tid >= 32 && tid < 32 + kNumContacts) {
      const int i = tid - 32;
      w.cbody[i] = md.contact_body[i];
      w.cpath[i] = md.contact_path[i];
      for (int a = 0; a < 3; ++a) w.m_coff[i][a] = md.contact_off[i][a];
    } else if (tid == 63) {
      int mx = 0;
      for (int b = 0; b < NB; ++b) mx = md.depth[b] > mx ? md.depth[b] : mx;
      w.maxdepth = mx;
    }
  }
  BP_SYNC();
}

// contact-point Jacobian column g of contact i (world aligned): e_g, or a_g x (p_i - o_g) when g moves the contact
template <int NJ>
BP_DEVICE void contact_jacobian_column(const CentroidalWorkspace<NJ>& w, int i, int g, double* col) {
  col[0] = col[1] = col[2] = 0.0;
  if (g < 3) {
    col[g] = 1.0;
  } else if (g < 6 || ((w.cpath[i] >> (g - 5)) & 1u)) {
    const double r[3] = {w.cpos[i][0] - w.og[g][0], w.cpos[i][1] - w.og[g][1], w.cpos[i][2] - w.og[g][2]};
    cross3(w.ah[g], r, col);
  }
}

// DERIV: also produce Ar/Br (and DJv when WITH_EE).  WITH_EE: contact positions / velocities (and J).
// cache_model() must have run on this workspace.
template <int NJ, bool DERIV, bool WITH_EE>
BP_DEVICE void eval_centroidal(const DeviceModel& md, CentroidalWorkspace<NJ>& w) {
  constexpr int NB = NJ + 1, G = 6 + NJ, NX = 12 + NJ, NU = 12 + NJ;
  const double mass_total = md.robot_mass;

  EVPROF_BEGIN();
  // ---- phase A: body lane b: sin/cos of its joint angle and joint-local rotation; lanes NB..NB+2: Euler angles
  BP_LANES(tid, kWave) {
    if (tid >= 1 && tid < NB) {
      const double* a = w.m_axis[tid];
      double s, c;
      sincos(w.x[11 + tid], &s, &c);
      const double v = 1.0 - c;
      const double rot[9] = {c + v * a[0] * a[0],        v * a[0] * a[1] - s * a[2], v * a[0] * a[2] + s * a[1],
                             v * a[1] * a[0] + s * a[2], c + v * a[1] * a[1],        v * a[1] * a[2] - s * a[0],
                             v * a[2] * a[0] - s * a[1], v * a[2] * a[1] + s * a[0], c + v * a[2] * a[2]};
      mat3_mul(w.m_Rfix[tid], rot, w.E[tid]);
    } else if (tid >= NB && tid < NB + 3) {
      double s, c;
      sincos(w.x[9 + (tid - NB)], &s, &c);
      w.eul[2 * (tid - NB)] = s;
      w.eul[2 * (tid - NB) + 1] = c;
    }
  }
  BP_SYNC();
  EVPROF(0);
  // ---- phase C: every body lane walks its own chain base -> body (tables and joint transforms in LDS)
  BP_LANES(tid, kWave) {
    if (tid < NB) {
      const int b = tid;
      const double sy = w.eul[0], cy = w.eul[1], sp = w.eul[2], cp = w.eul[3], sr = w.eul[4], cr = w.eul[5];
      double R[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                     sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                     -sp,     cp * sr,                cp * cr};
      double o[3] = {w.x[6], w.x[7], w.x[8]};
      const double o0[3] = {o[0], o[1], o[2]};
      const int depth = w.depth[b], maxdepth = w.maxdepth;
      for (int d = 0; d < maxdepth; ++d) {          // uniform trip count; lanes past their own depth keep R, o
        const bool on = d < depth;
        const int j = on ? w.path[b][d] : 1;
        double Ej[9], pj[3], t[3], Rn[9];
        for (int i = 0; i < 9; ++i) Ej[i] = w.E[j][i];
        for (int i = 0; i < 3; ++i) pj[i] = w.pfix[j][i];
        mat3_vec(R, pj, t);
        mat3_mul(R, Ej, Rn);
        for (int i = 0; i < 3; ++i) o[i] = on ? o[i] + t[i] : o[i];
        for (int i = 0; i < 9; ++i) R[i] = on ? Rn[i] : R[i];
      }
      for (int i = 0; i < 9; ++i) w.R[b][i] = R[i];
      for (int i = 0; i < 3; ++i) w.o[b][i] = o[i];
      if (b == 0) {
        const double unit[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int g = 0; g < 3; ++g)
          for (int i = 0; i < 3; ++i) { w.ah[g][i] = unit[3 * g + i]; w.og[g][i] = o0[i]; }
        // Euler ZYX as three successive revolute joints: z, then rotated y, then rotated x
        w.ah[3][0] = 0.0;     w.ah[3][1] = 0.0;     w.ah[3][2] = 1.0;
        w.ah[4][0] = -sy;     w.ah[4][1] = cy;      w.ah[4][2] = 0.0;
        w.ah[5][0] = cy * cp; w.ah[5][1] = sy * cp; w.ah[5][2] = -sp;
        for (int g = 3; g < 6; ++g)
          for (int i = 0; i < 3; ++i) w.og[g][i] = o0[i];
      } else {
        mat3_vec(R, w.m_axis[b], w.ah[5 + b]);
        for (int i = 0; i < 3; ++i) w.og[5 + b][i] = o[i];
      }
      // body com / inertia in the world, and this body's contribution to the subtree sums (about o0)
      double c[3], d[3];
      mat3_vec(R, w.m_com[b], c);
      for (int i = 0; i < 3; ++i) { c[i] += o[i]; d[i] = c[i] - o0[i]; w.cw[b][i] = c[i]; }
      const double* I = w.m_inertia[b];
      const double Ib[9] = {I[0], I[1], I[2], I[1], I[3], I[4], I[2], I[4], I[5]};
      double T[9];
      mat3_mul(R, Ib, T);  // T = R I
      const double Ixx = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];
      const double Ixy = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
      const double Ixz = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
      const double Iyy = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];
      const double Iyz = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
      const double Izz = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
      w.Iw[b][0] = Ixx; w.Iw[b][1] = Ixy; w.Iw[b][2] = Ixz; w.Iw[b][3] = Iyy; w.Iw[b][4] = Iyz; w.Iw[b][5] = Izz;
      const double m = w.m_mass[b];
      const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      w.comp[b][0] = m;
      w.comp[b][1] = m * d[0]; w.comp[b][2] = m * d[1]; w.comp[b][3] = m * d[2];
      w.comp[b][4] = Ixx + m * (dd - d[0] * d[0]);
      w.comp[b][5] = Ixy - m * d[0] * d[1];
      w.comp[b][6] = Ixz - m * d[0] * d[2];
      w.comp[b][7] = Iyy + m * (dd - d[1] * d[1]);
      w.comp[b][8] = Iyz - m * d[1] * d[2];
      w.comp[b][9] = Izz + m * (dd - d[2] * d[2]);
    }
  }
  BP_SYNC();
  EVPROF(1);
  // ---- phase D: body lane b sums its subtree (members in increasing body order) and forms the composite
  //      mass / com / inertia about the composite com; lanes 16..19: contact positions
  BP_LANES(tid, kWave) {
    if (tid < NB) {
      const int b = tid;
      const unsigned members = w.subtree[b];
      double s[10];
      for (int c = 0; c < 10; ++c) s[c] = 0.0;
      for (int m = NB - 1; m >= 0; --m) {           // leaf -> root order; branch free so that all loads issue up front
        const double sel = ((members >> m) & 1u) ? 1.0 : 0.0;
        for (int c = 0; c < 10; ++c) s[c] += sel * w.comp[m][c];
      }
      const double M = s[0];
      const double inv = M > 0.0 ? 1.0 / M : 0.0;
      const double D[3] = {s[1] * inv, s[2] * inv, s[3] * inv};
      const double DD = D[0] * D[0] + D[1] * D[1] + D[2] * D[2];
      w.Mc[b] = M;
      for (int i = 0; i < 3; ++i) w.Cc[b][i] = w.o[0][i] + D[i];
      w.Ic[b][0] = s[4] - M * (DD - D[0] * D[0]);
      w.Ic[b][1] = s[5] + M * D[0] * D[1];
      w.Ic[b][2] = s[6] + M * D[0] * D[2];
      w.Ic[b][3] = s[7] - M * (DD - D[1] * D[1]);
      w.Ic[b][4] = s[8] + M * D[1] * D[2];
      w.Ic[b][5] = s[9] - M * (DD - D[2] * D[2]);
    } else if (tid >= 16 && tid < 16 + kNumContacts) {
      const int i = tid - 16, b = w.cbody[i];
      double t[3];
      mat3_vec(w.R[b], w.m_coff[i], t);
      for (int a = 0; a < 3; ++a) w.cpos[i][a] = w.o[b][a] + t[a];
    }
  }
  BP_SYNC();
  EVPROF(2);
  // ---- phase F: centroidal momentum matrix, one column per lane
  BP_LANES(tid, kWave) {
    if (tid < G) {
      const int g = tid;
      const double* com = w.Cc[0];
      double lin[3], ang[3];
      if (g < 3) {
        for (int i = 0; i < 3; ++i) { lin[i] = (i == g) ? w.Mc[0] : 0.0; ang[i] = 0.0; }
      } else {
        const int b = g < 6 ? 0 : g - 5;
        const double M = w.Mc[b];
        const double* C = w.Cc[b];
        const double rC[3] = {C[0] - w.og[g][0], C[1] - w.og[g][1], C[2] - w.og[g][2]};
        double vC[3], t[3], Iw[3];
        cross3(w.ah[g], rC, vC);  // velocity of the composite com for a unit rate
        const double dC[3] = {C[0] - com[0], C[1] - com[1], C[2] - com[2]};
        cross3(dC, vC, t);
        sym3_mul(w.Ic[b], w.ah[g], Iw);
        for (int i = 0; i < 3; ++i) { lin[i] = M * vC[i]; ang[i] = Iw[i] + M * t[i]; }
      }
      for (int i = 0; i < 3; ++i) { w.A[i][g] = lin[i]; w.A[3 + i][g] = ang[i]; }
    }
  }
  BP_SYNC();
  EVPROF(3);
  // ---- phase G: momentum right-hand side, inverse of the base block, contact Jacobians
  BP_LANES(tid, kWave) {
    if (tid < 6) {
      double acc = mass_total * w.x[tid];
      for (int j = 0; j < NJ; ++j) acc -= w.A[tid][6 + j] * w.u[12 + j];
      w.rhs[tid] = acc;
    } else if (tid == 8) {
      // A_b = [[m I, A12],[0, A22]];  A_b^{-1} = [[I/m, -(1/m) A12 A22^{-1}],[0, A22^{-1}]]
      // ([OCS2-upstream] computeFloatingBaseCentroidalMomentumMatrixInverse)
      double M[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[3 * i + j] = w.A[3 + i][3 + j];
      const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
      const double idet = 1.0 / (M[0] * c00 + M[1] * c01 + M[2] * c02);
      double Xi[9];
      Xi[0] = c00 * idet; Xi[1] = (M[2] * M[7] - M[1] * M[8]) * idet; Xi[2] = (M[1] * M[5] - M[2] * M[4]) * idet;
      Xi[3] = c01 * idet; Xi[4] = (M[0] * M[8] - M[2] * M[6]) * idet; Xi[5] = (M[2] * M[3] - M[0] * M[5]) * idet;
      Xi[6] = c02 * idet; Xi[7] = (M[1] * M[6] - M[0] * M[7]) * idet; Xi[8] = (M[0] * M[4] - M[1] * M[3]) * idet;
      const double im = 1.0 / w.A[0][0];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          w.X22[3 * i + j] = Xi[3 * i + j];
          w.X12[3 * i + j] = -im * (w.A[i][3] * Xi[j] + w.A[i][4] * Xi[3 + j] + w.A[i][5] * Xi[6 + j]);
        }
    }
    if (WITH_EE) {
      // one (contact, coordinate) item per lane pass: g = lane % 32 keeps the index arithmetic to shifts
      const int g = tid & 31, i0 = tid >> 5;
      if (g < G)
        for (int i = i0; i < kNumContacts; i += 2) {
          double col[3];
          contact_jacobian_column<NJ>(w, i, g, col);
          for (int a = 0; a < 3; ++a) w.J[3 * i + a][g] = col[a];
        }
    }
  }
  BP_SYNC();
  EVPROF(4);
  // ---- phase H: base velocity, flow map value, twists of the three Euler "virtual bodies"
  BP_LANES(tid, kWave) {
    const double im = 1.0 / w.A[0][0];
    double th[3], pd[3];
    mat3_vec(w.X22, &w.rhs[3], th);
    mat3_vec(w.X12, &w.rhs[3], pd);
    for (int i = 0; i < 3; ++i) pd[i] += im * w.rhs[i];
    if (tid < G) w.v[tid] = tid < 3 ? pd[tid] : (tid < 6 ? th[tid - 3] : w.u[12 + tid - 6]);
    if (tid < 3) {
      // [OCS2-upstream] getNormalizedCentroidalMomentumRate: (m g + sum F) / m
      double acc = (tid == 2) ? -9.81 * mass_total : 0.0;
      for (int i = 0; i < kNumContacts; ++i) acc += w.u[3 * i + tid];
      w.f[tid] = acc / mass_total;
    } else if (tid < 6) {
      const int a = tid - 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
      double acc = 0.0;
      for (int i = 0; i < kNumContacts; ++i) {
        const double r1 = w.cpos[i][a1] - w.Cc[0][a1], r2 = w.cpos[i][a2] - w.Cc[0][a2];
        acc += r1 * w.u[3 * i + a2] - r2 * w.u[3 * i + a1];
      }
      w.f[tid] = acc / mass_total;
    } else if (tid < 12) {
      w.f[tid] = tid < 9 ? pd[tid - 6] : th[tid - 9];
    } else if (tid < NX) {
      w.f[tid] = w.u[tid];
    }
    if (tid >= 3 && tid < 6) {
      double om[3] = {0.0, 0.0, 0.0};
      for (int k = 3; k <= tid; ++k)
        for (int i = 0; i < 3; ++i) om[i] += w.ah[k][i] * th[k - 3];
      for (int i = 0; i < 3; ++i) { w.omg[tid][i] = om[i]; w.vog[tid][i] = pd[i]; }
    }
  }
  BP_SYNC();
  EVPROF(5);
  if (!DERIV && !WITH_EE) return;
  // ---- phase I: body twists (chain walk) and body momenta about o0
  BP_LANES(tid, kWave) {
    if (tid < NB) {
      const int b = tid;
      double om[3] = {w.omg[5][0], w.omg[5][1], w.omg[5][2]};
      double vo[3] = {w.vog[5][0], w.vog[5][1], w.vog[5][2]};
      int prev = 0;
      const int depth = w.depth[b], maxdepth = w.maxdepth;
      for (int d = 0; d < maxdepth; ++d) {
        const bool on = d < depth;
        const int j = on ? w.path[b][d] : 1;
        const double r[3] = {w.o[j][0] - w.o[prev][0], w.o[j][1] - w.o[prev][1], w.o[j][2] - w.o[prev][2]};
        double t[3];
        cross3(om, r, t);
        const double qd = on ? w.v[5 + j] : 0.0;
        for (int i = 0; i < 3; ++i) { vo[i] = on ? vo[i] + t[i] : vo[i]; om[i] += w.ah[5 + j][i] * qd; }
        prev = on ? j : prev;
      }
      if (b > 0)
        for (int i = 0; i < 3; ++i) { w.omg[5 + b][i] = om[i]; w.vog[5 + b][i] = vo[i]; }
      if (DERIV) {
        const double rc[3] = {w.cw[b][0] - w.o[b][0], w.cw[b][1] - w.o[b][1], w.cw[b][2] - w.o[b][2]};
        double t[3], l[3], Iw[3], L[3];
        cross3(om, rc, t);
        const double m = w.m_mass[b];
        for (int i = 0; i < 3; ++i) l[i] = m * (vo[i] + t[i]);
        sym3_mul(w.Iw[b], om, Iw);
        const double d0[3] = {w.cw[b][0] - w.o[0][0], w.cw[b][1] - w.o[0][1], w.cw[b][2] - w.o[0][2]};
        cross3(d0, l, L);
        for (int i = 0; i < 3; ++i) { w.hb[b][i] = l[i]; w.hb[b][3 + i] = Iw[i] + L[i]; }
      }
    }
  }
  BP_SYNC();
  EVPROF(6);
  // ---- phase J: subtree momenta (body lane sums its subtree), contact velocities
  BP_LANES(tid, kWave) {
    if (DERIV && tid < NB) {
      const unsigned members = w.subtree[tid];
      double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      for (int m = NB - 1; m >= 0; --m) {
        const double sel = ((members >> m) & 1u) ? 1.0 : 0.0;
        for (int c = 0; c < 6; ++c) s[c] += sel * w.hb[m][c];
      }
      for (int c = 0; c < 6; ++c) w.hs[tid][c] = s[c];
    } else if (WITH_EE && tid >= 16 && tid < 16 + kNumContacts) {
      const int i = tid - 16, g = 5 + w.cbody[i];
      const double r[3] = {w.cpos[i][0] - w.og[g][0], w.cpos[i][1] - w.og[g][1], w.cpos[i][2] - w.og[g][2]};
      double t[3];
      cross3(w.omg[g], r, t);
      for (int a = 0; a < 3; ++a) w.cvel[i][a] = w.vog[g][a] + t[a];
    }
  }
  BP_SYNC();
  EVPROF(7);
  if (!DERIV) return;
  // ---- phase K: one lane per column of [df/dx | df/du] (rows 3..11), everything column-local:
  //      x columns 6+g: d(A v)/dq_g -> d v_base/dq_g = -A_b^{-1} d(A v)/dq_g and the angular-momentum-rate row;
  //      x columns 0..5: m A_b^{-1};  u force columns: [p_i - com]_x / m;  u joint columns: -A_b^{-1} A_j.
  //      Lanes NX+NU.. : d(J_i v)/dq when WITH_EE.
  BP_LANES(tid, kWave) {
    const double im = 1.0 / w.A[0][0];
    const double imt = 1.0 / mass_total;
    if (tid < NX) {
      const int c = tid;
      double col[9];
      if (c < 6) {
        col[0] = col[1] = col[2] = 0.0;
        for (int i = 0; i < 6; ++i) {
          double e;
          if (i < 3) e = c < 3 ? (c == i ? im : 0.0) : w.X12[3 * i + (c - 3)];
          else e = c < 3 ? 0.0 : w.X22[3 * (i - 3) + (c - 3)];
          col[3 + i] = mass_total * e;
        }
      } else {
        const int g = c - 6;
        double dl[3] = {0.0, 0.0, 0.0}, dL[3] = {0.0, 0.0, 0.0};
        if (g >= 3) {
          const int b = g < 6 ? 0 : g - 5;
          const double* a = w.ah[g];
          const double* ok = w.og[g];
          const double M = w.Mc[b];
          const double* l = &w.hs[b][0];
          // subtree angular momentum about the joint origin
          const double s[3] = {w.o[0][0] - ok[0], w.o[0][1] - ok[1], w.o[0][2] - ok[2]};
          double t[3], Lk[3];
          cross3(s, l, t);
          for (int i = 0; i < 3; ++i) Lk[i] = w.hs[b][3 + i] + t[i];
          // crm(s) V_k
          double wp[3], up[3], vC[3], rC[3], linp[3], angp[3], Iw[3];
          cross3(a, w.omg[g], wp);
          cross3(a, w.vog[g], up);
          for (int i = 0; i < 3; ++i) rC[i] = w.Cc[b][i] - ok[i];
          cross3(wp, rC, t);
          for (int i = 0; i < 3; ++i) vC[i] = up[i] + t[i];
          for (int i = 0; i < 3; ++i) linp[i] = M * vC[i];
          sym3_mul(w.Ic[b], wp, Iw);
          cross3(rC, vC, t);
          for (int i = 0; i < 3; ++i) angp[i] = Iw[i] + M * t[i];
          double al[3], aL[3], dLk[3];
          cross3(a, l, al);
          cross3(a, Lk, aL);
          for (int i = 0; i < 3; ++i) { dl[i] = al[i] - linp[i]; dLk[i] = aL[i] - angp[i]; }
          // move the reference point to the (moving) centre of mass
          const double sc[3] = {ok[0] - w.Cc[0][0], ok[1] - w.Cc[0][1], ok[2] - w.Cc[0][2]};
          const double imc = 1.0 / w.Mc[0];
          const double jc[3] = {w.A[0][g] * imc, w.A[1][g] * imc, w.A[2][g] * imc};
          double t2[3];
          cross3(sc, dl, t);
          cross3(jc, &w.hs[0][0], t2);
          for (int i = 0; i < 3; ++i) dL[i] = dLk[i] + t[i] - t2[i];
        }
        // d v_base / dq_g = -A_b^{-1} [dl; dL]
        for (int i = 0; i < 3; ++i) {
          col[3 + i] = -(im * dl[i] + w.X12[3 * i] * dL[0] + w.X12[3 * i + 1] * dL[1] + w.X12[3 * i + 2] * dL[2]);
          col[6 + i] = -(w.X22[3 * i] * dL[0] + w.X22[3 * i + 1] * dL[1] + w.X22[3 * i + 2] * dL[2]);
        }
        // d(angular momentum rate)/dq_g = (1/m) sum_i (J_i - J_com)[:,g] x F_i
        const double jcm[3] = {w.A[0][g] / w.Mc[0], w.A[1][g] / w.Mc[0], w.A[2][g] / w.Mc[0]};
        double acc[3] = {0.0, 0.0, 0.0};
        for (int i = 0; i < kNumContacts; ++i) {
          double jcol[3];
          contact_jacobian_column<NJ>(w, i, g, jcol);
          const double d[3] = {jcol[0] - jcm[0], jcol[1] - jcm[1], jcol[2] - jcm[2]};
          const double* F = &w.u[3 * i];
          acc[0] += d[1] * F[2] - d[2] * F[1];
          acc[1] += d[2] * F[0] - d[0] * F[2];
          acc[2] += d[0] * F[1] - d[1] * F[0];
        }
        for (int r = 0; r < 3; ++r) col[r] = acc[r] * imt;
      }
      for (int r = 0; r < 9; ++r) w.Ar[r][c] = col[r];
    } else if (tid < NX + NU) {
      const int c = tid - NX;
      double col[9];
      for (int r = 0; r < 9; ++r) col[r] = 0.0;
      if (c < 12) {  // [p_i - com]_x / m
        const int i = c / 3, k = c % 3;
        for (int r = 0; r < 3; ++r) {
          if (k != r) {
            const int other = 3 - r - k;
            const double d = (w.cpos[i][other] - w.Cc[0][other]) * imt;
            col[r] = (k == (r + 2) % 3) ? d : -d;   // (r x F)_r = r_{r+1} F_{r+2} - r_{r+2} F_{r+1}
          }
        }
      } else {       // -A_b^{-1} A_j
        const int g = 6 + (c - 12);
        for (int i = 0; i < 3; ++i) {
          col[3 + i] = -(im * w.A[i][g] + w.X12[3 * i] * w.A[3][g] + w.X12[3 * i + 1] * w.A[4][g] + w.X12[3 * i + 2] * w.A[5][g]);
          col[6 + i] = -(w.X22[3 * i] * w.A[3][g] + w.X22[3 * i + 1] * w.A[4][g] + w.X22[3 * i + 2] * w.A[5][g]);
        }
      }
      for (int r = 0; r < 9; ++r) w.Br[r][c] = col[r];
    }
    if (WITH_EE) {
      const int g = tid & 31, i0 = tid >> 5;
      if (g < G)
        for (int i = i0; i < kNumContacts; i += 2) {
          double col[3] = {0.0, 0.0, 0.0};
          if (g >= 3 && (g < 6 || ((w.cpath[i] >> (g - 5)) & 1u))) {
            const double* a = w.ah[g];
            const double dv[3] = {w.cvel[i][0] - w.vog[g][0], w.cvel[i][1] - w.vog[g][1], w.cvel[i][2] - w.vog[g][2]};
            const double r[3] = {w.cpos[i][0] - w.og[g][0], w.cpos[i][1] - w.og[g][1], w.cpos[i][2] - w.og[g][2]};
            double t1[3], wa[3], t2[3];
            cross3(a, dv, t1);
            cross3(w.omg[g], a, wa);
            cross3(wa, r, t2);
            for (int k = 0; k < 3; ++k) col[k] = t1[k] + t2[k];
          }
          for (int k = 0; k < 3; ++k) w.DJv[3 * i + k][g] = col[k];
        }
    }
  }
  BP_SYNC();
  EVPROF(8);
}

}  // namespace bpmpc
