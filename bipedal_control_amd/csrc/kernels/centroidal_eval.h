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
// Lane roles: "lane b" = body b (0 = base, 1..NJ leg links), "lane g" = generalised coordinate g.
#pragma once
#include "../device_model.h"
#include "lane_model.h"

namespace bpmpc {

template <int NJ>
struct CentroidalWorkspace {
  static constexpr int NB = NJ + 1, G = 6 + NJ, NX = 12 + NJ, NU = 12 + NJ;
  double x[NX], u[NU];
  double sn[G], cs[G];
  double E[NB][9];                      // joint-local rotation Rfix * Rot(axis, q)
  double R[NB][9], o[NB][3];            // world placement of the body frames
  double cw[NB][3], Iw[NB][6];          // world com and world inertia (about own com) of each body
  double comp[NB][10];                  // subtree sums: mass, first moment about o0 (3), inertia about o0 (6)
  double Mc[NB], Cc[NB][3], Ic[NB][6];  // composite mass, com, inertia about the composite com
  double ah[G][3], og[G][3];            // world axis and a point on the axis of generalised coordinate g
  double A[6][G];                       // centroidal momentum matrix
  double cpos[kNumContacts][3], cvel[kNumContacts][3];
  double J[3 * kNumContacts][G];        // contact-point Jacobians
  double rhs[6];
  double X12[9], X22[9];                // blocks of A_b^{-1}: [[I/m, X12],[0, X22]]
  double v[G];                          // generalised velocity [v_base; v_joints]
  double omg[G][3], vog[G][3];          // twist (angular velocity, velocity of og) of the body moved by coordinate g>=3
  double hs[NB][6];                     // subtree momentum: linear, angular about o0
  double Dh[6][G], dvb[6][G];           // d(A v)/dq and d v_base / dq
  double DJv[3 * kNumContacts][G];      // d(J_i v)/dq at fixed v
  double f[NX];
  double Ar[9][NX], Br[9][NU];          // rows 3..11 of df/dx and df/du (the other rows are structural constants)
};

// DERIV: also produce Ar/Br (and DJv when WITH_EE).  WITH_EE: contact positions / velocities (and J).
template <int NJ, bool DERIV, bool WITH_EE>
BP_DEVICE void eval_centroidal(const DeviceModel& md, CentroidalWorkspace<NJ>& w) {
  constexpr int NB = NJ + 1, G = 6 + NJ, NX = 12 + NJ, NU = 12 + NJ;
  const double mass_total = md.robot_mass;

  // ---- phase A: sin/cos of every angle (one lane per angle)
  BP_LANES(tid, kWave) {
    if (tid >= 3 && tid < G) {
      double s, c;
      sincos(w.x[6 + tid], &s, &c);
      w.sn[tid] = s;
      w.cs[tid] = c;
    }
  }
  BP_SYNC();
  // ---- phase B: joint-local rotations
  BP_LANES(tid, kWave) {
    if (tid >= 1 && tid < NB) {
      const double* a = md.axis[tid];
      const double s = w.sn[5 + tid], c = w.cs[5 + tid], v = 1.0 - c;
      const double rot[9] = {c + v * a[0] * a[0],        v * a[0] * a[1] - s * a[2], v * a[0] * a[2] + s * a[1],
                             v * a[1] * a[0] + s * a[2], c + v * a[1] * a[1],        v * a[1] * a[2] - s * a[0],
                             v * a[2] * a[0] - s * a[1], v * a[2] * a[1] + s * a[0], c + v * a[2] * a[2]};
      mat3_mul(md.Rfix[tid], rot, w.E[tid]);
    }
  }
  BP_SYNC();
  // ---- phase C: every body lane walks its own chain base -> body
  BP_LANES(tid, kWave) {
    if (tid < NB) {
      const int b = tid;
      const double sy = w.sn[3], cy = w.cs[3], sp = w.sn[4], cp = w.cs[4], sr = w.sn[5], cr = w.cs[5];
      double R[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                     sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                     -sp,     cp * sr,                cp * cr};
      double o[3] = {w.x[6], w.x[7], w.x[8]};
      const double o0[3] = {o[0], o[1], o[2]};
      const int depth = md.depth[b];
      for (int d = 0; d < NJ; ++d) {
        if (d < depth) {
          const int j = md.path[b][d];
          double t[3], Rn[9];
          mat3_vec(R, md.pfix[j], t);
          o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
          mat3_mul(R, w.E[j], Rn);
          for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        }
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
        mat3_vec(R, md.axis[b], w.ah[5 + b]);
        for (int i = 0; i < 3; ++i) w.og[5 + b][i] = o[i];
      }
      // body com / inertia in the world, and this body's contribution to the subtree sums (about o0)
      double c[3], d[3];
      mat3_vec(R, md.com[b], c);
      for (int i = 0; i < 3; ++i) { c[i] += o[i]; d[i] = c[i] - o0[i]; w.cw[b][i] = c[i]; }
      const double* I = md.inertia[b];
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
      const double m = md.mass[b];
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
  // ---- phase D: subtree sums leaf -> root (lane = component), contact positions (lanes 16..19)
  BP_LANES(tid, kWave) {
    if (tid < 10) {
      for (int b = NB - 1; b >= 1; --b) w.comp[md.parent[b]][tid] += w.comp[b][tid];
    } else if (tid >= 16 && tid < 16 + kNumContacts) {
      const int i = tid - 16, b = md.contact_body[i];
      double t[3];
      mat3_vec(w.R[b], md.contact_off[i], t);
      for (int a = 0; a < 3; ++a) w.cpos[i][a] = w.o[b][a] + t[a];
    }
  }
  BP_SYNC();
  // ---- phase E: composite mass / com / inertia about the composite com
  BP_LANES(tid, kWave) {
    if (tid < NB) {
      const int b = tid;
      const double M = w.comp[b][0];
      const double inv = M > 0.0 ? 1.0 / M : 0.0;
      const double D[3] = {w.comp[b][1] * inv, w.comp[b][2] * inv, w.comp[b][3] * inv};
      const double DD = D[0] * D[0] + D[1] * D[1] + D[2] * D[2];
      w.Mc[b] = M;
      for (int i = 0; i < 3; ++i) w.Cc[b][i] = w.o[0][i] + D[i];
      w.Ic[b][0] = w.comp[b][4] - M * (DD - D[0] * D[0]);
      w.Ic[b][1] = w.comp[b][5] + M * D[0] * D[1];
      w.Ic[b][2] = w.comp[b][6] + M * D[0] * D[2];
      w.Ic[b][3] = w.comp[b][7] - M * (DD - D[1] * D[1]);
      w.Ic[b][4] = w.comp[b][8] + M * D[1] * D[2];
      w.Ic[b][5] = w.comp[b][9] - M * (DD - D[2] * D[2]);
    }
  }
  BP_SYNC();
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
      for (int idx = tid; idx < kNumContacts * G; idx += kWave) {
        const int i = idx / G, g = idx % G;
        double col[3] = {0.0, 0.0, 0.0};
        if (g < 3) {
          col[g] = 1.0;
        } else if (g < 6 || ((md.contact_path[i] >> (g - 5)) & 1u)) {
          const double r[3] = {w.cpos[i][0] - w.og[g][0], w.cpos[i][1] - w.og[g][1], w.cpos[i][2] - w.og[g][2]};
          cross3(w.ah[g], r, col);
        }
        for (int a = 0; a < 3; ++a) w.J[3 * i + a][g] = col[a];
      }
    }
  }
  BP_SYNC();
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
  if (!DERIV && !WITH_EE) return;
  // ---- phase I: body twists (chain walk) and body momenta about o0
  BP_LANES(tid, kWave) {
    if (tid < NB) {
      const int b = tid;
      double om[3] = {w.omg[5][0], w.omg[5][1], w.omg[5][2]};
      double vo[3] = {w.vog[5][0], w.vog[5][1], w.vog[5][2]};
      int prev = 0;
      const int depth = md.depth[b];
      for (int d = 0; d < NJ; ++d) {
        if (d < depth) {
          const int j = md.path[b][d];
          const double r[3] = {w.o[j][0] - w.o[prev][0], w.o[j][1] - w.o[prev][1], w.o[j][2] - w.o[prev][2]};
          double t[3];
          cross3(om, r, t);
          const double qd = w.v[5 + j];
          for (int i = 0; i < 3; ++i) { vo[i] += t[i]; om[i] += w.ah[5 + j][i] * qd; }
          prev = j;
        }
      }
      if (b > 0)
        for (int i = 0; i < 3; ++i) { w.omg[5 + b][i] = om[i]; w.vog[5 + b][i] = vo[i]; }
      if (DERIV) {
        const double rc[3] = {w.cw[b][0] - w.o[b][0], w.cw[b][1] - w.o[b][1], w.cw[b][2] - w.o[b][2]};
        double t[3], l[3], Iw[3], L[3];
        cross3(om, rc, t);
        const double m = md.mass[b];
        for (int i = 0; i < 3; ++i) l[i] = m * (vo[i] + t[i]);
        sym3_mul(w.Iw[b], om, Iw);
        const double d0[3] = {w.cw[b][0] - w.o[0][0], w.cw[b][1] - w.o[0][1], w.cw[b][2] - w.o[0][2]};
        cross3(d0, l, L);
        for (int i = 0; i < 3; ++i) { w.hs[b][i] = l[i]; w.hs[b][3 + i] = Iw[i] + L[i]; }
      }
    }
  }
  BP_SYNC();
  // ---- phase J: subtree momenta, contact velocities
  BP_LANES(tid, kWave) {
    if (DERIV && tid < 6) {
      for (int b = NB - 1; b >= 1; --b) w.hs[md.parent[b]][tid] += w.hs[b][tid];
    } else if (WITH_EE && tid >= 16 && tid < 16 + kNumContacts) {
      const int i = tid - 16, g = 5 + md.contact_body[i];
      const double r[3] = {w.cpos[i][0] - w.og[g][0], w.cpos[i][1] - w.og[g][1], w.cpos[i][2] - w.og[g][2]};
      double t[3];
      cross3(w.omg[g], r, t);
      for (int a = 0; a < 3; ++a) w.cvel[i][a] = w.vog[g][a] + t[a];
    }
  }
  BP_SYNC();
  if (!DERIV) return;
  // ---- phase K: d(A v)/dq column per lane; d(J_i v)/dq
  BP_LANES(tid, kWave) {
    if (tid < G) {
      const int g = tid;
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
        double al[3], aL[3];
        cross3(a, l, al);
        cross3(a, Lk, aL);
        double dLk[3];
        for (int i = 0; i < 3; ++i) { dl[i] = al[i] - linp[i]; dLk[i] = aL[i] - angp[i]; }
        // move the reference point to the (moving) centre of mass
        const double sc[3] = {ok[0] - w.Cc[0][0], ok[1] - w.Cc[0][1], ok[2] - w.Cc[0][2]};
        const double im = 1.0 / w.Mc[0];
        const double jc[3] = {w.A[0][g] * im, w.A[1][g] * im, w.A[2][g] * im};
        double t2[3];
        cross3(sc, dl, t);
        cross3(jc, &w.hs[0][0], t2);
        for (int i = 0; i < 3; ++i) dL[i] = dLk[i] + t[i] - t2[i];
      }
      for (int i = 0; i < 3; ++i) { w.Dh[i][g] = dl[i]; w.Dh[3 + i][g] = dL[i]; }
    }
    if (WITH_EE) {
      for (int idx = tid; idx < kNumContacts * G; idx += kWave) {
        const int i = idx / G, g = idx % G;
        double col[3] = {0.0, 0.0, 0.0};
        if (g >= 3 && (g < 6 || ((md.contact_path[i] >> (g - 5)) & 1u))) {
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
  // ---- phase L: d v_base / dq = -A_b^{-1} d(A v)/dq
  BP_LANES(tid, kWave) {
    const double im = 1.0 / w.A[0][0];
    for (int idx = tid; idx < 6 * G; idx += kWave) {
      const int i = idx / G, g = idx % G;
      double acc;
      if (i < 3) {
        acc = im * w.Dh[i][g] + w.X12[3 * i] * w.Dh[3][g] + w.X12[3 * i + 1] * w.Dh[4][g] + w.X12[3 * i + 2] * w.Dh[5][g];
      } else {
        const int r = i - 3;
        acc = w.X22[3 * r] * w.Dh[3][g] + w.X22[3 * r + 1] * w.Dh[4][g] + w.X22[3 * r + 2] * w.Dh[5][g];
      }
      w.dvb[i][g] = -acc;
    }
  }
  BP_SYNC();
  // ---- phase M: the nine dense rows of df/dx and df/du
  BP_LANES(tid, kWave) {
    const double im = 1.0 / w.A[0][0];
    const double imt = 1.0 / mass_total;
    for (int idx = tid; idx < 9 * NX; idx += kWave) {
      const int r = idx / NX, c = idx % NX;
      double val = 0.0;
      if (r < 3) {  // d(angular momentum rate)/dq = (1/m) sum_i (J_i - J_com)[:,g] x F_i   (needs J: WITH_EE evaluations
                    // carry it; otherwise rebuilt from the same cross products)
        if (c >= 6) {
          const int g = c - 6;
          const int a1 = (r + 1) % 3, a2 = (r + 2) % 3;
          const double jc1 = w.A[a1][g] / w.Mc[0], jc2 = w.A[a2][g] / w.Mc[0];
          for (int i = 0; i < kNumContacts; ++i) {
            double j1, j2;
            if (WITH_EE) {
              j1 = w.J[3 * i + a1][g];
              j2 = w.J[3 * i + a2][g];
            } else {
              double col[3] = {0.0, 0.0, 0.0};
              if (g < 3) {
                col[g] = 1.0;
              } else if (g < 6 || ((md.contact_path[i] >> (g - 5)) & 1u)) {
                const double rr[3] = {w.cpos[i][0] - w.og[g][0], w.cpos[i][1] - w.og[g][1], w.cpos[i][2] - w.og[g][2]};
                cross3(w.ah[g], rr, col);
              }
              j1 = col[a1];
              j2 = col[a2];
            }
            val += (j1 - jc1) * w.u[3 * i + a2] - (j2 - jc2) * w.u[3 * i + a1];
          }
          val *= imt;
        }
      } else {
        const int i = r - 3;
        if (c < 6) {  // m A_b^{-1}
          double e;
          if (i < 3) e = c < 3 ? (c == i ? im : 0.0) : w.X12[3 * i + (c - 3)];
          else e = c < 3 ? 0.0 : w.X22[3 * (i - 3) + (c - 3)];
          val = mass_total * e;
        } else {
          val = w.dvb[i][c - 6];
        }
      }
      w.Ar[r][c] = val;
    }
    for (int idx = tid; idx < 9 * NU; idx += kWave) {
      const int r = idx / NU, c = idx % NU;
      double val = 0.0;
      if (r < 3) {
        if (c < 12) {  // [p_i - com]_x / m
          const int i = c / 3, k = c % 3;
          if (k != r) {
            const int other = 3 - r - k;
            const double d = (w.cpos[i][other] - w.Cc[0][other]) * imt;
            // (r x F)_r = r_{r+1} F_{r+2} - r_{r+2} F_{r+1}
            val = (k == (r + 2) % 3) ? d : -d;
          }
        }
      } else if (c >= 12) {  // -A_b^{-1} A_j
        const int i = r - 3, g = 6 + (c - 12);
        double acc;
        if (i < 3) acc = im * w.A[i][g] + w.X12[3 * i] * w.A[3][g] + w.X12[3 * i + 1] * w.A[4][g] + w.X12[3 * i + 2] * w.A[5][g];
        else acc = w.X22[3 * (i - 3)] * w.A[3][g] + w.X22[3 * (i - 3) + 1] * w.A[4][g] + w.X22[3 * (i - 3) + 2] * w.A[5][g];
        val = -acc;
      }
      w.Br[r][c] = val;
    }
  }
  BP_SYNC();
}

}  // namespace bpmpc
