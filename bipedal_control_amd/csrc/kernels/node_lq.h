// One shooting node of the multiple-shooting transcription, one wavefront per node.
//
// linearize_node: RK2 sensitivity discretisation of the centroidal dynamics, quadratic model of the tracking cost
// plus soft friction cones (x dt), linear model of the active equality constraints; writes the dense LQ model of the
// node to HBM exactly as the reference materialises it per node ([OCS2-upstream] multiple_shooting::setupIntermediateNode /
// setupEventNode; terms: SURVEY.md section 8 rows a1-a3, a5-a9; assembly order src/BipedalRobotInterface.cpp:151,181-191).
// node_performance: the value-only evaluation used by the filter line search ([OCS2-upstream]
// multiple_shooting::computeIntermediatePerformance / computeEventPerformance).
#pragma once
#include "centroidal_eval.h"

namespace bpmpc {

struct NodeInputs {          // all pointers are to this node's data
  int kind;                  // 0 intermediate, 1 event
  int mode;
  double dt;
  const double* x;           // NX
  const double* u;           // NU
  const double* xnext;       // NX
  const double* xref;        // NX
  const double* zref;        // 4 (swing height reference, only used when pos_gain != 0)
  const double* zdref;       // 4
};

struct NodeLQOut {
  double *A, *B, *b;         // NX*NX, NX*NU, NX
  double *Q, *R, *P, *q, *r; // NX*NX, NU*NU, NU*NX, NX, NU
  double* c;                 // 1
  double *C, *D, *e;         // 16*NX, 16*NU, 16 (rows >= nc zero filled)
  int* nc;
  double* perf;              // cost, dynamics SSE, equality SSE
  double* prof = nullptr;    // debug: cycles per phase (BPMPC_LINEARIZE_PROFILE)
};

template <int NJ>
struct NodeWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  CentroidalWorkspace<NJ> k;
  double x0[NX], xn[NX], xref[NX], dx[NX], du[NU];
  double f1[NX];
  double bvec[NX], evec[kMaxEqRows];
  double cone[kNumContacts][16];   // per contact: value, p, p', p'', grad(3), hess(6 sym: xx xy xz yy yz zz)
  double partial[kWave];
  int row_contact[kMaxEqRows], row_axis[kMaxEqRows], row_type[kMaxEqRows];  // 0 zero force, 1 zero velocity, 2 normal velocity
  int nc;
};

// [OCS2-upstream] RelaxedBarrierPenalty
BP_DEVICE void relaxed_barrier(double mu, double delta, double h, double* p, double* dp, double* ddp) {
  if (!(mu > 0.0)) { *p = 0.0; *dp = 0.0; *ddp = 0.0; return; }      // no penalty configured (hard cone with sqp.inequalityConstraintMu = 0, the upstream default)
  if (h > delta) {
    *p = -mu * log(h);
    *dp = -mu / h;
    *ddp = mu / (h * h);
  } else {
    const double t = (h - 2.0 * delta) / delta;
    *p = mu * (-log(delta) + 0.5 * t * t - 0.5);
    *dp = mu * ((h - 2.0 * delta) / (delta * delta));
    *ddp = mu / (delta * delta);
  }
}

// weight-compensating nominal input (include/ocs2_bipedal_robot/common/utils.h:63-76), component idx
template <class Model>   // DeviceModel or its scalar constants staged in LDS (linearize_fast.h LinFastScalars)
BP_DEVICE double nominal_input(const Model& md, int mode, int idx) {
  if (idx >= 12 || (idx % 3) != 2) return 0.0;
  const int c = idx / 3;
  if (!stance_flag(mode, c)) return 0.0;
  const int n = (mode == 3) ? 4 : ((mode == 1 || mode == 2) ? 2 : 0);
  return md.robot_mass * 9.81 / n;
}

// friction cone value / derivatives of one contact force (src/constraint/FrictionConeConstraint.cpp:129-160, t_R_w = I)
template <class Model>
BP_DEVICE void cone_terms(const Model& md, const double* F, bool deriv, double* out /*16*/) {
  const double Fx2 = F[0] * F[0], Fy2 = F[1] * F[1];
  const double T2 = Fx2 + Fy2 + md.cone_reg, T = sqrt(T2);
  const double h = md.friction * (F[2] + md.cone_grip) - T;
  double p, dp, ddp;
  relaxed_barrier(md.barrier_mu, md.barrier_delta, h, &p, &dp, &ddp);
  out[0] = h; out[1] = p; out[2] = dp; out[3] = ddp;
  if (deriv) {
    const double T32 = T * T2;
    out[4] = -F[0] / T; out[5] = -F[1] / T; out[6] = md.friction;
    out[7] = -(Fy2 + md.cone_reg) / T32;  // xx
    out[8] = F[0] * F[1] / T32;           // xy
    out[9] = 0.0;                         // xz
    out[10] = -(Fx2 + md.cone_reg) / T32; // yy
    out[11] = 0.0;                        // yz
    out[12] = 0.0;                        // zz
    if (md.cone_gauss_newton) {           // hard cone: penalty of the LINEAR approximation of the constraint, p'' dh dh' only (device_model.h)
      out[7] = 0.0; out[8] = 0.0; out[10] = 0.0;
    }
  }
}

// Enumerate the active equality rows of a mode in registration order zeroForce_i, zeroVelocity_i, normalVelocity_i
// (src/BipedalRobotInterface.cpp:187-191).  Executed by one lane.
template <int NJ>
BP_DEVICE void enumerate_rows(int mode, NodeWorkspace<NJ>& ws) {
  int row = 0;
  for (int i = 0; i < kNumContacts; ++i) {
    if (!stance_flag(mode, i)) {
      for (int a = 0; a < 3; ++a) { ws.row_contact[row] = i; ws.row_axis[row] = a; ws.row_type[row] = 0; ++row; }
      ws.row_contact[row] = i; ws.row_axis[row] = 2; ws.row_type[row] = 2; ++row;
    } else {
      for (int a = 0; a < 3; ++a) { ws.row_contact[row] = i; ws.row_axis[row] = a; ws.row_type[row] = 1; ++row; }
    }
  }
  ws.nc = row;
}

template <int NJ>
BP_DEVICE double eq_row_value(const DeviceModel& md, const NodeWorkspace<NJ>& ws, const NodeInputs& in, int row) {
  const int i = ws.row_contact[row], a = ws.row_axis[row], type = ws.row_type[row];
  if (type == 0) return ws.k.u[3 * i + a];                      // ZeroForceConstraint.cpp:58-60
  double g = ws.k.cvel[i][a];                                  // EndEffectorLinearConstraint.cpp:74-85
  if (type == 1) {
    if (md.pos_gain != 0.0 && a == 2) g += md.pos_gain * ws.k.cpos[i][2];    // BipedalRobotInterface.cpp:350-359
  } else {
    g -= in.zdref[i];                                         // BipedalRobotPreComputation.cpp:71-80
    if (md.pos_gain != 0.0) g += md.pos_gain * (ws.k.cpos[i][2] - in.zref[i]);
  }
  return g;
}

// ilqr (the DDP slice, solver.hip run_ddp; [OCS2-upstream, recalled] ILQR::discreteLQWorker): the continuous-time model at the node is
// discretised by ONE Euler step - A = I + dt A_c, B = dt B_c, no dynamics bias (the nominal trajectory of a DDP is a roll-out) - and
// input_shift (hessian_correction::shiftHessian, DIAGONAL_SHIFT: added to every diagonal entry of Hm = R + B' S B, i.e. of R) is added to
// the dt-scaled R.  Cost and constraints as in the multiple-shooting transcription.
template <int NJ>
BP_DEVICE void linearize_node(const DeviceModel& md, NodeWorkspace<NJ>& ws, const NodeInputs& in, const NodeLQOut& out, bool ilqr = false, double input_shift = 0.0) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ, G = 6 + NJ;
  CentroidalWorkspace<NJ>& k = ws.k;

  if (in.kind == 1) {  // event node: identity jump map, no input, no cost
    BP_LANES(tid, kWave) {
      for (int idx = tid; idx < NX * NX; idx += kWave) { out.A[idx] = (idx / NX == idx % NX) ? 1.0 : 0.0; out.Q[idx] = 0.0; }
      for (int idx = tid; idx < NX * NU; idx += kWave) { out.B[idx] = 0.0; out.P[idx] = 0.0; }
      for (int idx = tid; idx < NU * NU; idx += kWave) out.R[idx] = 0.0;
      for (int idx = tid; idx < kMaxEqRows * NX; idx += kWave) out.C[idx] = 0.0;
      for (int idx = tid; idx < kMaxEqRows * NU; idx += kWave) out.D[idx] = 0.0;
      if (tid < kMaxEqRows) out.e[tid] = 0.0;
      double d = 0.0;
      if (tid < NX) {
        d = ilqr ? 0.0 : in.x[tid] - in.xnext[tid];
        out.b[tid] = d;
        out.q[tid] = 0.0;
      }
      if (tid < NU) out.r[tid] = 0.0;
      ws.partial[tid] = d * d;
    }
    BP_SYNC();
    BP_LANES(tid, kWave) {
      if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < NX; ++i) s += ws.partial[i];
        out.c[0] = 0.0;
        out.nc[0] = 0;
        out.perf[0] = 0.0; out.perf[1] = s; out.perf[2] = 0.0;
      }
    }
    return;
  }

#if defined(BPMPC_LINEARIZE_PROFILE) && !defined(BPMPC_HOST_EMULATION)
  long long lq_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long lq_prev = clock64();
#define LQPROF(slot) do { const long long tn_ = clock64(); lq_t[slot] += tn_ - lq_prev; lq_prev = tn_; } while (0)
#else
#define LQPROF(slot) ((void)0)
#endif
  const double dt = in.dt, hdt = 0.5 * in.dt;
  const int mode = in.mode;
  // ---- load the node
  BP_LANES(tid, kWave) {
    if (tid < NX) {
      const double xv = in.x[tid];
      k.x[tid] = xv; ws.x0[tid] = xv;
      ws.xn[tid] = in.xnext[tid];
      ws.xref[tid] = in.xref[tid];
    }
    if (tid < NU) k.u[tid] = in.u[tid];
    if (tid == kWave - 1) enumerate_rows<NJ>(mode, ws);
  }
  cache_model<NJ>(md, k);
  LQPROF(0);
  eval_centroidal<NJ, true, true>(md, k);
  LQPROF(1);

  // ---- equality-constraint rows (linear model) straight to HBM, one lane per column of [C | D]:
  //        d v_i / d(x|u) = J_base (d v_base / d(x|u)) + [0 | d(J_i v)/dq]  resp.  + [0 | J_joints]
  //      the lane keeps its column of d v_base/d(x|u) (rows 3..8 of Ar / Br) in registers across the rows.
  BP_LANES(tid, kWave) {
    const int nc = ws.nc;
    if (tid < NX + NU) {
      const bool is_x = tid < NX;
      const int c = is_x ? tid : tid - NX;
      double vb[6];
      for (int l = 0; l < 6; ++l) vb[l] = is_x ? k.Ar[3 + l][c] : k.Br[3 + l][c];
      double* dst = is_x ? out.C + c : out.D + c;
      const int ld = is_x ? NX : NU;
      for (int row = 0; row < kMaxEqRows; ++row) {
        double val = 0.0;
        if (row < nc) {
          const int i = ws.row_contact[row], a = ws.row_axis[row], type = ws.row_type[row];
          if (type == 0) {
            val = (!is_x && c == 3 * i + a) ? 1.0 : 0.0;                 // ZeroForceConstraint.cpp:64-72
          } else {
            const double* Jr = k.J[3 * i + a];
            for (int l = 0; l < 6; ++l) val += Jr[l] * vb[l];
            if (is_x) {
              if (c >= 6) {
                val += k.DJv[3 * i + a][c - 6];
                if (md.pos_gain != 0.0 && a == 2) val += md.pos_gain * Jr[c - 6];
              }
            } else if (c >= 12) {
              val += Jr[6 + (c - 12)];
            }
          }
        }
        dst[row * ld] = val;
      }
    } else if (tid < NX + NU + kMaxEqRows) {
      const int row = tid - NX - NU;
      const double ev = row < nc ? eq_row_value<NJ>(md, ws, in, row) : 0.0;
      ws.evec[row] = ev;
      out.e[row] = ev;
    }
    if (tid >= 32 && tid < 32 + kNumContacts) {
      const int i = tid - 32;
      if (stance_flag(mode, i)) cone_terms(md, &k.u[3 * i], true, ws.cone[i]);
    }
  }
  BP_SYNC();
  // keep k1 (Ar1/Br1 share LDS with the contact Jacobians, which are dead from here on)
  BP_LANES(tid, kWave) {
    if (tid < NX + NU) {
      const bool is_x = tid < NX;
      const int c = is_x ? tid : tid - NX;
      for (int r = 0; r < 9; ++r) { if (is_x) k.Ar1[r][c] = k.Ar[r][c]; else k.Br1[r][c] = k.Br[r][c]; }
    }
    if (tid < NX) ws.f1[tid] = k.f[tid];
  }
  BP_SYNC();
  BP_LANES(tid, kWave) {
    if (tid < NX) k.x[tid] = ws.x0[tid] + dt * ws.f1[tid];
  }
  BP_SYNC();
  LQPROF(2);
  if (!ilqr) eval_centroidal<NJ, true, false>(md, k);      // (ilqr: the stage-one derivatives are the model; k.Ar / k.Br keep them)
  LQPROF(3);

  // ---- RK2 sensitivities ([OCS2-upstream] SensitivityIntegrator rk2):
  //   A = I + dt/2 (A1 + A2 + dt A2 A1),  B = dt/2 (B1 + B2 + dt A2 B1),  b = x + dt/2 (f1 + f2) - x_next
  // Rows 0..2 and 12.. of A1/A2 vanish and rows 0..2 / 12.. of B1/B2 are constants ([I/m | 0] and [0 | I]).
  // One lane per column of [A | B]: the lane holds its k1 and k2 columns (rows 3..11) in registers, reads the 9x9 block
  // A2[3:12, 3:12] as LDS broadcasts and writes one coalesced row of A and of B per step.
  BP_LANES(tid, kWave) {
    const double imt = 1.0 / md.robot_mass;
    if (tid < NX + NU) {
      const bool is_x = tid < NX;
      const int c = is_x ? tid : tid - NX;
      double c1[9], c2[9];
      for (int l = 0; l < 9; ++l) { c1[l] = is_x ? k.Ar1[l][c] : k.Br1[l][c]; c2[l] = is_x ? k.Ar[l][c] : k.Br[l][c]; }
      double* dst = is_x ? out.A + c : out.B + c;
      const int ld = is_x ? NX : NU;
      for (int r = 0; r < NX; ++r) {
        double val;
        if (r < 3) {
          val = is_x ? (r == c ? 1.0 : 0.0) : ((c < 12 && (c % 3) == r) ? dt * imt : 0.0);
        } else if (r >= 12) {
          val = is_x ? (r == c ? 1.0 : 0.0) : (c == r ? dt : 0.0);
        } else {
          const int rr = r - 3;
          double prod = 0.0;
          for (int l = 0; l < 9; ++l) prod += k.Ar[rr][3 + l] * c1[l];
          if (!is_x) prod += (c < 12) ? k.Ar[rr][c % 3] * imt : k.Ar[rr][c];   // A2[:,0:3] (I/m) and A2[:,12+j] identity blocks of B1
          val = (is_x && r == c ? 1.0 : 0.0) + (ilqr ? dt * c1[rr] : hdt * (c1[rr] + c2[rr] + dt * prod));
        }
        dst[r * ld] = val;
      }
    }
    if (tid < NX) {
      const double bb = ilqr ? 0.0 : ws.x0[tid] + hdt * ws.f1[tid] + hdt * k.f[tid] - ws.xn[tid];
      out.b[tid] = bb;
      ws.bvec[tid] = bb;
      ws.dx[tid] = ws.x0[tid] - ws.xref[tid];
    }
    if (tid < NU) ws.du[tid] = k.u[tid] - nominal_input(md, mode, tid);
  }
  BP_SYNC();
  LQPROF(4);
  // ---- cost: tracking + soft cones, multiplied by dt
  BP_LANES(tid, kWave) {
    // total first-derivative of the barrier over active cones: every active cone shifts ALL diagonal entries by
    // -p' * hessianDiagonalShift (FrictionConeConstraint.cpp:192-205)
    double shift = 0.0;
    for (int i = 0; i < kNumContacts; ++i)
      if (stance_flag(mode, i)) shift += -ws.cone[i][2] * md.cone_shift;
    // lanes 0..NX-1: column c of Q; lanes NX..NX+NU-1: column c of R; the remaining lanes zero P
    if (tid < NX) {
      const int c = tid;
      double col[NX];
      for (int r = 0; r < NX; ++r) col[r] = md.Q[r * NX + c];   // all reads first: the stores below may alias for the compiler
      for (int r = 0; r < NX; ++r) {
        double val = col[r];
        if (r == c) val += shift;
        out.Q[r * NX + c] = dt * val;
      }
    } else if (tid < NX + NU) {
      const int c = tid - NX;
      const bool cone_col = c < 12 && stance_flag(mode, c / 3);
      double col[NU];
      for (int r = 0; r < NU; ++r) col[r] = md.R[r * NU + c];
      for (int r = 0; r < NU; ++r) {
        double val = col[r];
        if (r == c) val += shift;
        if (cone_col && r < 12 && r / 3 == c / 3) {
          const double* cn = ws.cone[r / 3];
          const int a = r % 3, b = c % 3;
          const int lo = a < b ? a : b, hi = a < b ? b : a;
          const int sidx = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);  // xx xy xz yy yz zz
          val += cn[3] * cn[4 + a] * cn[4 + b] + cn[2] * cn[7 + sidx];
        }
        out.R[r * NU + c] = dt * val + ((ilqr && r == c) ? input_shift : 0.0);
      }
    } else {
      for (int idx = tid - NX - NU; idx < NU * NX; idx += kWave - NX - NU) out.P[idx] = 0.0;
    }
    double part = 0.0;
    if (tid < NX) {
      double acc = 0.0;
      for (int l = 0; l < NX; ++l) acc += md.Q[tid * NX + l] * ws.dx[l];
      out.q[tid] = dt * acc;
      part += 0.5 * ws.dx[tid] * acc;
    }
    if (tid < NU) {
      double acc = 0.0;
      for (int l = 0; l < NU; ++l) acc += md.R[tid * NU + l] * ws.du[l];
      part += 0.5 * ws.du[tid] * acc;
      if (tid < 12 && stance_flag(mode, tid / 3)) acc += ws.cone[tid / 3][2] * ws.cone[tid / 3][4 + tid % 3];
      out.r[tid] = dt * acc;
    }
    ws.partial[tid] = part;
  }
  BP_SYNC();
  BP_LANES(tid, kWave) {
    if (tid == 0) {
      double c = 0.0;
      for (int i = 0; i < kWave; ++i) c += ws.partial[i];
      for (int i = 0; i < kNumContacts; ++i)
        if (stance_flag(mode, i)) c += ws.cone[i][1];
      double dyn = 0.0, eq = 0.0;
      for (int i = 0; i < NX; ++i) dyn += ws.bvec[i] * ws.bvec[i];
      for (int i = 0; i < ws.nc; ++i) eq += ws.evec[i] * ws.evec[i];
      out.c[0] = dt * c;
      out.nc[0] = ws.nc;
      out.perf[0] = dt * c; out.perf[1] = dt * dyn; out.perf[2] = dt * eq;
    }
  }
  LQPROF(5);
#if defined(BPMPC_LINEARIZE_PROFILE) && !defined(BPMPC_HOST_EMULATION)
  if (out.prof && threadIdx.x == 0) for (int i = 0; i < 8; ++i) out.prof[i] = (double)lq_t[i];
#endif
  (void)G;
}

// Value-only metrics of one node for the line search.
template <int NJ>
BP_DEVICE void node_performance(const DeviceModel& md, NodeWorkspace<NJ>& ws, const NodeInputs& in, double* perf) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  CentroidalWorkspace<NJ>& k = ws.k;
  if (in.kind == 1) {
    BP_LANES(tid, kWave) {
      double d = 0.0;
      if (tid < NX) d = in.x[tid] - in.xnext[tid];
      ws.partial[tid] = d * d;
    }
    BP_SYNC();
    BP_LANES(tid, kWave) {
      if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < NX; ++i) s += ws.partial[i];
        perf[0] = 0.0; perf[1] = s; perf[2] = 0.0;
      }
    }
    return;
  }
  const double dt = in.dt, hdt = 0.5 * in.dt;
  const int mode = in.mode;
  BP_LANES(tid, kWave) {
    if (tid < NX) {
      const double xv = in.x[tid];
      k.x[tid] = xv; ws.x0[tid] = xv;
      ws.xn[tid] = in.xnext[tid];
      ws.dx[tid] = xv - in.xref[tid];
    }
    if (tid < NU) {
      const double uv = in.u[tid];
      k.u[tid] = uv;
      ws.du[tid] = uv - nominal_input(md, mode, tid);
    }
    if (tid == kWave - 1) enumerate_rows<NJ>(mode, ws);
  }
  cache_model<NJ>(md, k);
  eval_centroidal<NJ, false, true>(md, k);
  BP_LANES(tid, kWave) {
    if (tid < kMaxEqRows) ws.evec[tid] = tid < ws.nc ? eq_row_value<NJ>(md, ws, in, tid) : 0.0;
    if (tid >= 32 && tid < 32 + kNumContacts) {
      const int i = tid - 32;
      if (stance_flag(mode, i)) cone_terms(md, &k.u[3 * i], false, ws.cone[i]);
    }
    if (tid < NX) ws.f1[tid] = k.f[tid];
  }
  BP_SYNC();
  BP_LANES(tid, kWave) {
    if (tid < NX) k.x[tid] = ws.x0[tid] + dt * ws.f1[tid];
  }
  BP_SYNC();
  eval_centroidal<NJ, false, false>(md, k);
  BP_LANES(tid, kWave) {
    double part = 0.0;
    if (tid < NX) {
      ws.bvec[tid] = ws.x0[tid] + hdt * ws.f1[tid] + hdt * k.f[tid] - ws.xn[tid];
      double acc = 0.0;
      for (int l = 0; l < NX; ++l) acc += md.Q[tid * NX + l] * ws.dx[l];
      part += 0.5 * ws.dx[tid] * acc;
    }
    if (tid < NU) {
      double acc = 0.0;
      for (int l = 0; l < NU; ++l) acc += md.R[tid * NU + l] * ws.du[l];
      part += 0.5 * ws.du[tid] * acc;
    }
    ws.partial[tid] = part;
  }
  BP_SYNC();
  BP_LANES(tid, kWave) {
    if (tid == 0) {
      double c = 0.0;
      for (int i = 0; i < kWave; ++i) c += ws.partial[i];
      for (int i = 0; i < kNumContacts; ++i)
        if (stance_flag(mode, i)) c += ws.cone[i][1];
      double dyn = 0.0, eq = 0.0;
      for (int i = 0; i < NX; ++i) dyn += ws.bvec[i] * ws.bvec[i];
      for (int i = 0; i < ws.nc; ++i) eq += ws.evec[i] * ws.evec[i];
      perf[0] = dt * c; perf[1] = dt * dyn; perf[2] = dt * eq;
    }
  }
}

}  // namespace bpmpc
