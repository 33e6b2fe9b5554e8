// Backward Riccati sweep + forward roll-out of the projected (equality-free) QP of one MPC problem,
// one workgroup per problem (sequential over the horizon, parallel inside each stage).
//
// [OCS2-upstream] The reference hands this QP to HPIPM (hpipm_catkin, call path SqpSolver::getOCPSolution); without
// inequality rows HPIPM performs one Riccati factorise + solve, and with sqp.useFeedbackPolicy = true (task.info:80)
// extracts the feedback gains.  Minimiser and gains are unique, so this exact Riccati recursion returns the same
// (dx, du, K) up to round-off.  No terminal cost is registered (only src/BipedalRobotInterface.cpp:151 adds a cost)
// so the recursion starts from S_N = 0.  The forward pass also maps the reduced input back
// (du = Px dx + Pu dut + Pe, K = Px + Pu Kt: [OCS2-upstream] remapProjectedInput / remapProjectedGain) and accumulates
// the Armijo descent metric sum q~'dx + r~'dut and the step norms.
#pragma once
#include "../device_model.h"
#include "lane_model.h"

namespace bpmpc {

constexpr int kRiccatiThreads = 256;

template <int NJ>
struct RiccatiWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  double S[NX][NX], s[NX];
  double A[NX][NX], B[NX][NU], b[NX];
  double SA[NX][NX], SB[NX][NU], Sb[NX];
  double G[NU][NX + 1];        // [G | g]
  double Ks[NU][NX + 1];       // [Kt | kt] of the current stage
  double H[NU][NU];
  double Sn[NX][NX];
  double dx[NX], dxn[NX], dut[NU], du[NU];
  double red[kRiccatiThreads];
  int status;
};

// Per-problem views; node k data at base + k * stride
struct RiccatiIO {
  int N;                                   // number of intervals of this problem
  const int* nut;                          // [N]
  const double *At, *Bt, *bt, *Qt, *Rt, *Pt, *qt, *rt;   // projected LQ, per node
  const double *Px, *Pu, *Pe;              // projection, per node
  const double* dx0;                       // NX: x_measured - x_0
  double *Kt, *kt;                         // scratch per node: NU*NX, NU
  double *dx, *du;                         // outputs: (N+1)*NX, N*NU
  double* K;                               // optional output N*NU*NX (nullable)
  double* summary;                         // 4: armijo descent metric, |dx|^2, |du|^2, status (0 ok, 1 Cholesky failure)
};

template <int NJ>
BP_DEVICE void riccati_problem(RiccatiWorkspace<NJ>& ws, const RiccatiIO& io) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ, NT = kRiccatiThreads;
  const int N = io.N;
  BP_LANES(tid, NT) {
    for (int idx = tid; idx < NX * NX; idx += NT) ws.S[idx / NX][idx % NX] = 0.0;
    if (tid < NX) ws.s[tid] = 0.0;
    if (tid == 0) ws.status = 0;
  }
  BP_SYNC();
  for (int k = N - 1; k >= 0; --k) {
    const int nt = io.nut[k];
    const double* At = io.At + (size_t)k * NX * NX;
    const double* Bt = io.Bt + (size_t)k * NX * NU;
    const double* bt = io.bt + (size_t)k * NX;
    const double* Qt = io.Qt + (size_t)k * NX * NX;
    const double* Rt = io.Rt + (size_t)k * NU * NU;
    const double* Pt = io.Pt + (size_t)k * NU * NX;
    const double* qt = io.qt + (size_t)k * NX;
    const double* rt = io.rt + (size_t)k * NU;
    BP_LANES(tid, NT) {
      for (int idx = tid; idx < NX * NX; idx += NT) ws.A[idx / NX][idx % NX] = At[idx];
      for (int idx = tid; idx < NX * NU; idx += NT) ws.B[idx / NU][idx % NU] = Bt[idx];
      if (tid < NX) ws.b[tid] = bt[tid];
    }
    BP_SYNC();
    BP_LANES(tid, NT) {
      for (int idx = tid; idx < NX * NX; idx += NT) {
        const int i = idx / NX, j = idx % NX;
        double t = 0.0;
        for (int l = 0; l < NX; ++l) t += ws.S[i][l] * ws.A[l][j];
        ws.SA[i][j] = t;
      }
      for (int idx = tid; idx < NX * nt; idx += NT) {
        const int i = idx / nt, j = idx % nt;
        double t = 0.0;
        for (int l = 0; l < NX; ++l) t += ws.S[i][l] * ws.B[l][j];
        ws.SB[i][j] = t;
      }
      if (tid < NX) {
        double t = ws.s[tid];
        for (int l = 0; l < NX; ++l) t += ws.S[tid][l] * ws.b[l];
        ws.Sb[tid] = t;
      }
    }
    BP_SYNC();
    BP_LANES(tid, NT) {
      for (int idx = tid; idx < nt * (NX + 1); idx += NT) {
        const int i = idx / (NX + 1), j = idx % (NX + 1);
        double t;
        if (j < NX) {
          t = Pt[i * NX + j];
          for (int l = 0; l < NX; ++l) t += ws.B[l][i] * ws.SA[l][j];
        } else {
          t = rt[i];
          for (int l = 0; l < NX; ++l) t += ws.B[l][i] * ws.Sb[l];
        }
        ws.G[i][j] = t;
      }
      for (int idx = tid; idx < nt * nt; idx += NT) {
        const int i = idx / nt, j = idx % nt;
        double t = Rt[i * NU + j];
        for (int l = 0; l < NX; ++l) t += ws.B[l][i] * ws.SB[l][j];
        ws.H[i][j] = t;
      }
    }
    BP_SYNC();
    // Cholesky H = L L^T (lower, in place), column by column
    for (int j = 0; j < nt; ++j) {
      BP_LANES(tid, NT) {
        if (tid == 0) {
          double d = ws.H[j][j];
          for (int l = 0; l < j; ++l) d -= ws.H[j][l] * ws.H[j][l];
          if (!(d > 0.0)) { ws.status = 1; d = 1.0; }
          ws.H[j][j] = sqrt(d);
        }
      }
      BP_SYNC();
      BP_LANES(tid, NT) {
        const int i = j + 1 + tid;
        if (i < nt) {
          double t = ws.H[i][j];
          for (int l = 0; l < j; ++l) t -= ws.H[i][l] * ws.H[j][l];
          ws.H[i][j] = t / ws.H[j][j];
        }
      }
      BP_SYNC();
    }
    // [Kt | kt] = -H^{-1} [G | g]: one thread per right-hand-side column
    double* Ktk = io.Kt + (size_t)k * NU * NX;
    double* ktk = io.kt + (size_t)k * NU;
    BP_LANES(tid, NT) {
      if (tid < NX + 1) {
        double y[NU];
        for (int i = 0; i < nt; ++i) {
          double t = -ws.G[i][tid];
          for (int l = 0; l < i; ++l) t -= ws.H[i][l] * y[l];
          y[i] = t / ws.H[i][i];
        }
        for (int i = nt - 1; i >= 0; --i) {
          double t = y[i];
          for (int l = i + 1; l < nt; ++l) t -= ws.H[l][i] * y[l];
          y[i] = t / ws.H[i][i];
        }
        for (int i = 0; i < NU; ++i) ws.Ks[i][tid] = i < nt ? y[i] : 0.0;
      }
    }
    BP_SYNC();
    BP_LANES(tid, NT) {
      for (int idx = tid; idx < NU * NX; idx += NT) Ktk[idx] = ws.Ks[idx / NX][idx % NX];
      if (tid < NU) ktk[tid] = ws.Ks[tid][NX];
    }
    BP_LANES(tid, NT) {
      for (int idx = tid; idx < NX * NX; idx += NT) {
        const int i = idx / NX, j = idx % NX;
        double t = Qt[idx];
        for (int l = 0; l < NX; ++l) t += ws.A[l][i] * ws.SA[l][j];
        for (int l = 0; l < nt; ++l) t += ws.G[l][i] * ws.Ks[l][j];
        ws.Sn[i][j] = t;
      }
      if (tid < NX) {
        double t = qt[tid];
        for (int l = 0; l < NX; ++l) t += ws.A[l][tid] * ws.Sb[l];
        for (int l = 0; l < nt; ++l) t += ws.G[l][tid] * ws.Ks[l][NX];
        ws.s[tid] = t;
      }
    }
    BP_SYNC();
    BP_LANES(tid, NT) {
      for (int idx = tid; idx < NX * NX; idx += NT) {
        const int i = idx / NX, j = idx % NX;
        ws.S[i][j] = 0.5 * (ws.Sn[i][j] + ws.Sn[j][i]);
      }
    }
    BP_SYNC();
  }
  // ---- forward pass
  BP_LANES(tid, NT) {
    if (tid < NX) { ws.dx[tid] = io.dx0[tid]; io.dx[tid] = io.dx0[tid]; }
    ws.red[tid] = 0.0;
  }
  BP_SYNC();
  for (int k = 0; k < N; ++k) {
    const int nt = io.nut[k];
    const double* At = io.At + (size_t)k * NX * NX;
    const double* Bt = io.Bt + (size_t)k * NX * NU;
    const double* bt = io.bt + (size_t)k * NX;
    const double* qt = io.qt + (size_t)k * NX;
    const double* rt = io.rt + (size_t)k * NU;
    const double* Px = io.Px + (size_t)k * NU * NX;
    const double* Pu = io.Pu + (size_t)k * NU * NU;
    const double* Pe = io.Pe + (size_t)k * NU;
    const double* Ktk = io.Kt + (size_t)k * NU * NX;
    const double* ktk = io.kt + (size_t)k * NU;
    BP_LANES(tid, NT) {
      if (tid < NU) {
        double t = 0.0;
        if (tid < nt) {
          t = ktk[tid];
          for (int l = 0; l < NX; ++l) t += Ktk[tid * NX + l] * ws.dx[l];
        }
        ws.dut[tid] = t;
      }
    }
    BP_SYNC();
    BP_LANES(tid, NT) {
      double acc = 0.0;  // contributions to [armijo, |dx|^2, |du|^2] are folded into three strided slots of red[]
      if (tid < NX) {
        double t = bt[tid];
        for (int l = 0; l < NX; ++l) t += At[tid * NX + l] * ws.dx[l];
        for (int l = 0; l < nt; ++l) t += Bt[tid * NU + l] * ws.dut[l];
        ws.dxn[tid] = t;
        io.dx[(size_t)(k + 1) * NX + tid] = t;
        acc = qt[tid] * ws.dx[tid];
        ws.red[tid] += acc;                              // slot group 0: armijo (state part)
        ws.red[64 + tid] += ws.dx[tid] * ws.dx[tid];     // slot group 1: |dx|^2
      } else if (tid >= 64 && tid < 64 + NU) {
        const int i = tid - 64;
        double t = Pe[i];
        for (int l = 0; l < NX; ++l) t += Px[i * NX + l] * ws.dx[l];
        for (int l = 0; l < nt; ++l) t += Pu[i * NU + l] * ws.dut[l];
        io.du[(size_t)k * NU + i] = t;
        ws.red[128 + i] += t * t;                        // slot group 2: |du|^2
        if (i < nt) ws.red[192 + i] += rt[i] * ws.dut[i];  // slot group 3: armijo (input part)
      }
      if (io.K != nullptr) {
        double* Kk = io.K + (size_t)k * NU * NX;
        for (int idx = tid; idx < NU * NX; idx += NT) {
          const int i = idx / NX, j = idx % NX;
          double t = Px[idx];
          for (int l = 0; l < nt; ++l) t += Pu[i * NU + l] * Ktk[l * NX + j];
          Kk[idx] = t;
        }
      }
    }
    BP_SYNC();
    BP_LANES(tid, NT) {
      if (tid < NX) ws.dx[tid] = ws.dxn[tid];
    }
    BP_SYNC();
  }
  BP_LANES(tid, NT) {
    if (tid < NX) ws.red[64 + tid] += ws.dx[tid] * ws.dx[tid];  // terminal state joins the norm
  }
  BP_SYNC();
  BP_LANES(tid, NT) {
    if (tid == 0) {
      double arm = 0.0, nx2 = 0.0, nu2 = 0.0;
      for (int i = 0; i < 64; ++i) { arm += ws.red[i] + ws.red[192 + i]; nx2 += ws.red[64 + i]; nu2 += ws.red[128 + i]; }
      io.summary[0] = arm;
      io.summary[1] = nx2;
      io.summary[2] = nu2;
      io.summary[3] = (double)ws.status;
    }
  }
}

}  // namespace bpmpc
