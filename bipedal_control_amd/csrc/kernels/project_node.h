// Elimination of the state-input equality constraints of one node, one wavefront per node.
//
// [OCS2-upstream] multiple_shooting::projectTranscription -> LinearAlgebra::luConstraintProjection +
// changeOfInputVariables (enabled by sqp.projectStateInputEqualityConstraints = true,
// bipedal_robot_example/unitree_h1/h1_ocs2_config/config/task/task.info:76):
//     C dx + D du + e = 0   ->   du = Px dx + Pu dut + Pe,   Pu = null(D),  Px = -D^+ C,  Pe = -D^+ e
// with the factorisation semantics of Eigen::FullPivLU (complete pivoting, first maximum in column-major order,
// rank threshold eps * min(rows, cols) * |max pivot|, free variables of solve() = 0).  D is rank deficient for this
// robot (two contact points per rigid foot), so the rank decision is part of the result.
// Then the node's LQ model is rewritten in the reduced input dut (dimension nut = nu - rank).
#pragma once
#include "../device_model.h"
#include "lane_model.h"

namespace bpmpc {

struct ProjectIn {
  int kind;                       // 1: event node -> passthrough, nut = 0
  int nc;
  const double *C, *D, *e;        // 16*NX, 16*NU, 16
  const double *A, *B, *b, *Q, *R, *P, *q, *r;
  const double* qrd = nullptr;    // compact node-dependent part of Q, R (linearize_fast.h kQrdStride), fast kernels only
  int r_shift_at = 0;             // entry of the record that holds the shift of R's diagonal: 0 (the Hessian shift, as Q's) or kQrdRShift (ILQR: + DIAGONAL_SHIFT / dt)
  int mode = 3;                   // contact mode of the node (fast kernels after the structured elimination: the force rows of [Px | Pe | Pu] are generated from it)
  const double* zero = nullptr;   // a 0.0 in global memory (masked loads by address)
  const double* Vt = nullptr;     // joint rows of the packed [Px | Pe | Pu] (project_lu_s.h, row stride PackedLq::WP), structured fast path only
};
struct ProjectOut {
  double *Px, *Pu, *Pe;           // NU*NX, NU*NU (first nut columns), NU
  int* nut;
  double *At, *Bt, *bt;           // NX*NX, NX*NU (first nut columns, stride NU), NX
  double *Qt, *Rt, *Pt, *qt, *rt; // NX*NX, NU*NU (stride NU), NU*NX, NX, NU
  double *Wt = nullptr, *Qp = nullptr, *Mt = nullptr;   // the same model in the packed layout of the fast kernels (PackedLq)
  double* Vt = nullptr;           // dense fast path: the joint rows of [Px | Pe | Pu] packed for the sweep's loaders (the structured elimination writes them itself)
};

// The projected LQ model as the fast kernels exchange it (project_mfma.h writes, riccati_mfma*.h stage it into LDS unchanged):
//     Wt [nx][WP] = [At | bt | Bt]      Qp [nx][QP] = [Qt | qt]      Mt [nu][WP] = [Pt | rt | Rt]
// row-major with row strides of whole 16-column blocks, i.e. the column layout of the packed products.  A 16x16 accumulator
// block of the matrix cores then leaves as 16 full, aligned 128-byte row segments under a row predicate only, where the
// separate 22-wide matrices needed a three-way column predicate per store and 64-bit address arithmetic for each piece.
// Contract: columns >= 16 nbc (nbc = block columns of nx + 1 + nut) and rows >= nut of Mt are NOT written; the reader masks them.
template <int NJ>
struct PackedLq {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int WP = ((NX + 1 + NU + 15) / 16) * 16, QP = 32;
  static constexpr int W_SIZE = NX * WP, Q_SIZE = NX * QP, M_SIZE = NU * WP;
  static_assert(NX + 1 <= QP, "column nx inside the second block column");
};

template <int NJ>
struct ProjectWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ, NR = NX + 1 + NU;
  double lu[kMaxEqRows][NU];
  double rhs[kMaxEqRows][NR];     // [C | e | (U12 copy for the kernel solve)]
  double Px[NU][NX], Pu[NU][NU], Pe[NU];
  double B[NX][NU], R[NU][NU], Pc[NU][NX];   // staged inputs (Pc = cost cross term P)
  double RPx[NU][NX], RPu[NU][NU];
  double rr[NU];
  double redv[kWave];
  int redi[kWave];
  int colidx[NU];
  int pivot_row, pivot_col, rank, nonzero;
  double maxpivot;
};

template <int NJ>
BP_DEVICE void project_node(ProjectWorkspace<NJ>& ws, const ProjectIn& in, const ProjectOut& out) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ, NR = NX + 1 + NU;

  if (in.kind == 1) {  // event node: no input, dynamics and (zero) cost pass through
    BP_LANES(tid, kWave) {
      for (int idx = tid; idx < NX * NX; idx += kWave) { out.At[idx] = in.A[idx]; out.Qt[idx] = in.Q[idx]; }
      for (int idx = tid; idx < NX * NU; idx += kWave) { out.Bt[idx] = 0.0; out.Pt[idx] = 0.0; out.Px[idx] = 0.0; }
      for (int idx = tid; idx < NU * NU; idx += kWave) { out.Rt[idx] = 0.0; out.Pu[idx] = 0.0; }
      if (tid < NX) { out.bt[tid] = in.b[tid]; out.qt[tid] = in.q[tid]; }
      if (tid < NU) { out.rt[tid] = 0.0; out.Pe[tid] = 0.0; }
      if (tid == 0) out.nut[0] = 0;
    }
    return;
  }

  const int rows = in.nc, cols = NU;
  const int size = rows < cols ? rows : cols;
  // ---- stage inputs
  BP_LANES(tid, kWave) {
    for (int idx = tid; idx < kMaxEqRows * NU; idx += kWave) ws.lu[idx / NU][idx % NU] = in.D[idx];
    for (int idx = tid; idx < kMaxEqRows * NX; idx += kWave) ws.rhs[idx / NX][idx % NX] = in.C[idx];
    if (tid < kMaxEqRows) ws.rhs[tid][NX] = in.e[tid];
    for (int idx = tid; idx < NX * NU; idx += kWave) { ws.B[idx / NU][idx % NU] = in.B[idx]; ws.Pc[idx / NX][idx % NX] = in.P[idx]; }
    for (int idx = tid; idx < NU * NU; idx += kWave) ws.R[idx / NU][idx % NU] = in.R[idx];
    for (int idx = tid; idx < NU * NX; idx += kWave) ws.Px[idx / NX][idx % NX] = 0.0;
    for (int idx = tid; idx < NU * NU; idx += kWave) ws.Pu[idx / NU][idx % NU] = 0.0;
    if (tid < NU) { ws.colidx[tid] = tid; ws.Pe[tid] = 0.0; }
    if (tid == 0) { ws.maxpivot = 0.0; ws.nonzero = size; }
  }
  BP_SYNC();
  // ---- LU with complete pivoting; the unit-lower solve of the right-hand sides is folded into the elimination
  for (int k = 0; k < size; ++k) {
    BP_LANES(tid, kWave) {
      // first maximum of |a_ij| over the trailing block in column-major order
      const int h = rows - k, wdt = cols - k;
      double best = -1.0;
      int bidx = 0x7fffffff;
      for (int l = tid; l < h * wdt; l += kWave) {
        const double a = fabs(ws.lu[k + l % h][k + l / h]);
        if (a > best) { best = a; bidx = l; }
      }
      ws.redv[tid] = best;
      ws.redi[tid] = bidx;
    }
    BP_SYNC();
    for (int stride = kWave / 2; stride >= 1; stride >>= 1) {
      BP_LANES(tid, kWave) {
        if (tid < stride) {
          const double v2 = ws.redv[tid + stride];
          const int i2 = ws.redi[tid + stride];
          if (v2 > ws.redv[tid] || (v2 == ws.redv[tid] && i2 < ws.redi[tid])) { ws.redv[tid] = v2; ws.redi[tid] = i2; }
        }
      }
      BP_SYNC();
    }
    const double best = ws.redv[0];
    if (best == 0.0) {  // the rest of the matrix is exactly zero (wave-uniform branch)
      BP_LANES(tid, kWave) { if (tid == 0) ws.nonzero = k; }
      BP_SYNC();
      break;
    }
    const int h = rows - k;
    const int pr = k + ws.redi[0] % h, pc = k + ws.redi[0] / h;
    BP_SYNC();  // everyone has read redv/redi before they are reused
    BP_LANES(tid, kWave) {
      if (tid == 0 && best > ws.maxpivot) ws.maxpivot = best;
      if (pr != k) {
        for (int j = tid; j < cols; j += kWave) { const double t = ws.lu[k][j]; ws.lu[k][j] = ws.lu[pr][j]; ws.lu[pr][j] = t; }
        for (int j = tid; j < NX + 1; j += kWave) { const double t = ws.rhs[k][j]; ws.rhs[k][j] = ws.rhs[pr][j]; ws.rhs[pr][j] = t; }
      }
    }
    BP_SYNC();
    BP_LANES(tid, kWave) {
      if (pc != k) {
        for (int i = tid; i < rows; i += kWave) { const double t = ws.lu[i][k]; ws.lu[i][k] = ws.lu[i][pc]; ws.lu[i][pc] = t; }
        if (tid == kWave - 1) { const int t = ws.colidx[k]; ws.colidx[k] = ws.colidx[pc]; ws.colidx[pc] = t; }
      }
    }
    BP_SYNC();
    BP_LANES(tid, kWave) {
      const double piv = ws.lu[k][k];
      for (int i = k + 1 + tid; i < rows; i += kWave) ws.lu[i][k] /= piv;
    }
    BP_SYNC();
    BP_LANES(tid, kWave) {
      const int h2 = rows - k - 1, w2 = cols - k - 1;
      for (int l = tid; l < h2 * w2; l += kWave) {
        const int i = k + 1 + l / w2, j = k + 1 + l % w2;
        ws.lu[i][j] -= ws.lu[i][k] * ws.lu[k][j];
      }
      for (int l = tid; l < h2 * (NX + 1); l += kWave) {
        const int i = k + 1 + l / (NX + 1), j = l % (NX + 1);
        ws.rhs[i][j] -= ws.lu[i][k] * ws.rhs[k][j];
      }
    }
    BP_SYNC();
  }
  // ---- rank
  BP_LANES(tid, kWave) {
    if (tid == 0) {
      const double thr = fabs(ws.maxpivot) * (2.220446049250313e-16 * size);
      int r = 0;
      for (int i = 0; i < ws.nonzero; ++i) r += (fabs(ws.lu[i][i]) > thr) ? 1 : 0;
      ws.rank = r;
    }
  }
  BP_SYNC();
  const int rank = ws.rank, nut = NU - rank;
  // ---- back substitution with U11 on [c | U12] (one lane per right-hand-side column)
  BP_LANES(tid, kWave) {
    for (int i = tid; i < rank * nut; i += kWave) ws.rhs[i / nut][NX + 1 + i % nut] = ws.lu[i / nut][rank + i % nut];
  }
  BP_SYNC();
  BP_LANES(tid, kWave) {
    if (tid < NX + 1 + nut) {
      for (int i = rank - 1; i >= 0; --i) {
        double t = ws.rhs[i][tid];
        for (int l = i + 1; l < rank; ++l) t -= ws.lu[i][l] * ws.rhs[l][tid];
        ws.rhs[i][tid] = t / ws.lu[i][i];
      }
    }
  }
  BP_SYNC();
  // ---- scatter through the column permutation:  Px = -Q [y; 0],  Pe likewise,  Pu = Q [-U11^{-1} U12; I]
  BP_LANES(tid, kWave) {
    for (int idx = tid; idx < rank * NX; idx += kWave) ws.Px[ws.colidx[idx / NX]][idx % NX] = -ws.rhs[idx / NX][idx % NX];
    if (tid < rank) ws.Pe[ws.colidx[tid]] = -ws.rhs[tid][NX];
    for (int idx = tid; idx < rank * nut; idx += kWave) ws.Pu[ws.colidx[idx / nut]][idx % nut] = -ws.rhs[idx / nut][NX + 1 + idx % nut];
    if (tid < nut) ws.Pu[ws.colidx[rank + tid]][tid] = 1.0;
  }
  BP_SYNC();
  // ---- change of input variables: products that are reused
  BP_LANES(tid, kWave) {
    for (int idx = tid; idx < NU * NX; idx += kWave) {
      const int i = idx / NX, j = idx % NX;
      double t = 0.0;
      for (int l = 0; l < NU; ++l) t += ws.R[i][l] * ws.Px[l][j];
      ws.RPx[i][j] = t;
      out.Px[idx] = ws.Px[i][j];
    }
    for (int idx = tid; idx < NU * NU; idx += kWave) {
      const int i = idx / NU, j = idx % NU;
      double t = 0.0;
      if (j < nut)
        for (int l = 0; l < NU; ++l) t += ws.R[i][l] * ws.Pu[l][j];
      ws.RPu[i][j] = t;
      out.Pu[idx] = ws.Pu[i][j];
    }
    if (tid < NU) {
      double t = in.r[tid];
      for (int l = 0; l < NU; ++l) t += ws.R[tid][l] * ws.Pe[l];
      ws.rr[tid] = t;  // r + R Pe
      out.Pe[tid] = ws.Pe[tid];
    }
    if (tid == 0) out.nut[0] = nut;
  }
  BP_SYNC();
  BP_LANES(tid, kWave) {
    // dynamics: At = A + B Px, Bt = B Pu, bt = b + B Pe
    for (int idx = tid; idx < NX * NX; idx += kWave) {
      const int i = idx / NX, j = idx % NX;
      double t = in.A[idx];
      for (int l = 0; l < NU; ++l) t += ws.B[i][l] * ws.Px[l][j];
      out.At[idx] = t;
      // Qt = Q + Px^T P + P^T Px + Px^T R Px
      double s = in.Q[idx];
      for (int l = 0; l < NU; ++l) s += ws.Px[l][i] * ws.Pc[l][j] + ws.Pc[l][i] * ws.Px[l][j] + ws.Px[l][i] * ws.RPx[l][j];
      out.Qt[idx] = s;
    }
    for (int idx = tid; idx < NX * NU; idx += kWave) {
      const int i = idx / NU, j = idx % NU;
      double t = 0.0;
      if (j < nut)
        for (int l = 0; l < NU; ++l) t += ws.B[i][l] * ws.Pu[l][j];
      out.Bt[idx] = t;
    }
    for (int idx = tid; idx < NU * NX; idx += kWave) {  // Pt = Pu^T (P + R Px)
      const int i = idx / NX, j = idx % NX;
      double t = 0.0;
      if (i < nut)
        for (int l = 0; l < NU; ++l) t += ws.Pu[l][i] * (ws.Pc[l][j] + ws.RPx[l][j]);
      out.Pt[idx] = t;
    }
    for (int idx = tid; idx < NU * NU; idx += kWave) {  // Rt = Pu^T R Pu
      const int i = idx / NU, j = idx % NU;
      double t = 0.0;
      if (i < nut && j < nut)
        for (int l = 0; l < NU; ++l) t += ws.Pu[l][i] * ws.RPu[l][j];
      out.Rt[idx] = t;
    }
    if (tid < NX) {
      double t = in.b[tid];
      for (int l = 0; l < NU; ++l) t += ws.B[tid][l] * ws.Pe[l];
      out.bt[tid] = t;
      // qt = q + P^T Pe + Px^T (r + R Pe)
      double s = in.q[tid];
      for (int l = 0; l < NU; ++l) s += ws.Pc[l][tid] * ws.Pe[l] + ws.Px[l][tid] * ws.rr[l];
      out.qt[tid] = s;
    }
    if (tid < NU) {
      double t = 0.0;
      if (tid < nut)
        for (int l = 0; l < NU; ++l) t += ws.Pu[l][tid] * ws.rr[l];
      out.rt[tid] = t;
    }
  }
  (void)NR;
}

}  // namespace bpmpc
