// Constraint elimination, part two, STRUCTURED (HIP only; the dense version is project_mfma.h, the reference body project_node.h):
// the change of input variables  du = Px dx + Pu du~ + Pe  of one node on the FP64 matrix cores, using what the structured
// constraint elimination (project_lu_s.h) and this problem family guarantee about X = [Px | Pe | Pu]:
//   force rows c < 12 of X:  zero in the state columns, Pe_c (= -F_c for a swing component, 0 in stance) in column nx, and for a stance
//                            component a single 1 in its own reduced-input column nx + 1 + s (s = position among the stance components);
//   joint rows 12 + j:       V_j = [Px_v | pe_v | 0 (stance columns) | Z_v]  - the only dense part, NJ rows;
//   R = blkdiag(R_FF, R_vv)  (BipedalRobotInterface.cpp:239-271: force block and J'R J block; checked when the model is built),
//   B: rows 0..2 dt/m on the matching force components, rows 12 + j dt on joint j, rows 3..11 dense (linearize_fast.h).
// The three products of project_mfma.h then need the JOINT rows only as their inner dimension (3 k-steps instead of 6):
//     RV          = R_vv V + [0 | r_v | 0]                                                   1 x NBC blocks
//     [At|bt|Bt]  = [A | b + B_F Pe_F | B_F(stance) | 0] + B_v V     block row 0 by MFMA, block row 1 (joint rows: dt V + identity) by copy
//     V' RV + [Q | q | 0 ; 0]  and, added to the rows of the stance components,  [0 | r_F + R_FF Pe_F | R_FF(stance, stance) | 0]
// 42 matrix-core instructions per node with three block columns (24 with two) where the dense kernel issues 120 (72), and half of its
// operand gathers.  Same packed outputs (PackedLq: Wt, Qp, Mt), same values up to rounding: the skipped terms are exact zeros, the
// copied ones products with exact ones - only the summation order changes (tests/test_gpu_parity.py compares the two kernels' buffers).
#pragma once
#include <hip/hip_runtime.h>

#include "project_node.h"
#include "riccati_mfma.h"   // v4d, lds_wave_sync

namespace bpmpc {

template <int NJ>
struct ProjectStructWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int KJ = ((NJ + 3) / 4) * 4;                  // joint rows rounded up to the k-step
  static constexpr int NBC_MAX = 3;                               // as project_mfma.h: nx + 1 + nut <= 48
  static constexpr int LDV = 16 * NBC_MAX + 2;
  alignas(16) double V[KJ][LDV];        // joint rows of [Px | Pe | Pu], zero padded
  alignas(16) double RV[KJ][LDV];       // R_vv V + [0 | r_v | 0]
  double BF[9][12];                     // rows 3..11 of B, force columns
  double qr[kQrdStride];                // the node's compact Q / R record
  double peF[12], bF[12], RFpe[12], rF[12];
};

// first stance component and their number by mode (FLY, LF, RF, STANCE: kForceRow of project_lu_s.h)
__device__ __forceinline__ int stance_first(int mode) { return mode == 2 ? 6 : 0; }
__device__ __forceinline__ int stance_count(int mode) { return mode == 3 ? 12 : (mode == 0 ? 0 : 6); }

template <int NJ, int NBC>
__device__ __forceinline__ void project_struct_blocks(ProjectStructWorkspace<NJ>& ws, const ProjectIn& in, const ProjectOut& out, int mode, double dt,
                                                      double dt_over_mass, const double* Qc, const double* Rc, double reg, int nut) {
  using WS = ProjectStructWorkspace<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, KJ = WS::KJ, KSJ = KJ / 4, BC = NX + 1, WP = PackedLq<NJ>::WP, QP = PackedLq<NJ>::QP;
  static_assert(NJ <= 16 && KJ <= 16 && NX >= 16 && NX < 32, "joint rows fit one block row; column nx sits in block column 1");
  const int l = threadIdx.x, li = l & 15, lk = l >> 4;
  const int nsf = stance_count(mode), c0s = stance_first(mode);
  const double shift = ws.qr[0];
  // R_FF(c, c2): the 3x3 block of a contact from the node's record, dt x the constant weight across contacts
  auto rff = [&](int c, int c2) { return (c / 3 == c2 / 3) ? ws.qr[1 + 3 * c2 + c % 3] : dt * Rc[c * NU + c2]; };
  // B(rr, c) for a force component c
  auto bfc = [&](int rr, int c) { return rr < 3 ? ((c % 3 == rr) ? dt_over_mass : 0.0) : (rr < 12 ? ws.BF[rr - 3][c] : 0.0); };

  // ---- A-operands of the two products with a constant / gathered left factor (rows li, inner index = joint 4 ks + lk)
  double aR[KSJ], aB[KSJ];
#pragma unroll
  for (int ks = 0; ks < KSJ; ++ks) {
    const int j = 4 * ks + lk;
    const bool okr = li < NJ && j < NJ;
    const double rv = Rc[okr ? (12 + li) * NU + 12 + j : 0];
    aR[ks] = okr ? dt * (li == j ? rv + shift : rv) : 0.0;
    const bool dense = li >= 3 && li < 12 && j < NJ;
    const double bv = in.B[dense ? li * NU + 12 + j : 3 * NU];
    aB[ks] = dense ? bv : ((li >= 12 && li - 12 == j) ? dt : 0.0);          // rows 0..2: no joint dependence; rows 12..15: dt on the own joint
  }
  // ---- accumulator initial values (D layout: row lk + 4 r of the block, column li)
  v4d cA[NBC], cQ[2][2], cS[NBC][NBC], cR[NBC];
#pragma unroll
  for (int bj = 0; bj < NBC; ++bj) {
    const int col = 16 * bj + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = lk + 4 * r;                                   // block row 0 of W: rows 0..15
      double v;
      if (col < NX) {
        const bool dense = rr >= 3 && rr < 12;
        const double av = in.A[dense ? rr * NX + col : 3 * NX];
        v = dense ? av : (rr == col ? 1.0 : 0.0);
      } else if (col == NX) {
        v = in.b[rr] + (rr < 12 ? ws.bF[rr] : 0.0);
      } else {
        const int s = col - BC;
        v = s < nsf ? bfc(rr, c0s + s) : 0.0;
      }
      cA[bj][r] = v;
      const int j = rr;                                            // RV: row = joint lk + 4 r
      cR[bj][r] = (col == NX && j < NJ) ? in.r[12 + j] : 0.0;
    }
  }
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const int col = 16 * bj + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * bi + lk + 4 * r;
        const bool in_m = rr < NX && col < NX, in_v = rr < NX && col == NX;
        double qv = *(in_m ? Qc + rr * NX + col : (in_v ? in.q + rr : in.q));
        if (in_m) qv = dt * (rr == col ? qv + shift : qv);
        cQ[bi][bj][r] = (in_m || in_v) ? qv : 0.0;
      }
    }
  // rows of the stance components in the packed result (row nx + 1 + s): [0 | r_F + R_FF Pe_F | R_FF(stance, stance) | 0]
#pragma unroll
  for (int bi = 1; bi < NBC; ++bi)
#pragma unroll
    for (int bj = 0; bj < NBC; ++bj) {
      const int col = 16 * bj + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s = 16 * bi + lk + 4 * r - BC;                   // reduced input of this row
        double v = 0.0;
        if (s >= 0 && s < nsf) {
          const int c = c0s + s, s2 = col - BC;
          if (col == NX) v = ws.rF[c] + ws.RFpe[c];
          else if (s2 >= 0 && s2 < nsf) v = rff(c, c0s + s2);
        }
        cS[bi][bj][r] = v;
      }
    }
  lds_wave_sync();                                       // V is in LDS (written by the caller)

  // ---- RV = R_vv V + [0 | r_v | 0]  ->  LDS
#pragma unroll
  for (int bj = 0; bj < NBC; ++bj) {
    double b[KSJ];
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) b[ks] = ws.V[4 * ks + lk][16 * bj + li];
    v4d acc = cR[bj];
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[ks], b[ks], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (lk + 4 * r < KJ) ws.RV[lk + 4 * r][16 * bj + li] = acc[r];
  }
  // ---- [At | bt | Bt]: block row 0 by MFMA, the joint rows 16.. as identity / b + dt V
#pragma unroll
  for (int bj = 0; bj < NBC; ++bj) {
    double b[KSJ];
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) b[ks] = ws.V[4 * ks + lk][16 * bj + li];
    v4d acc = cA[bj];
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aB[ks], b[ks], acc, 0, 0, 0);
    const int col = 16 * bj + li;
    double* wrow = out.Wt + lk * WP + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) wrow[4 * r * WP] = acc[r];          // rows 0..15 < nx
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 + lk + 4 * r;
      if (rr < NX) {
        const double base = col < NX ? (rr == col ? 1.0 : 0.0) : (col == NX ? in.b[rr] : 0.0);
        out.Wt[rr * WP + col] = base + dt * ws.V[rr - 12][col];
      }
    }
  }
  lds_wave_sync();                                       // RV is complete
  // ---- V' RV + initial values  ->  Qt, qt (rows < nx);  Pt, rt, Rt (rows > nx)
#pragma unroll
  for (int bi = 0; bi < NBC; ++bi) {
    double a[KSJ];
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) a[ks] = ws.V[4 * ks + lk][16 * bi + li];           // V'(i, k)
#pragma unroll
    for (int bj = 0; bj < NBC; ++bj) {
      if (bi == 0 && bj >= 2) continue;                  // block row 0 is all state rows: only the columns up to nx are kept
      double b[KSJ];
#pragma unroll
      for (int ks = 0; ks < KSJ; ++ks) b[ks] = ws.RV[4 * ks + lk][16 * bj + li];
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      if (bi < 2 && bj < 2) acc = cQ[bi < 2 ? bi : 0][bj < 2 ? bj : 0];
      if (bi >= 1) acc += cS[bi >= 1 ? bi : 1][bj];       // rows of the stance components (the state rows of block row 1 get zeros)
#pragma unroll
      for (int ks = 0; ks < KSJ; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
      const int col = 16 * bj + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * bi + lk + 4 * r;
        const int ru = rr - BC;
        if (rr < NX) {
          if (bj < 2) out.Qp[rr * QP + col] = col <= NX ? acc[r] + (rr == col ? reg : 0.0) : 0.0;      // reg: settings.reg_prim (HPIPM's), 0 by default
          // (whole 128-byte segments: leaving the zero columns 24.. unwritten made the kernel SLOWER, 2.65 -> 2.87 ms at batch 4096 - partial lines)
        } else if (rr > NX && ru < NU) {
          out.Mt[ru * WP + col] = acc[r] + ((ru == col - BC && ru < nut) ? reg : 0.0);
        }
      }
    }
  }
}

template <int NJ>
__device__ __forceinline__ void project_apply_struct(ProjectStructWorkspace<NJ>& ws, const ProjectIn& in, const ProjectOut& out, int mode, double dt,
                                                     double dt_over_mass, const double* Qc, const double* Rc, double reg = 0.0) {
  using WS = ProjectStructWorkspace<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, LDV = WS::LDV, KJ = WS::KJ, BC = NX + 1;
  static_assert(NX == NU, "packed layout assumes nx == nu");
  const int l = threadIdx.x;

  if (in.kind == 1) {  // event node: identity jump map, no input, no cost (as project_mfma.h)
    constexpr int WP = PackedLq<NJ>::WP, QP = PackedLq<NJ>::QP;
    for (int idx = l; idx < NX * 32; idx += kWave) {
      const int i = idx >> 5, j = idx & 31;
      out.Wt[i * WP + j] = j < NX ? (i == j ? 1.0 : 0.0) : (j == NX ? in.b[i] : 0.0);
      out.Qp[i * QP + j] = (i == j) ? reg : 0.0;
    }
    return;
  }
  const int nut = out.nut[0];
  const int nbc = (BC + nut + 15) >> 4;                // block columns (and rows) of the packed width nx + 1 + nut

  // ---- everything this node needs beyond the operands of project_struct_blocks: joint rows of X (packed by the elimination kernel),
  //      B_F, the Q / R record, Pe_F, r_F
  {
    constexpr int WP = PackedLq<NJ>::WP, NV = NJ * WP, IT = (NV + kWave - 1) / kWave, IB = (9 * 12 + kWave - 1) / kWave;
    double vv[IT], vb[IB];
#pragma unroll
    for (int it = 0; it < IT; ++it) {                  // all loads in flight before the first LDS write
      const int idx = l + it * kWave;
      vv[it] = (idx < NV && idx % WP < 16 * (nbc < WS::NBC_MAX ? nbc : WS::NBC_MAX)) ? in.Vt[idx] : 0.0;      // the columns the elimination kernel wrote
    }
#pragma unroll
    for (int it = 0; it < IB; ++it) {
      const int idx = l + it * kWave;
      vb[it] = idx < 9 * 12 ? in.B[(3 + idx / 12) * NU + idx % 12] : 0.0;
    }
    const double qv = l < kQrdStride ? in.qrd[l] : 0.0;
    const double pev = l < 12 ? out.Pe[l] : 0.0;
    const double rv = l < 12 ? in.r[l] : 0.0;
    for (int idx = l; idx < KJ * LDV; idx += kWave) (&ws.V[0][0])[idx] = 0.0;         // padding: rows nj.., columns beyond the packed row
    lds_wave_sync();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = l + it * kWave;
      if (idx < NV && idx % WP < LDV - 2) ws.V[idx / WP][idx % WP] = vv[it];
    }
#pragma unroll
    for (int it = 0; it < IB; ++it) {
      const int idx = l + it * kWave;
      if (idx < 9 * 12) ws.BF[idx / 12][idx % 12] = vb[it];
    }
    if (l < kQrdStride) ws.qr[l] = qv;
    if (l < 12) { ws.peF[l] = pev; ws.rF[l] = rv; }
    lds_wave_sync();
    // B_F Pe_F (rows 0..11 of the b column) and R_FF Pe_F: twelve short sums each, on two groups of lanes
    if (l < 12) {
      double s = 0.0;
      if (l < 3) { for (int c = l; c < 12; c += 3) s += ws.peF[c]; s *= dt_over_mass; }
      else for (int c = 0; c < 12; ++c) s += ws.BF[l - 3][c] * ws.peF[c];
      ws.bF[l] = s;
    } else if (l >= 16 && l < 28) {
      const int c = l - 16;
      double s = 0.0;
      for (int c2 = 0; c2 < 12; ++c2) s += ((c / 3 == c2 / 3) ? ws.qr[1 + 3 * c2 + c % 3] : dt * Rc[c * NU + c2]) * ws.peF[c2];
      ws.RFpe[c] = s;
    }
  }
  lds_wave_sync();
  if (nbc <= 2) project_struct_blocks<NJ, 2>(ws, in, out, mode, dt, dt_over_mass, Qc, Rc, reg, nut);
  else project_struct_blocks<NJ, WS::NBC_MAX>(ws, in, out, mode, dt, dt_over_mass, Qc, Rc, reg, nut);
}

}  // namespace bpmpc
