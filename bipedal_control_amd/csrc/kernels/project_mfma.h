// Constraint elimination, part two, on the FP64 matrix cores (HIP only): the change of input variables of one node
//     du = Px dx + Pu dut + Pe     ([OCS2-upstream] multiple_shooting::projectTranscription / changeOfInputVariables;
// reference body with a general cost cross term: project_node.h).  Px, Pu, Pe and nut come from project_lu4.h.
//
// One wavefront per node.  With the packed operand X = [Px | Pe | Pu] (nu x (nx + 1 + nut), zero padded) the whole
// transformation is three products of 16x16 blocks accumulated with v_mfma_f64_16x16x4_f64:
//     RX          = R X + [0 | r | 0]                 -> [R Px | r + R Pe | R Pu]
//     [At|bt|Bt]  = [A | b | 0] + B X
//     X' RX + [Q | q | 0 ; 0]                         -> Qt (rows < nx, cols < nx), qt (col nx), Pt, rt, Rt (rows > nx)
// The operands R and B are used exactly once each, so they go from HBM straight into the A-operand registers (lane =
// (row, k) of the 16x4 operand); the accumulator blocks are initialised from A, b, Q, q in the D layout and leave for HBM
// in the same layout (16 consecutive doubles per row).  Only X and RX go through LDS.  The cost cross term P of the LQ
// model is structurally zero for this problem and is not read (project_node.h handles a general P).
// The results leave in the packed layout of project_node.h (PackedLq): Wt = [At | bt | Bt], Qp = [Qt | qt], Mt = [Pt | rt | Rt], the
// column layout of the products themselves.  Everything beyond the reduced input dimension nut must read as zero for the Riccati
// sweep: inside the block columns that are computed the products are exact zeros there (X is zero padded); block columns
// >= nbc and rows >= nut of Mt are not written at all - the sweep's staging masks them by nut.
#pragma once
#include <hip/hip_runtime.h>

#include "project_node.h"
#include "riccati_mfma.h"   // v4d, lds_wave_sync

namespace bpmpc {

template <int NJ, bool PK = false>
struct ProjectMfmaWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int KR = ((NX + 3) / 4) * 4;                  // rows used as the k index (nx rounded up to the k-step)
  // Three block columns of the packed operand: nx + 1 + nut <= 48, i.e. up to 25 (nx = 22: all) / 23 (nx = 24) reduced inputs.  The sweeps
  // hold at most 16 at nx = 24 and report a numerical failure beyond that, so a wider node is cut off here rather than given a fourth block
  // column - which cost the nx = 24 kernel its third wave per SIMD (16.1 KB of LDS and 171 registers instead of 13.1 KB and <= 168).
  static constexpr int NBC_MAX = 3;
  static constexpr int WC = (NX + 1 + NU < 16 * NBC_MAX) ? NX + 1 + NU : 16 * NBC_MAX;     // columns of [Px | Pe | Pu] that are kept
  static constexpr int LDW = 16 * NBC_MAX + 2;
  // PK (after the structured elimination): the force rows 0..11 of X hold Pe in column nx and a single 1 per stance component and are
  // generated in registers where they are used; LDS holds the joint rows only - 8.3 KB per wave instead of 13 KB: 16 waves per CU, not 12
  static constexpr int X0 = PK ? 12 : 0;                         // first row of X that lives in LDS
  alignas(16) double X[KR - X0][LDW];   // [Px | Pe | Pu], zero padded
  alignas(16) double RX[KR][16 + 2];    // one block column of R X + [0 | r | 0] at a time
};

// The three products for a compile-time number of block columns NBC (packed width nx + 1 + nut <= 16 NBC).  Everything
// that is read from HBM (operands, accumulator initial values) is loaded before the first output store: vmcnt retires in
// order, a load issued behind a store would wait for that store to reach memory.
// WJ = false: the joint rows 12.. of Wt = [At | bt | Bt] are neither computed nor written - they are [I | b | 0] + dt x (joint rows of
// [Px | Pe | Pu]) = Vt, which the wave-per-problem sweeps load anyway and complete themselves (riccati_wave2.h, JW): at the batch sizes where
// this kernel streams (4.7 TB/s) that is 3.8 KB per node it does not write and the sweep does not read, and the whole second block row of
// the first product (rows 16..) is not issued.
template <int NJ, int NBC, bool PK, bool WJ, class IssueX, class WriteX>
__device__ __forceinline__ void project_apply_blocks(ProjectMfmaWorkspace<NJ, PK>& ws, const ProjectIn& in, const ProjectOut& out, double dt,
                                                     double dt_over_mass, const double* Qc, const double* Rc, double reg, int nut,
                                                     IssueX&& issue_x, WriteX&& write_x, const double& pev) {
  using WS = ProjectMfmaWorkspace<NJ, PK>;
  constexpr int NX = WS::NX, NU = WS::NU, KR = WS::KR, KS = KR / 4, BC = NX + 1, WP = PackedLq<NJ>::WP, QP = PackedLq<NJ>::QP;
  const int l = threadIdx.x, li = l & 15, lk = l >> 4;
  // The memory latency of a node is paid in as few round trips as the registers allow:
  //   (1) the loads of X (the caller's issue_x) together with the RAW operands of the first product, [A | b | 0] + B X;
  //   (2) X goes to LDS (write_x), the RAW operands of the other two products are issued behind it and arrive under the first product.
  // RAW: the arithmetic on a loaded value (shift, dt, identity rows, masks) comes after the loads of its group - a select or a product
  // on a loaded value makes the wave wait for that load, and interleaved with the loads it serialised the latency several times over.
  issue_x();
  const double shift = in.qrd[0];
  const double shift_r = in.qrd[in.r_shift_at];        // (the same entry except for the DDP solver's ILQR lineariser)
  // A-operands from HBM: rows 16 bi + li of R and of B, k = 4 ks + lk (out-of-range lanes read element 0 and are masked)
  double aR[2][KS], aB[2][KS];
  // accumulator initial values in the D layout: [A | b | 0] (2 x NBC blocks), [Q | q | 0] (block row 0 and, for rows < nx, 1),
  // r for the b column of R X
  v4d cA[2][NBC], cQ[2][NBC], cR[2];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int row = 16 * bi + li, kk = 4 * ks + lk;
      // Only the rows 3..11 of the discretised centroidal dynamics are dense (linearize_fast.h writes the others as
      //   A: identity rows;   B rows 0..2: dt/m on the matching component of the four contact forces;   B rows 12..: dt on the
      // joint velocity), so those are generated here instead of being read (a fifth of this kernel's HBM reads).
      const bool dense = row < NU && kk < NU && row >= 3 && row < 12;
      aB[bi][ks] = in.B[dense ? row * NU + kk : 3 * NU];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * bi + lk + 4 * r;
      const bool rin = rr < NX;
#pragma unroll
      for (int bj = 0; bj < NBC; ++bj) {
        const int col = 16 * bj + li;
        const bool in_m = rin && col < NX, in_v = rin && col == NX;
        const bool a_dense = in_m && rr >= 3 && rr < 12;
        cA[bi][bj][r] = *(a_dense ? in.A + rr * NX + col : (in_v ? in.b + rr : in.A + 3 * NX));
      }
    }
  }
  write_x();
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int row = 16 * bi + li, kk = 4 * ks + lk;
      const bool ok = row < NU && kk < NU;
      // R and Q of the node are dt x (constant weight) except on the diagonal (Hessian shift of the relaxed barriers) and in the
      // four 3x3 force blocks of R (cone Hessians), linearize_fast.h: those come from the node's compact record (320 B), the
      // rest is regenerated from the model constants (cache resident) with the lineariser's own expressions - 7 KB less HBM
      // read per node than fetching Q and R
      const bool r_block = ok && row < 12 && kk < 12 && row / 3 == kk / 3;
      aR[bi][ks] = *(r_block ? in.qrd + 1 + 3 * kk + row % 3 : Rc + (ok ? row * NU + kk : 0));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * bi + lk + 4 * r;
      const bool rin = rr < NX;
      cR[bi][r] = in.r[rin ? rr : 0];
#pragma unroll
      for (int bj = 0; bj < NBC; ++bj) {
        const int col = 16 * bj + li;
        const bool in_m = rin && col < NX, in_v = rin && col == NX;
        cQ[bi][bj][r] = *(in_m ? Qc + rr * NX + col : (in_v ? in.q + rr : in.q));
      }
    }
  }
  // ---- the arithmetic on the operands of the first product
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int row = 16 * bi + li, kk = 4 * ks + lk;
      const bool ok = row < NU && kk < NU;
      const bool dense = ok && row >= 3 && row < 12;
      const double synth = row < 3 ? ((kk < 12 && kk % 3 == row) ? dt_over_mass : 0.0) : (kk == row ? dt : 0.0);
      aB[bi][ks] = dense ? aB[bi][ks] : (ok ? synth : 0.0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * bi + lk + 4 * r;
      const bool rin = rr < NX;
#pragma unroll
      for (int bj = 0; bj < NBC; ++bj) {
        const int col = 16 * bj + li;
        const bool in_m = rin && col < NX, in_v = rin && col == NX;
        const bool a_dense = in_m && rr >= 3 && rr < 12;
        double av = cA[bi][bj][r];
        if (in_m && !a_dense) av = (rr == col) ? 1.0 : 0.0;          // identity rows of A
        cA[bi][bj][r] = (in_m || in_v) ? av : 0.0;
      }
    }
  }
  static_assert(NX >= 16 && NX < 32, "column nx sits in block column 1");
  lds_wave_sync();                                     // X is in LDS (write_x)
  // PK: element (4 ks + lk, col) of the force rows, ks < 3: Pe in column nx, a 1 in column nx + 1 + (row - first stance component) of a stance
  // component, zero elsewhere - in particular in the whole block column 0, whose k-steps 0..2 are skipped (exact zeros either way)
  constexpr int KF = PK ? 3 : 0;                        // k-steps that cover the force rows
  double pe_k[3] = {0.0, 0.0, 0.0};
  int one_k[3] = {-1, -1, -1};
  if constexpr (PK) {
    const int mode = in.mode, c0s = mode == 2 ? 6 : 0, nsf = mode == 3 ? 12 : (mode == 0 ? 0 : 6);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      pe_k[ks] = __shfl(pev, 4 * ks + lk);
      const int sidx = 4 * ks + lk - c0s;
      one_k[ks] = (sidx >= 0 && sidx < nsf) ? BC + sidx : -1;
    }
  }
  // operand element X(4 ks + lk, 16 blk + li)
  auto x_elem = [&](int ks, int blk) -> double {
    if (ks < KF) {
      const int col = 16 * blk + li;
      return col == NX ? pe_k[ks] : (col == one_k[ks] ? 1.0 : 0.0);
    }
    return ws.X[4 * ks + lk - WS::X0][16 * blk + li];
  };

  // Every HBM load of this node has been issued by now, so results may leave as soon as they exist.
  // ---- [At | bt | Bt] = [A | b | 0] + B X, stored at once (frees the B operands and these accumulators)
#pragma unroll
  for (int bi = 0; bi < (WJ ? 2 : 1); ++bi)
#pragma unroll
    for (int bj = 0; bj < NBC; ++bj) {
      double b[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) b[ks] = x_elem(ks, bj);
      v4d acc = cA[bi][bj];
#pragma unroll
      for (int ks = (bj == 0 ? KF : 0); ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aB[bi][ks], b[ks], acc, 0, 0, 0);
      // packed layout (PackedLq): the block is 16 aligned row segments of Wt; columns beyond nx + 1 + nut hold exact zeros (B x 0)
      double* wrow = out.Wt + (16 * bi + lk) * WP + 16 * bj + li;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * bi + lk + 4 * r < (WJ ? NX : 12)) wrow[4 * r * WP] = acc[r];
    }
  // ---- the arithmetic on the operands of the other two products (their loads travelled under the first one)
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int row = 16 * bi + li, kk = 4 * ks + lk;
      const bool ok = row < NU && kk < NU;
      const bool r_block = ok && row < 12 && kk < 12 && row / 3 == kk / 3;
      const double rv = aR[bi][ks];
      aR[bi][ks] = ok ? (r_block ? rv : dt * (row == kk ? rv + shift_r : rv)) : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * bi + lk + 4 * r;
      const bool rin = rr < NX;
      cR[bi][r] = (li == NX - 16 && rin) ? cR[bi][r] : 0.0;      // column nx lives in block column 1 (nx in 16..31)
#pragma unroll
      for (int bj = 0; bj < NBC; ++bj) {
        const int col = 16 * bj + li;
        const bool in_m = rin && col < NX, in_v = rin && col == NX;
        double qv = cQ[bi][bj][r];
        if (in_m) qv = dt * (rr == col ? qv + shift : qv);
        cQ[bi][bj][r] = (in_m || in_v) ? qv : 0.0;
      }
    }
  }
  // ---- per block column bj:  RX(:, bj) = R X(:, bj) + [0 | r | 0]  ->  LDS,  then  X' RX(:, bj) + [Q | q | 0 ; 0](:, bj)
  //      -> Qt, qt (rows < nx);  Pt, rt, Rt (rows > nx)
#pragma unroll
  for (int bj = 0; bj < NBC; ++bj) {
    const int col = 16 * bj + li;
    {
      double b[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) b[ks] = x_elem(ks, bj);
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        if (bj == 1) acc = cR[bi];
#pragma unroll
        for (int ks = (bj == 0 ? KF : 0); ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[bi][ks], b[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 16 * bi + lk + 4 * r;
          if (rr < KR) ws.RX[rr][li] = acc[r];
        }
      }
    }
    lds_wave_sync();
    double b[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) b[ks] = ws.RX[4 * ks + lk][li];
#pragma unroll
    for (int bi = 0; bi < NBC; ++bi) {
      // rows of the packed result: < nx: [Qt | qt | Pt'], == nx: the Pe row (unused), > nx: [Pt | rt | Rt].  Block row 0 is all
      // top rows, and of those only the columns up to nx are kept (Pt' is Pt again): nothing to do for block columns >= 2
      if (bi == 0 && bj >= 2) continue;
      double a[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[ks] = x_elem(ks, bi);         // X'(i, k)
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      if (bi < 2) acc = cQ[bi < 2 ? bi : 0][bj];         // rows >= nx of cQ are zero
#pragma unroll
      for (int ks = (bi == 0 ? KF : 0); ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * bi + lk + 4 * r;
        const int ru = rr - BC;
        if (rr < NX) {
          if (bj < 2) out.Qp[rr * QP + col] = col <= NX ? acc[r] + (rr == col ? reg : 0.0) : 0.0;      // reg: settings.reg_prim (HPIPM's), 0 by default
          // (whole 128-byte segments: leaving the zero columns 24.. unwritten made the kernel SLOWER, 2.65 -> 2.87 ms at batch 4096 - partial lines)      // reg: settings.reg_prim (HPIPM's), 0 by default
        } else if (rr > NX && ru < NU) {
          out.Mt[ru * WP + col] = acc[r] + ((ru == col - BC && ru < nut) ? reg : 0.0);
        }
      }
    }
    lds_wave_sync();                                     // the block column buffer is rewritten by the next bj
  }
}

// PK: [Px | Pe | Pu] arrives as the packed joint rows of the structured elimination (in.Vt; the force rows are generated from in.mode)
// instead of as Px, Pu, Pe - a compile-time choice: both paths in one kernel cost nx = 24 its third wave per SIMD.
template <int NJ, bool PK = false, bool WJ = true>
__device__ __forceinline__ void project_apply_mfma(ProjectMfmaWorkspace<NJ, PK>& ws, const ProjectIn& in, const ProjectOut& out, double dt,
                                                   double dt_over_mass, const double* Qc, const double* Rc, double reg = 0.0) {
  using WS = ProjectMfmaWorkspace<NJ, PK>;
  constexpr int NX = WS::NX, NU = WS::NU, LDW = WS::LDW, KR = WS::KR, BC = NX + 1, WC = WS::WC;
  static_assert(NX == NU, "packed layout assumes nx == nu");
  const int l = threadIdx.x;

  if (in.kind == 1) {  // event node: identity jump map, no input, no cost (what the lineariser writes for it in the materialised mode;
                       // generated here so that the fused mode need not write it); Px, Pu, Pe, nut = 0 were written by the LU kernel.
                       // Packed layout: the first two block columns of Wt = [I | b | 0] and Qp = [reg I | 0]; the reader masks the rest
    constexpr int WP = PackedLq<NJ>::WP, QP = PackedLq<NJ>::QP;
    for (int idx = l; idx < NX * 32; idx += kWave) {
      const int i = idx >> 5, j = idx & 31;
      if (WJ || i < 12) out.Wt[i * WP + j] = j < NX ? (i == j ? 1.0 : 0.0) : (j == NX ? in.b[i] : 0.0);
      out.Qp[i * QP + j] = (i == j) ? reg : 0.0;
    }
    if (!PK && out.Vt)
      for (int idx = l; idx < NJ * 48; idx += kWave) out.Vt[(idx / 48) * WP + idx % 48] = 0.0;
    return;
  }
  const int nut = out.nut[0];
  const int nbc = (BC + nut + 15) >> 4;                // block columns (and rows) of the packed width nx + 1 + nut

  // ---- X to LDS (coalesced reads), padding zeroed; the workgroup is this one wave.  Split in two: the loads are issued together with the
  //      first operand loads of the products (project_apply_blocks), the LDS writes follow once those are in flight.
  if constexpr (PK) {
    // after the structured elimination: the joint rows come packed (columns below the first unwritten block column), the
    // force rows are generated - zero except Pe_c in column nx and a single 1 for a stance component (project_lu_s.h)
    constexpr int WP = PackedLq<NJ>::WP, NV = NJ * 48, IT = (NV + kWave - 1) / kWave;
    double vv[IT];
    double pev = 0.0;
    auto issue_x = [&]() {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = l + it * kWave;
        // (the elimination kernel wrote the columns < 16 nbc; beyond them a zero is loaded: an unconditional load with a selected ADDRESS,
        //  where a conditional load is a branch and a wait for everything in flight)
        vv[it] = *((idx < NV && idx % 48 < 16 * nbc) ? in.Vt + (idx / 48) * WP + idx % 48 : in.zero);
      }
      pev = *(l < 12 ? out.Pe + l : in.zero);
    };
    auto write_x = [&]() {                               // LDS rows: joint rows 12.. of X (the force rows are generated, project_apply_blocks)
      for (int idx = l; idx < NJ * (LDW - 48); idx += kWave) ws.X[idx / (LDW - 48)][48 + idx % (LDW - 48)] = 0.0;   // padding columns of the joint rows
      for (int idx = l; idx < (KR - NU) * LDW; idx += kWave) (&ws.X[NU - 12][0])[idx] = 0.0;            // rows nu..
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = l + it * kWave;
        if (idx < NV) ws.X[idx / 48][idx % 48] = vv[it];
      }
    };
    if (nbc <= 2) project_apply_blocks<NJ, 2, true, WJ>(ws, in, out, dt, dt_over_mass, Qc, Rc, reg, nut, issue_x, write_x, pev);
    else project_apply_blocks<NJ, WS::NBC_MAX, true, WJ>(ws, in, out, dt, dt_over_mass, Qc, Rc, reg, nut, issue_x, write_x, pev);
  } else {
    auto issue_x = [&]() {                             // (test path, FullPivLU-format inputs: staged in one go)
      constexpr int IT = (NU * NX + kWave - 1) / kWave;
      double vx[IT], vu[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) {                  // all loads in flight before the first LDS write
        const int idx = l + it * kWave;
        const bool ok = idx < NU * NX;
        vx[it] = ok ? out.Px[idx] : 0.0;
        vu[it] = (ok && idx % NX < nut) ? out.Pu[idx] : 0.0;     // columns >= nut of Pu are zero: not read
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = l + it * kWave;
        if (idx < NU * NX) { ws.X[idx / NX][idx % NX] = vx[it]; if (BC + idx % NX < WC) ws.X[idx / NX][BC + idx % NX] = vu[it]; }
      }
      if (l < NU) ws.X[l][NX] = out.Pe[l];
      for (int idx = l; idx < NU * (LDW - WC); idx += kWave) ws.X[idx / (LDW - WC)][WC + idx % (LDW - WC)] = 0.0;   // columns beyond [Px Pe Pu]
      for (int idx = l; idx < (KR - NU) * LDW; idx += kWave) (&ws.X[NU][0])[idx] = 0.0;                 // rows nu..
      if (out.Vt) {                                      // joint rows of X in the packed layout the sweep's loaders read (columns < 16 nbc)
        constexpr int WP = PackedLq<NJ>::WP;
        lds_wave_sync();
        for (int idx = l; idx < NJ * WP; idx += kWave) {
          const int r = idx / WP, c = idx % WP;
          if (c < 48) out.Vt[idx] = ws.X[12 + r][c];       // complete rows of three block columns (X is zero padded)
        }
      }
    };
    auto write_x = []() {};
    const double no_pe = 0.0;
    if (nbc <= 2) project_apply_blocks<NJ, 2, false, WJ>(ws, in, out, dt, dt_over_mass, Qc, Rc, reg, nut, issue_x, write_x, no_pe);
    else project_apply_blocks<NJ, WS::NBC_MAX, false, WJ>(ws, in, out, dt, dt_over_mass, Qc, Rc, reg, nut, issue_x, write_x, no_pe);
  }
}

}  // namespace bpmpc
