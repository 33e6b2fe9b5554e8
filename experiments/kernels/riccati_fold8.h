// Riccati sweep, eight wavefronts per problem, with the CHANGE OF VARIABLES FOLDED IN (HIP only; gfx950; nx = 22).
//
// The unfused step writes the projected LQ model of every node to HBM (k_project_struct: 13.5 KB per node) and the sweep reads it back
// (22.5 KB incl. padding, plus [Px | Pe | Pu]): both kernels run near the HBM rate at every batch size, a quarter of a step.  Here the
// projected model never exists in HBM: two waves of the sweep's workgroup that used to prefetch and stage it - the loaders L4, L5 -
// COMPUTE it, one stage ahead of the chain, from what the lineariser and the structured elimination leave (rows 3..11 of A and B, b, q, r,
// the 320-byte Q / R record, the packed joint rows Vt of [Px | Pe | Pu]: 6.7 KB per node), with the structured products of
// project_struct.h (inner dimension = joint rows, force rows assembled: 24 / 42 matrix-core instructions per node), straight into the
// sweep's LDS operands.  Roles (riccati_mfma8.h for the chain, unchanged):
//     P4  w = 4   dynamics side: [A~ | b~ | B~] (9 MFMAs for block row 0, the joint rows as dt V + identity); blocks 0, 2 of Sn as before
//     P5  w = 5   cost side: RV = R_vv V + r (9), then V' RV + [Q q; stance rows] (24) -> [Q~ | q~], [P~ | r~ | R~]
//     F   w = 6   a block of SW; the fourth output block, block 1 of Sn, m
// Timeline of node j = k - 1 while the chain works on stage k (phases P1..P3 of stage k; every wave meets all four barriers):
//     P1  registers (requested a stage earlier) -> LDS: [Px | Pe | Pu] of node j complete (joint rows + generated force rows), its raw rows
//         of A, B, b, q, r, the Q / R record; requests of node j - 1
//     P2  P4: B_F Pe_F, initial values, the nine MFMAs of block row 0;  P5: R_FF Pe_F, RV
//     P3  P4: stores of W, joint rows, its Sn blocks;  P5: V' RV -> Qq, M
// Buffers as in riccati_dma8.h (whose schedule this kernel inherits): W and PW triple buffered (the outputs of stage k + 1 still read
// them in P3 of stage k), Qq and M double buffered, the gain in Yb, r~ / q~ copied out in P1.  LDS: 157 KB.
#pragma once
#include <hip/hip_runtime.h>

#include "project_struct.h"
#include "riccati_mfma8.h"

namespace bpmpc {

template <int NJ>
struct RiccatiFold8Workspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int RB = 32, RE = 16, LDN = 34;
  static constexpr int WC = NX + 1 + NU;
  static constexpr int LDW = ((WC + 15) / 16) * 16 + 2;
  static constexpr int KJ = ((NJ + 3) / 4) * 4;
  static_assert(NX + 1 <= 32 && NU <= 32 && NX % 2 == 0, "two block rows / columns; rows of nx doubles are whole 16-byte pairs");
  alignas(16) double S[RB][LDN];
  alignas(16) double Sn[RB][LDN];
  alignas(16) double Zt[RE][LDN];
  alignas(16) double Yn[RE][LDN];
  alignas(16) double Yb[RE][LDN];       // -Y of the stage eliminated last
  alignas(16) double SW[RB][LDW];
  alignas(16) double Qq[2][RB][LDN];    // [Q~ | q~]          written by P5
  alignas(16) double M[2][RE][LDW];     // [P~ | r~ | R~]     written by P5
  alignas(16) double W[3][RB][LDW];     // [A~ | b~ | B~]     written by P4
  alignas(16) double PW[3][RB][LDW];    // [Px | Pe | Pu]     staged by P4, P5
  // the projector's own data of the node being projected
  alignas(16) double RV[KJ][LDW];       // R_vv V + [0 | r_v | 0]
  alignas(16) double A9[9][NX + 2];     // rows 3..11 of A
  alignas(16) double B9[9][NX + 2];     // rows 3..11 of B (force columns 0..11, joint columns 12..)
  alignas(16) double bqr[3][NX + 2];    // b, q, r
  alignas(16) double qr[kQrdStride];    // Q / R record of the node (linearize_fast.h)
  double bF[12], RFpe[12];
  double qd[NX + 2];                    // diagonal of the state weight Q (constant)
  double r[2][RE];
  int status;
  unsigned char nut[kMaxRiccatiStages];
  unsigned char mode[kMaxRiccatiStages];
};

// What the projector waves request from HBM for one node and how it reaches LDS.  128 lanes (P4, P5), lane t owns the 16-byte pairs
// t, t + 128, .. of every stream; everything is requested in P1 of one stage and written to LDS in P1 of the next.
template <int NJ, int LDW>
struct FoldLoader {
  using PL = PackedLq<NJ>;
  static constexpr int NX = PL::NX, NU = PL::NU, WP = PL::WP, BC = NX + 1, HW = WP / 2, NLD = 2 * kWave;
  static constexpr int NPV = NJ * HW, NPF = 12 * HW, NPA = 9 * (NX / 2), NPS = 3 * (NX / 2) + kQrdStride / 2;
  static constexpr int SV = (NPV + NLD - 1) / NLD, SF = (NPF + NLD - 1) / NLD;
  static_assert(NPA <= NLD && NPS <= NLD && NX % 2 == 0 && kQrdStride % 2 == 0 && WP <= LDW, "one slot per lane for the raw rows");
  double vx[SV], vy[SV], fx[SF], fy[SF];
  double ax, ay, bx, by, sx, sy;            // a pair of A rows 3..11, of B rows 3..11, of [b | q | r | record]
  double dt;                                // of the requested node (uniform)
  int tl;
  // (the offsets are recomputed from tl where they are used - a few integer operations off the chain - instead of living in sixteen
  //  registers through the whole sweep: the kernel is at the register limit)
  static constexpr int HP = NX / 2;
  __device__ __forceinline__ void init(int tl_) { tl = tl_; }
  __device__ __forceinline__ int v_off(int e) const { const int p = tl + e * NLD; return p < NPV ? (12 + p / HW) * LDW + 2 * (p % HW) : -1; }
  __device__ __forceinline__ int f_off(int e) const { const int p = tl + e * NLD; return p < NPF ? (p / HW) * LDW + 2 * (p % HW) : -1; }
  __device__ __forceinline__ int f_col(int e) const { return 2 * ((tl + e * NLD) % HW); }
  __device__ __forceinline__ int f_row(int e) const { const int p = tl + e * NLD; return p < NPF ? p / HW : 0; }
  // requests of node j of the problem (mode: 0..3 or kModeEvent)
  __device__ __forceinline__ void prefetch(const RiccatiFastIO& io, size_t j, int nt, int mode) {
    const int c0s = mode == 2 ? 6 : 0, nsf = mode == 3 ? 12 : ((mode == 0 || mode >= kModeEvent) ? 0 : 6);
    const double2* gV = reinterpret_cast<const double2*>(io.Vt + j * (NJ * WP));
    const double2* zero2 = reinterpret_cast<const double2*>(io.zero_one + 2);
    const double* gPe = io.base.Pe + j * NU;
    const int cend = 16 * ((BC + nt + 15) >> 4);       // the elimination kernel wrote the columns below this one
#pragma unroll
    for (int e = 0; e < SV; ++e) { const int p = tl + e * NLD; const double2 v = *((p < NPV && 2 * (p % HW) < cend) ? gV + p : zero2); vx[e] = v.x; vy[e] = v.y; }
#pragma unroll
    for (int e = 0; e < SF; ++e) {
      const int s = f_row(e) - c0s, fc = f_col(e);
      const int ucol = (s >= 0 && s < nsf) ? BC + s : -1;
      fx[e] = *(fc == NX ? gPe + f_row(e) : io.zero_one + (fc == ucol ? 1 : 0));
      fy[e] = fc + 1 == ucol ? 1.0 : 0.0;
    }
    const int a_src = tl < NPA ? (3 + tl / HP) * NX + 2 * (tl % HP) : 3 * NX;
    { const double2 v = *reinterpret_cast<const double2*>(io.lqA + j * (NX * NX) + a_src); ax = v.x; ay = v.y; }
    { const double2 v = *reinterpret_cast<const double2*>(io.lqB + j * (NX * NU) + a_src); bx = v.x; by = v.y; }
    {
      const int which = tl < 3 * HP ? tl / HP : 3;
      const int src = tl < 3 * HP ? 2 * (tl % HP) : (tl < NPS ? 2 * (tl - 3 * HP) : 0);
      const double* base = which == 0 ? io.lqb + j * NX : (which == 1 ? io.lqq + j * NX : (which == 2 ? io.lqr + j * NU : io.qrd + j * kQrdStride));
      const double2 v = *reinterpret_cast<const double2*>(base + src);
      sx = v.x; sy = v.y;
    }
    dt = io.gdt[j];
  }
  // registers -> LDS: PW of the node (complete), its raw rows
  template <class WS>
  __device__ __forceinline__ void stage(WS& ws, double (*PW)[LDW]) const {
    double* PWf = &PW[0][0];
#pragma unroll
    for (int e = 0; e < SV; ++e)
      if ((e + 1) * NLD <= NPV || v_off(e) >= 0) { double2 v; v.x = vx[e]; v.y = vy[e]; *reinterpret_cast<double2*>(PWf + v_off(e)) = v; }
#pragma unroll
    for (int e = 0; e < SF; ++e)
      if ((e + 1) * NLD <= NPF || f_off(e) >= 0) { double2 v; v.x = fx[e]; v.y = fy[e]; *reinterpret_cast<double2*>(PWf + f_off(e)) = v; }
    if (tl < NPA) {
      const int ao = (tl / HP) * (NX + 2) + 2 * (tl % HP);
      double2 v; v.x = ax; v.y = ay; *reinterpret_cast<double2*>(&ws.A9[0][0] + ao) = v;
      v.x = bx; v.y = by; *reinterpret_cast<double2*>(&ws.B9[0][0] + ao) = v;
    }
    if (tl < NPS) {
      const bool rec = tl >= 3 * HP;
      const int so = rec ? 2 * (tl - 3 * HP) : (tl / HP) * (NX + 2) + 2 * (tl % HP);
      double2 v; v.x = sx; v.y = sy; *reinterpret_cast<double2*>((rec ? &ws.qr[0] : &ws.bqr[0][0]) + so) = v;
    }
  }
};

// ---- the two halves of the structured change of variables (project_struct.h), LDS to LDS, written for a LONE wave: it issues an
// instruction every ~8 cycles and nothing hides an LDS round trip, so every block reads all it needs first, the initial values of an
// accumulator are decided per BLOCK (which of the four kinds of entries it can hold) instead of per element, and the weights are
// restricted to what the shipped configurations have (checked when the solver is created, otherwise the unfused path runs): Q diagonal,
// the force block of R block diagonal per contact, no force / joint-velocity cross terms.
//   PW: [Px | Pe | Pu] of the node (rows 12.. = V), nt reduced inputs, contact mode, dt.  NBC = block columns of nx + 1 + nt.
template <int NJ, int NBC, class WS>
__device__ __forceinline__ void fold_dynamics_mfma(WS& ws, const double (*PW)[WS::LDW], double (*W)[WS::LDW], int mode, double dt, double dt_over_mass) {
  v4d acc[NBC];
  constexpr int NX = WS::NX, KSJ = WS::KJ / 4, BC = NX + 1;
  const int l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const int nsf = stance_count(mode), c0s = stance_first(mode);
  if (l < 12) {                                        // B_F Pe_F, rows 0..11 of the b column: all operands first
    double pe[12], bb[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) { pe[c] = PW[c][NX]; bb[c] = l >= 3 ? ws.B9[l - 3][c] : ((c % 3 == l) ? dt_over_mass : 0.0); }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int c = 0; c < 12; c += 3) { s0 += bb[c] * pe[c]; s1 += bb[c + 1] * pe[c + 1]; s2 += bb[c + 2] * pe[c + 2]; }
    ws.bF[l] = s0 + s1 + s2;
  }
  // A-operand: rows 3..11 of B_v (rows 0..2 do not depend on the joint velocities; the joint rows 12.. are copied, not multiplied)
  double aB[KSJ], bop[NBC][KSJ];
#pragma unroll
  for (int ks = 0; ks < KSJ; ++ks) {
    const int j = 4 * ks + lk;
    const bool dense = li >= 3 && li < 12 && j < NJ;
    const double bv = ws.B9[dense ? li - 3 : 0][12 + (dense ? j : 0)];
    aB[ks] = dense ? bv : 0.0;
  }
#pragma unroll
  for (int bj = 0; bj < NBC; ++bj)
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) bop[bj][ks] = PW[12 + 4 * ks + lk][16 * bj + li];
  lds_wave_sync();                                     // bF
  // initial values of rows 0..11 (D layout: row lk + 4 r): block column 0 holds state columns only, block column 1 the state columns
  // 16.., the b column and the first reduced inputs, block column 2 reduced inputs only
#pragma unroll
  for (int bj = 0; bj < NBC; ++bj) {
    const int col = 16 * bj + li;
    const int s = col - BC;
    const bool is_x = col < NX, is_b = col == NX, is_s = s >= 0 && s < nsf;
    const int c = is_s ? c0s + s : 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {                      // rows lk + 4 r < 12
      const int rr = lk + 4 * r;
      const bool dense = rr >= 3;
      const double av = ws.A9[dense ? rr - 3 : 0][is_x ? col : 0];
      const double bv = ws.B9[dense ? rr - 3 : 0][c];
      const double xb = ws.bqr[0][rr] + ws.bF[rr];
      const double fv = dense ? bv : ((c % 3 == rr) ? dt_over_mass : 0.0);
      acc[bj][r] = is_x ? (dense ? av : (rr == col ? 1.0 : 0.0)) : (is_b ? xb : (is_s ? fv : 0.0));
    }
    acc[bj][3] = 0.0;
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB[ks], bop[bj][ks], acc[bj], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 3; ++r) W[lk + 4 * r][col] = acc[bj][r];       // rows 0..11 (the buffer has been free since the stage before)
  }
  (void)dt;
}
template <int NJ, int NBC, class WS>
__device__ __forceinline__ void fold_dynamics_joint_rows(WS& ws, const double (*PW)[WS::LDW], double (*W)[WS::LDW], double dt, bool clear_wide) {
  constexpr int NX = WS::NX;
  const int l = threadIdx.x & 63;
  // joint rows 12..: identity / b + dt V, one 16-byte pair per lane and pass (nj x 24 pairs)
  constexpr int HW = 24, NP = NJ * HW, IT = (NP + kWave - 1) / kWave;
  double2 v[IT];
  double bb[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int p = l + it * kWave, row = 12 + (p < NP ? p / HW : 0), c2 = 2 * (p % HW);
    v[it] = *reinterpret_cast<const double2*>(&PW[row][c2]);
    bb[it] = ws.bqr[0][row];
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int p = l + it * kWave, row = 12 + p / HW, c2 = 2 * (p % HW);
    if (p < NP && (c2 < 16 * NBC || clear_wide)) {
      double2 o;
      o.x = dt * v[it].x + (c2 == row ? 1.0 : (c2 == NX ? bb[it] : 0.0));
      o.y = dt * v[it].y + (c2 + 1 == row ? 1.0 : 0.0);
      *reinterpret_cast<double2*>(&W[row][c2]) = o;
    }
  }
  if (NBC < 3 && clear_wide) {                         // the buffer last held a node with three block columns: rows 0..11 of the third one
    for (int idx = l; idx < 12 * 16; idx += kWave) W[idx >> 4][32 + (idx & 15)] = 0.0;
  }
}
// event node: W = [I | b | 0]
template <class WS>
__device__ __forceinline__ void fold_dynamics_event(WS& ws, double (*W)[WS::LDW], bool clear_wide) {
  constexpr int NX = WS::NX;
  const int l = threadIdx.x & 63;
  const int wc = clear_wide ? 48 : 32;
  for (int idx = l; idx < NX * wc; idx += kWave) {
    const int i = idx / wc, c = idx % wc;
    W[i][c] = c < NX ? (i == c ? 1.0 : 0.0) : (c == NX ? ws.bqr[0][i] : 0.0);
  }
}

// cost side, part one: RV = R_vv V + [0 | r_v | 0] -> ws.RV, and R_FF Pe_F.  rvv[ks]: R(12 + li, 12 + 4 ks + lk) of the model (constant).
template <int NJ, int NBC, class WS>
__device__ __forceinline__ void fold_cost_rv(WS& ws, const double (*PW)[WS::LDW], double dt, const double (&rvv)[WS::KJ / 4]) {
  constexpr int NX = WS::NX, KJ = WS::KJ, KSJ = KJ / 4;
  const int l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const double shift = ws.qr[0];
  double bop[NBC][KSJ], rv[3];
#pragma unroll
  for (int bj = 0; bj < NBC; ++bj)
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) bop[bj][ks] = PW[12 + 4 * ks + lk][16 * bj + li];
#pragma unroll
  for (int r = 0; r < 3; ++r) rv[r] = ws.bqr[2][12 + lk + 4 * r];
  if (l < 12) {                                        // R_FF Pe_F: the force block is block diagonal per contact (3 terms)
    const int c = l, c0 = 3 * (c / 3);
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < 3; ++m) s += ws.qr[1 + 3 * (c0 + m) + c % 3] * PW[c0 + m][NX];
    ws.RFpe[c] = s;
  }
  double aR[KSJ];
#pragma unroll
  for (int ks = 0; ks < KSJ; ++ks) {
    const int j = 4 * ks + lk;
    aR[ks] = (li < NJ && j < NJ) ? dt * (li == j ? rvv[ks] + shift : rvv[ks]) : 0.0;
  }
#pragma unroll
  for (int bj = 0; bj < NBC; ++bj) {
    const int col = 16 * bj + li;
    v4d acc;
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[r] = (col == NX && lk + 4 * r < NJ) ? rv[r] : 0.0;
    acc[3] = 0.0;
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[ks], bop[bj][ks], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 3; ++r) ws.RV[lk + 4 * r][col] = acc[r];       // rows 0..11 = KJ
  }
  static_assert(KJ == 12, "three D-layout rows per lane cover the joint rows");
}
// cost side, part two: V' RV + [Q | q | 0 ; stance rows] -> Qq (rows < nx), M (rows > nx).  Q is diagonal (qd in LDS).
template <int NJ, int NBC, class WS>
__device__ __forceinline__ void fold_cost_blocks(WS& ws, const double (*PW)[WS::LDW], double (*Qq)[WS::LDN], double (*M)[WS::LDW], int mode, int nut,
                                                 double dt, double reg) {
  constexpr int NX = WS::NX, KSJ = WS::KJ / 4, BC = NX + 1, RE = WS::RE;
  const int l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const int nsf = stance_count(mode), c0s = stance_first(mode);
  const double shift = ws.qr[0];
  double bop[NBC][KSJ];                                // RV, all block columns (reused by every block row)
#pragma unroll
  for (int b = 0; b < NBC; ++b)
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) bop[b][ks] = ws.RV[4 * ks + lk][16 * b + li];
#pragma unroll
  for (int bi = 0; bi < NBC; ++bi) {
    // per block row: V' operand; the lane's diagonal weight and q (state rows), r~ of the stance rows
    double aop[KSJ], dq[4], qv[4], sr[4];
#pragma unroll
    for (int ks = 0; ks < KSJ; ++ks) aop[ks] = PW[12 + 4 * ks + lk][16 * bi + li];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * bi + lk + 4 * r, rc = rr < NX ? rr : 0;
      const int s = rr - BC;
      const bool st = s >= 0 && s < nsf;
      const int c = st ? c0s + s : 0;
      const double qdv = ws.qd[rc], qq = ws.bqr[1][rc], rf = ws.bqr[2][c], rp = ws.RFpe[c];
      dq[r] = rr < NX ? dt * (qdv + shift) : 0.0;
      qv[r] = rr < NX ? qq : 0.0;
      sr[r] = st ? rf + rp : 0.0;
    }
#pragma unroll
    for (int bj = 0; bj < NBC; ++bj) {
      if (bi == 0 && bj >= 2) continue;
      const int col = 16 * bj + li;
      const int s2 = col - BC;
      const bool st2 = s2 >= 0 && s2 < nsf;
      v4d acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * bi + lk + 4 * r;
        double v = 0.0;
        if (bi < 2 && bj < 2) v = (rr == col) ? dq[r] : ((col == NX) ? qv[r] : 0.0);      // state rows (zeros beyond nx)
        if (bi >= 1 && bj >= 1) {                                                            // stance rows: [. | r~ | R_FF block of the contact]
          const int s = rr - BC;
          const bool hit = s >= 0 && s < nsf && st2 && (c0s + s2) / 3 == (c0s + s) / 3;
          const double rv = ws.qr[hit ? 1 + 3 * (c0s + s2) + (c0s + s) % 3 : 0];
          if (rr > NX) v = (col == NX) ? sr[r] : (hit ? rv : 0.0);
        }
        acc[r] = v;
      }
#pragma unroll
      for (int ks = 0; ks < KSJ; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ks], bop[bj][ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * bi + lk + 4 * r;
        const int ru = rr - BC;
        if (rr < NX) {
          if (bj < 2) Qq[rr][col] = col <= NX ? acc[r] + (rr == col ? reg : 0.0) : 0.0;
        } else if (rr > NX && ru < RE) {
          M[ru][col] = acc[r] + ((ru == col - BC && ru < nut) ? reg : 0.0);
        }
      }
    }
  }
}
template <class WS>
__device__ __forceinline__ void fold_cost_event(double (*Qq)[WS::LDN], double reg) {
  constexpr int NX = WS::NX;
  const int l = threadIdx.x & 63;
  for (int idx = l; idx < NX * 32; idx += kWave) Qq[idx >> 5][idx & 31] = ((idx >> 5) == (idx & 31)) ? reg : 0.0;
}

template <int NJ>
__device__ __forceinline__ void riccati_fold8(RiccatiFold8Workspace<NJ>& ws, const RiccatiFastIO& io) {
  using WS = RiccatiFold8Workspace<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, NT = kRiccati8Threads, LDN = WS::LDN, LDW = WS::LDW, RE = WS::RE;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  constexpr int KS = (NX + 3) / 4, KSJ = WS::KJ / 4;
  constexpr int BC = NX + 1;
  static_assert(NX == NU, "packed layouts assume nx == nu");
  static_assert(NX + 1 + RE <= kWave, "one lane per column of [H | G g]");
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int li = l & 15, lk = l >> 4;
  const int N = io.base.N;
  const bool role_c = w < 4, role_p = w == 4 || w == 5, role_f = w == 6, role_e = w == 7;
  const double inv_mass = 1.0 / io.model->robot_mass;

  const int k_top = (io.k_hi < N ? io.k_hi : N) - 1;
  const bool resumed = io.k_hi < N;
  {
    double* z = &ws.S[0][0];
    constexpr int total = (int)(offsetof(WS, status) / sizeof(double));
    for (int idx = tid; idx < total; idx += NT) z[idx] = 0.0;     // every matrix and its padding
  }
  __syncthreads();
  if (tid == 0) ws.status = resumed ? (int)io.carry[NXX + NX] : 0;
  if (!resumed && io.reg != 0.0 && tid < NX) ws.S[tid][tid] = io.reg;
  if (resumed) {
    for (int idx = tid; idx < NXX; idx += NT) ws.S[idx / NX][idx % NX] = io.carry[idx];
    if (tid < NX) ws.S[tid][NX] = io.carry[NXX + tid];
  }
  int too_wide = 0;
  for (int idx = tid; idx < N && idx < kMaxRiccatiStages; idx += NT) {
    const int n = io.base.nut[idx];
    ws.nut[idx] = (unsigned char)n;
    ws.mode[idx] = (unsigned char)(n > 0 ? (io.mode[idx] & 3) : kModeEvent);
    too_wide |= n > RE ? 1 : 0;
  }
  if (__syncthreads_or(too_wide)) {         // more reduced inputs than this variant holds: fail loudly (status 2 in bpmpc_stats)
    if (tid == 0) {
      if (io.k_lo > 0) io.carry[NXX + NX] = 1.0;
      else { io.base.summary[0] = 0.0; io.base.summary[1] = 0.0; io.base.summary[2] = 0.0; io.base.summary[3] = 1.0; }
    }
    if (io.k_lo == 0 && io.with_ls && tid < kWave) linesearch_begin_wave<NJ>(&ws.S[0][0], io.ls, tid);
    return;
  }

  // ---- projector state (waves 4, 5)
  FoldLoader<NJ, LDW> fl;
  fl.init(role_p ? tid - 4 * kWave : 0);
  double rvv[KSJ];                          // the joint block of the input weight R at this lane's A-operand positions (constant)
#pragma unroll
  for (int ks = 0; ks < KSJ; ++ks) { const int j = 4 * ks + lk; rvv[ks] = (li < NJ && j < NJ) ? io.model->R[(12 + li) * NU + 12 + j] : 0.0; }
  if (tid < NX) ws.qd[tid] = io.model->Q[tid * NX + tid];       // (published by the barriers of the prologue)
  int wide_mask = 0;                        // W buffers (k mod 3) that last held three block columns
  double p_dt = 0.0;                        // dt of the node whose data is in LDS
  auto stage_node = [&](int j) { fl.stage(ws, ws.PW[j % 3]); p_dt = fl.dt; };
  auto request_node = [&](int j, int nt, int mode) { fl.prefetch(io, (size_t)j, nt, mode); };
  // part one (P2 of the chain's stage): everything up to the stores
  auto project_part1 = [&](int j) {
    const int nt = ws.nut[j], mode = ws.mode[j];
    if (mode >= kModeEvent) return;
    const double (*PW)[LDW] = ws.PW[j % 3];
    const int nbc = (BC + nt + 15) >> 4;
    if (w == 4) {
      if (nbc <= 2) fold_dynamics_mfma<NJ, 2>(ws, PW, ws.W[j % 3], mode, p_dt, p_dt * inv_mass);
      else fold_dynamics_mfma<NJ, 3>(ws, PW, ws.W[j % 3], mode, p_dt, p_dt * inv_mass);
    } else {
      if (nbc <= 2) fold_cost_rv<NJ, 2>(ws, PW, p_dt, rvv);
      else fold_cost_rv<NJ, 3>(ws, PW, p_dt, rvv);
    }
  };
  auto project_part2 = [&](int j) {
    const int nt = ws.nut[j], mode = ws.mode[j];
    const double (*PW)[LDW] = ws.PW[j % 3];
    const int nbc = (BC + nt + 15) >> 4;
    const int bit = 1 << (j % 3);
    if (w == 4) {
      const bool clear = (wide_mask & bit) != 0;
      if (mode >= kModeEvent) fold_dynamics_event(ws, ws.W[j % 3], clear);
      else if (nbc <= 2) fold_dynamics_joint_rows<NJ, 2>(ws, PW, ws.W[j % 3], p_dt, clear);
      else fold_dynamics_joint_rows<NJ, 3>(ws, PW, ws.W[j % 3], p_dt, false);
      wide_mask = (mode < kModeEvent && nbc > 2) ? (wide_mask | bit) : (wide_mask & ~bit);
    } else {
      lds_wave_sync();                       // RV (written by this wave in part one)
      if (mode >= kModeEvent) fold_cost_event<WS>(ws.Qq[j & 1], io.reg);
      else if (nbc <= 2) fold_cost_blocks<NJ, 2>(ws, PW, ws.Qq[j & 1], ws.M[j & 1], mode, nt, p_dt, io.reg);
      else fold_cost_blocks<NJ, 3>(ws, PW, ws.Qq[j & 1], ws.M[j & 1], mode, nt, p_dt, io.reg);
    }
  };
  // prologue: node k_top completely, the requests of node k_top - 1
  if (role_p && k_top >= io.k_lo) { const int n0 = io.base.nut[k_top]; request_node(k_top, n0, n0 > 0 ? (io.mode[k_top] & 3) : kModeEvent); }
  __syncthreads();
  if (role_p && k_top >= io.k_lo) {
    stage_node(k_top);
    if (k_top > io.k_lo) request_node(k_top - 1, ws.nut[k_top - 1], ws.mode[k_top - 1]);
  }
  __syncthreads();
  if (role_p && k_top >= io.k_lo) project_part1(k_top);
  __syncthreads();
  if (role_p && k_top >= io.k_lo) project_part2(k_top);
  __syncthreads();

  // Outputs of a stage that are not on the chain (block bw of each): [Acl | bcl] = [A | b] - B Y, [K | kff] = [Px | Pe] - Pu Y
  auto finish_outputs = [&](int k, int nt, int bw) {
    double (*const W)[LDW] = ws.W[k % 3];
    double (*const PW)[LDW] = ws.PW[k % 3];
    const int ksn = (nt + 3) >> 2;
    const int r0 = 16 * (bw >> 1), c0 = 16 * (bw & 1);
    const int row = r0 + li;
    v4d acl = blk_load<LDW, 32, 0>(&W[0][0], r0, c0, l);
    v4d kf = blk_load<LDW, 32, 0>(&PW[0][0], r0, c0, l);
    double ab[4], ap[4], yb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = 4 * ks + lk;
      yb[ks] = ws.Yb[kk][c0 + li];                               // -Y (E stores the gain negated); rows >= nt are zero
      ab[ks] = W[row][BC + kk];                                  // B(i, kk); rows >= nx of W and PW are zero
      ap[ks] = PW[row][BC + kk];                                 // Pu(i, kk)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < ksn) {                                            // wave-uniform
        acl = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[ks], yb[ks], acl, 0, 0, 0);
        kf = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[ks], yb[ks], kf, 0, 0, 0);
      }
    }
    double* Acl = io.Acl + (size_t)k * NXX;
    double* Kf = io.Kfull + (size_t)k * NXU;
    const int col = c0 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = r0 + lk + 4 * r;
      if (rr < NX) {
        if (col < NX) { Acl[rr * NX + col] = acl[r]; Kf[rr * NX + col] = kf[r]; }
        else if (col == NX) { io.bcl[(size_t)k * NX + rr] = acl[r]; io.kff[(size_t)k * NU + rr] = kf[r]; }
      }
    }
  };
  // m = q~ - Y' r~, m0 = -r~' H^-1 g of a finished stage (wave 6; q~ of that stage in `qv`, lane l = component l)
  auto finish_m = [&](int k, double qv) {
    if (l <= NX) {
      double yv[RE], rv[RE];
#pragma unroll
      for (int i = 0; i < RE; ++i) { yv[i] = ws.Yb[i][l]; rv[i] = ws.r[k & 1][i]; }
      double m0 = l < NX ? qv : 0.0, m1 = 0.0;
#pragma unroll
      for (int i = 0; i < RE; i += 2) { m0 += yv[i] * rv[i]; m1 += yv[i + 1] * rv[i + 1]; }     // yv: -Y
      if (l < NX) io.mvec[(size_t)k * NX + l] = m0 + m1; else io.mscal[k] = m0 + m1;
    }
  };

  int pend_k = -1, pend_nt = 0;
  double q_pend = 0.0, q_cur = 0.0;      // wave 6: q~ of the pending / the current stage (lane l = component l)
#ifdef BPMPC_FOLD_PROFILE
  long long fp_acc[4] = {0, 0, 0, 0}, fp_t0 = 0;       // own work of this wave in P1, P2, P3 and the whole loop
  const long long fp_start = clock64();
#define FP_BEGIN() (fp_t0 = clock64())
#define FP_END(i) (fp_acc[i] += clock64() - fp_t0)
#else
#define FP_BEGIN() ((void)0)
#define FP_END(i) ((void)0)
#endif
  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = ws.nut[k];
    const int b2 = k & 1, b3 = k % 3;
    double (*const W)[LDW] = ws.W[b3];
    double (*const Qq)[LDN] = ws.Qq[b2];
    double (*const M)[LDW] = ws.M[b2];
    const int ksn = (nt + 3) >> 2;
    const int nbc = (BC + nt + 15) >> 4;
    const bool ahead = role_p && k > io.k_lo;        // the projector works on node k - 1
    auto sn_block = [&](int sid) {     // [Sn | sn] = [Q | q] + A' SW(:, 0..nx), block sid of four
      const int r0 = 16 * (sid >> 1), c0 = 16 * (sid & 1);
      v4d acc = blk_load<LDN, 32, 0>(&Qq[0][0], r0, c0, l);
      const int acol = r0 + li < NX ? r0 + li : LDW - 1;
      double a[KS], b[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 4 * ks + lk;
        a[ks] = W[kk][acol];
        b[ks] = ws.SW[kk][c0 + li];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
      blk_store<LDN, 32>(&ws.Sn[0][0], r0, c0, l, acc);
    };
    lds_barrier();                     // B0: the operands of this stage are complete (projected during stage k + 1)
    // ---- P1: SW = sym(S) W on C0..C3, F, E;  P4, P5: node k - 1 registers -> LDS, requests of node k - 2;  F keeps r~, q~ of this stage
    FP_BEGIN();
    if (ahead) {
      stage_node(k - 1);
      if (k - 1 > io.k_lo) request_node(k - 2, ws.nut[k - 2], ws.mode[k - 2]);
    }
    if (role_p) FP_END(0);
    if (role_f) {
      if (l < RE) ws.r[b2][l] = l < nt ? M[l][NX] : 0.0;
      q_cur = l < NX ? Qq[l][NX] : 0.0;
    }
    if (!role_p) {
      const int id = w < 4 ? w : w - 2;
      if (id < 2 * nbc) {
        const int bi = id >= nbc ? 1 : 0;
        const int r0 = 16 * bi, c0 = 16 * (id - bi * nbc);
        const int row = r0 + li;
        const double half = row < NX ? 0.5 : 0.0;
        double a[KS], b[KS], sv[4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int kk = 4 * ks + lk;
          a[ks] = half * (ws.S[row][kk] + ws.S[kk][row]);
          b[ks] = W[kk][c0 + li];
        }
        const double smask = (c0 + li == NX) ? 1.0 : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = smask * ws.S[r0 + lk + 4 * r][NX];
        __builtin_amdgcn_sched_barrier(0);
        v4d acc = {sv[0], sv[1], sv[2], sv[3]};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
        blk_store<LDW, 32>(&ws.SW[0][0], r0, c0, l, acc);
      }
    }
    lds_barrier();                     // B1
    // ---- P2: [G | g | H] on C0..C2; C3: block 3 of Sn;  P4, P5: part one of node k - 1
    if (role_c) {
      if (w < nbc) {
        const int c0 = 16 * w;
        v4d acc = blk_load<LDW, 32, 0>(&M[0][0], 0, c0, l);
        double a[KS], b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int kk = 4 * ks + lk;
          a[ks] = W[kk][BC + li];
          b[ks] = ws.SW[kk][c0 + li];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
        blk_store<LDW, 32>(&M[0][0], 0, c0, l, acc);
      } else if (w == 3) {
        sn_block(3);
      }
    }
    FP_BEGIN();
    if (ahead) project_part1(k - 1);
    if (role_p) FP_END(1);
    lds_barrier();                     // B2
    // ---- P3 (E): forward elimination -> Z, Yn;  B3;  back substitution -> Yb
    //      C0..C2, F: outputs of stage k + 1;  P4: stores of node k - 1, blocks 0, 2 of Sn;  P5: cost blocks of node k - 1;  F: block 1 of Sn, m
    if (role_e) {
      const int rpr = 16 - nt;
      const bool rows_layout = BPMPC_RICCATI_GJ_DPP && 4 * rpr >= NX + 1;
      const int c16 = l & 15;
      const int rid = rows_layout ? (l >> 4) * rpr + (c16 - nt) : l - nt;
      const bool is_h = rows_layout ? c16 < nt : l < nt;
      const bool rhs = !is_h && rid < NX + 1;
      const bool used = is_h || rhs;
      const int col = is_h ? BC + (rows_layout ? c16 : l) : (rhs ? rid : 0);
      bool ok;
      if (rhs) {
        for (int i = nt; i < 4 * ksn; ++i) { ws.Zt[i][col] = 0.0; ws.Yn[i][col] = 0.0; }
      }
      static_assert(NX + 2 + 3 < LDN - 1 && 4 * KS <= NX + 2, "spare columns of Z / Yn");
      const int ecol = rhs ? col : NX + 2 + (l & 3);
      auto emit = [&](int p, double z, double y) { ws.Zt[p][ecol] = z; ws.Yn[p][ecol] = y; };
#define BP_GJ_CASE(ROWS, FWD, BWD)                                                            \
      {                                                                                       \
        double v[ROWS];                                                                       \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) { const double t = M[i][col]; v[i] = (used && i < nt) ? t : 0.0; } \
        ok = FWD<ROWS>(v, nt, emit);                                                          \
        if (l == 0 && !ok) ws.status = 1;                                                     \
        lds_barrier();                 /* B3 */                                               \
        BWD<ROWS>(v, nt);                                                                     \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) if (rhs && i < 4 * ksn) ws.Yb[i][col] = -v[i];                       \
        if (ROWS < 4 * ksn && rhs) for (int i = ROWS; i < 4 * ksn; ++i) ws.Yb[i][col] = 0.0;                                  \
      }
      if (rows_layout) {
        if (nt <= 8) BP_GJ_CASE(8, forward_eliminate_rows, back_substitute_rows)
        else if (nt == 9) BP_GJ_CASE(9, forward_eliminate_rows, back_substitute_rows)
        else BP_GJ_CASE(10, forward_eliminate_rows, back_substitute_rows)
      } else {
        if (nt <= 12) BP_GJ_CASE(12, forward_eliminate_wave, back_substitute_wave)
        else BP_GJ_CASE(RE, forward_eliminate_wave, back_substitute_wave)
      }
#undef BP_GJ_CASE
    } else {
      FP_BEGIN();
      if (ahead) project_part2(k - 1);
      if (role_p) FP_END(2);
      if (w == 4) { sn_block(0); sn_block(2); }
      if (role_f) sn_block(1);
      if ((w < 3 || role_f) && pend_k >= 0) finish_outputs(pend_k, pend_nt, w < 3 ? w : 3);
      if (role_f && pend_k >= 0) finish_m(pend_k, q_pend);
      lds_barrier();                   // B3
      if (role_c) {
        const int r0 = 16 * (w >> 1), c0 = 16 * (w & 1);
        v4d acc = blk_load<LDN, 32, 0>(&ws.Sn[0][0], r0, c0, l);
        const int gcol = r0 + li < NX ? r0 + li : LDN - 1;
        double ag[4], yb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kk = 4 * ks + lk;
          ag[ks] = -ws.Zt[kk][gcol];
          yb[ks] = ws.Yn[kk][c0 + li];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks < ksn) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[ks], yb[ks], acc, 0, 0, 0);
        blk_store<LDN, 32>(&ws.S[0][0], r0, c0, l, acc);
      }
    }
    pend_k = k; pend_nt = nt; q_pend = q_cur;
  }
#ifdef BPMPC_FOLD_PROFILE
  fp_acc[3] = clock64() - fp_start;
  if (io.prof && l == 0 && (w == 4 || w == 5)) { for (int i = 0; i < 3; ++i) io.prof[(w - 4) * 3 + i] = (double)fp_acc[i]; }
  if (io.prof && tid == 0) { io.prof[6] = (double)fp_acc[3]; io.prof[7] = (double)(k_top - io.k_lo + 1); }
#endif
  __syncthreads();
  if ((w < 3 || role_f) && pend_k >= 0) finish_outputs(pend_k, pend_nt, w < 3 ? w : 3);
  if (role_f && pend_k >= 0) finish_m(pend_k, q_pend);
  __syncthreads();
  if (io.k_lo > 0) {
    for (int idx = tid; idx < NXX; idx += NT) io.carry[idx] = ws.S[idx / NX][idx % NX];
    if (tid < NX) io.carry[NXX + tid] = ws.S[tid][NX];
    if (tid == 0) io.carry[NXX + NX] = (double)ws.status;
    return;
  }
  {
    const int st = ws.status;
    __syncthreads();                                   // the workspace is dead from here on: it holds the state history
    constexpr int kHistCap = ((int)(offsetof(WS, status) / sizeof(double)) - kStepNormsScratch * NT / kWave) / NX - 8;
    static_assert(kHistCap >= 64, "roll-out history");
    riccati_rollout_deep<NJ, NT>(reinterpret_cast<double*>(&ws), kHistCap, st, io);
  }
}

}  // namespace bpmpc
