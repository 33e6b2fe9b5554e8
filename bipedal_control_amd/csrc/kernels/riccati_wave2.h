// The wave-per-problem Riccati sweep (riccati_wave.h) rearranged so that TWO waves fit a SIMD: <= 256 registers and <= 20 KB of LDS per
// problem, eight problems per CU.  Same mathematics, same layouts, same results to rounding; what changes is the order of a stage, chosen
// for the number of values that are alive at once (riccati_wave.h keeps ~210 doubles in registers, this file ~105):
//   * the products are taken one BLOCK COLUMN of the packed width at a time: SW(:, bj) = S W(:, bj) (two blocks), at once
//     M(:, bj) += B' SW(:, bj) and Sn(:, bj) += A' SW(:, bj), then SW(:, bj) is dropped - 8 instead of 24 doubles of S W;
//   * the operand of a column is loaded into the registers it is used from and nothing is double buffered: by the time a group of loads
//     for the next stage is issued, its registers are dead.  Three groups, each issued as late as its first use allows and as early as the
//     registers allow: (1) column 0 of W, the B~ columns, the first blocks of Mt and Qp - after the update of S; (2) the other columns -
//     after the closed-loop dynamics have left for LDS; (3) the row-major copies of B~, Pu and [Px | Pe] of the SAME stage - around the
//     elimination.  The lead is a third of a stage instead of a whole one; the second wave of the SIMD covers the rest;
//   * the registers of W double as the initial values of [Acl | bcl] (the accumulator layout of a block of W is its operand layout);
//   * the updates are taken one after the other - S (then symmetrised), [Acl | bcl] -> LDS, [K | kff] -> LDS - instead of interleaved;
//   * the output tiles lie over the tiles of the elimination (dead by then), 17 KB of LDS in all.
// With two instruction streams per SIMD the stage is bound by the pipe the FP64 matrix instructions share with the FP64 VALU
// (~95 x 64 + ~500 x 4 cycles) instead of by the issue rate of a lone wave (~2 300 x 5).
#pragma once
#include <hip/hip_runtime.h>

#include "riccati_wave.h"

namespace bpmpc {

template <int NJ>
struct RiccatiWave2Workspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int LDM = PackedLq<NJ>::WP + 2;
  static constexpr int kTile = 16 * LDM + 2 * 16 * 34;      // [G g H] -> Y (16 x LDM) | Z (16 x 34) | Yn (16 x 34); the output tiles lie over it
  static_assert(2 * NX * NX <= kTile && (16 * LDM) % 2 == 0, "Acl and K fit the tiles of the elimination");
  alignas(16) double T[kTile];
  double ob[NX], ok[NU], om[NX + 2];
  double rv[16], qv[32];
  double bq[12];                 // b of the joint rows of the current stage (JW off)
  unsigned char nut[kMaxRiccatiStages], mode[kMaxRiccatiStages];
};

template <int NJ, bool JW = true>
__device__ __forceinline__ void riccati_wave2(RiccatiWave2Workspace<NJ>& ws, const RiccatiFastIO& io) {
  using WS = RiccatiWave2Workspace<NJ>;
  using PL = PackedLq<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, WP = PL::WP, QP = PL::QP, BC = NX + 1, NXX = NX * NX, NXU = NX * NU, LDM = WS::LDM;
  constexpr int KS = (NX + 3) / 4, NB = 3, XR = NX - 16;
  static_assert(KS == 6 && NX + 1 <= 32 && NX + 1 + 16 <= 48 && WP >= 48 && NX == NU, "two block rows, three block columns");
  int l = threadIdx.x & 63, li = l & 15, lk = l >> 4;      // (not const: re-derived behind an opaque statement at the top of every stage, see there)
  const int N = io.base.N;
  const int k_top = (io.k_hi < N ? io.k_hi : N) - 1;
  const bool resumed = io.k_hi < N;
  const double* const zero = io.zero_one + 2;
  double (*const Mx)[LDM] = reinterpret_cast<double (*)[LDM]>(ws.T);
  double (*const Zt)[34] = reinterpret_cast<double (*)[34]>(ws.T + 16 * LDM);
  double (*const Yn)[34] = reinterpret_cast<double (*)[34]>(ws.T + 16 * LDM + 16 * 34);
  double* const oA = ws.T;
  double* const oK = ws.T + NXX;

  {
    int tw = 0;
    for (int idx = l; idx < N && idx < kMaxRiccatiStages; idx += kWave) {
      const int n = io.base.nut[idx];
      tw |= n > 16 ? 1 : 0;
      ws.nut[idx] = (unsigned char)n;
      ws.mode[idx] = (unsigned char)(n > 0 ? (io.mode[idx] & 3) : kModeEvent);
    }
    lds_wave_sync();
    if (__any(tw)) {
      if (l == 0) io.carry[NXX + NX] = 1.0;
      return;
    }
  }
  // ---- [S | s]: block (bi, bj), register r <-> row 16 bi + lk + 4 r, column 16 bj + li
  v4d S[2][2];
  int status = 0;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * bi + lk + 4 * r, col = 16 * bj + li;
        double v;
        if (resumed) {
          const bool in = row < NX && col <= NX;
          v = *(in ? (col < NX ? io.carry + row * NX + col : io.carry + NXX + row) : zero);
        } else {
          v = (row == col && row < NX) ? io.reg : 0.0;
        }
        S[bi][bj][r] = v;
      }
  if (resumed) status = (int)io.carry[NXX + NX];
  if (k_top < io.k_lo) {
    if (l == 0) io.carry[NXX + NX] = (double)status;
    return;
  }

  // ---- buffer loads masked by their offsets (riccati_wave.h)
  constexpr unsigned kOut = 0x80000000u;
  auto rsrc = [](const double* p, int doubles) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, doubles * 8, 0x00020000); };
  auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
  };
  constexpr unsigned RS = 8u * 4 * WP;
  constexpr int PKS = (2 * RS + 8 * 48 < 4096) ? 3 : 2;
  constexpr int NG = (KS + PKS - 1) / PKS;
  constexpr int KL = (4 * KS > NX) ? KS - 1 : KS;
  unsigned gW[NG], gWl;
#pragma unroll
  for (int g = 0; g < NG; ++g) gW[g] = 8u * (unsigned)((4 * PKS * g + lk) * WP + li);
  gWl = (4 * (KS - 1) + lk < NX) ? 8u * (unsigned)((4 * (KS - 1) + lk) * WP + li) : kOut;
  auto offW = [&](int ks) { return ks < KL ? gW[ks / PKS] + RS * (unsigned)(ks % PKS) : gWl; };
  bool lx = li < XR, lxe = li <= XR;
  unsigned oBt[2], oPu[2];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
    const int row = 16 * bi + li, j = row - 12;
    // (B~ row-major feeds rows 3..11 of Acl only: operand rows >= 12 - the joint rows, which Wt does not hold when JW is off - read as zero)
    oBt[bi] = row < (JW ? NX : 12) ? 8u * (unsigned)(row * WP + BC + lk) : kOut;
    oPu[bi] = (j >= 0 && row < NX) ? 8u * (unsigned)(j * WP + BC + lk) : kOut;
  }
  const unsigned gQ = 8u * (unsigned)(lk * QP + li);                   // Qp[lk + 4 r][li]: rows 1024 bytes apart
  constexpr int RQL = (NX - 16) / 4;
  const unsigned gQl = (16 + 4 * RQL + lk < NX) ? gQ + 1024u * (4 + RQL) : kOut;
  const unsigned gM = 8u * (unsigned)(lk * WP + li);
  const unsigned gV = 8u * (unsigned)(lk * WP + li);
  const unsigned gVl = (8 + lk < NJ) ? gV + 2 * RS : kOut;
  const unsigned gPe = li == XR ? 8u * (unsigned)lk : kOut;
  // JW off: the joint rows 12.. of W = [A~ | b~ | B~] are not in HBM (k_project_fast<.., false, ..> does not write them); they are
  // [I | b | 0] + dt [Px | Pe | Pu](joint rows) and the joint rows of [Px | Pe | Pu] are Vt, which this sweep loads anyway:
  // the k-steps 3.. of every column of W come from Vt (same row stride, same columns) and are completed at the top of the stage
  auto offJ = [&](int ks) { return ks == 3 ? gV : (ks == 4 ? gV + RS : gVl); };
  double bjl = 0.0;                            // b[12 + l] of the stage being loaded (lanes 0..11)

  // ---- operand registers (no double buffers)
  double cW[NB][KS];                          // column bj of W: W[4 ks + lk][16 bj + li]
  double cWT[KS];                             // column 1 restricted to the state columns (A-operand of the second block row of A' SW)
  double cB[KS];                              // W[4 ks + lk][BC + li], li < nt
  // JW off: the k-steps 3.. (joint rows) of the columns live in arrays of their own - loaded from Vt, completed at the top of the stage
  // (kept apart from cW / cB: with the in-place completion the compiler left parts of cW in scratch memory)
  double jW[NB][KS - 3], jB[KS - 3];
#define BP_W(bj, ks) ((JW || (ks) < 3) ? cW[bj][(ks) < KS ? (ks) : 0] : jW[bj][(ks) >= 3 ? (ks) - 3 : 0])
#define BP_B(ks) ((JW || (ks) < 3) ? cB[(ks) < KS ? (ks) : 0] : jB[(ks) >= 3 ? (ks) - 3 : 0])
  v4d cM[NB];                                 // Mt, rows < nt
  v4d cQ00, cQ01, cQ11;                       // Qp blocks (0, 0), (0, 1), (1, 1)
  double cBt[2][4], cPu[2][4];
  v4d cPI[2][2];

  auto load_g1 = [&](int k, int nt) {         // column 0 and what its products need
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
    const __amdgpu_buffer_rsrc_t rm = rsrc(io.Mt + (size_t)k * PL::M_SIZE, PL::M_SIZE);
    const __amdgpu_buffer_rsrc_t rq = rsrc(io.Qp + (size_t)k * PL::Q_SIZE, PL::Q_SIZE);
    const __amdgpu_buffer_rsrc_t rv = rsrc(io.Vt + (size_t)k * (NJ * WP), NJ * WP);
    const bool in = li < nt;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (JW || ks < 3) { cW[0][ks] = bload(rw, offW(ks)); cB[ks] = bload(rw, in ? offW(ks) + 8u * BC : kOut); }
      else { jW[0][ks - 3] = bload(rv, offJ(ks)); jB[ks - 3] = bload(rv, in ? offJ(ks) + 8u * BC : kOut); }
    }
    if constexpr (!JW) {                      // b of the joint rows: one value per lane, through LDS to the lanes of column nx at the top of the stage
      const __amdgpu_buffer_rsrc_t rb = rsrc(io.lqb + (size_t)k * NX, NX);
      bjl = bload(rb, l < 12 ? 8u * (unsigned)(12 + l) : kOut);      // (beyond nx: zero)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cM[0][r] = bload(rm, (lk + 4 * r < nt) ? gM + RS * (unsigned)r : kOut);
      cQ00[r] = bload(rq, gQ + 1024u * r);
    }
  };
  auto load_g2 = [&](int k, int nt) {         // the other columns
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
    const __amdgpu_buffer_rsrc_t rm = rsrc(io.Mt + (size_t)k * PL::M_SIZE, PL::M_SIZE);
    const __amdgpu_buffer_rsrc_t rq = rsrc(io.Qp + (size_t)k * PL::Q_SIZE, PL::Q_SIZE);
    const __amdgpu_buffer_rsrc_t rv = rsrc(io.Vt + (size_t)k * (NJ * WP), NJ * WP);
    const int nbc = (BC + nt + 15) >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (JW || ks < 3) {
        cW[1][ks] = bload(rw, offW(ks) + 128u);
        if constexpr (JW) cWT[ks] = bload(rw, lx ? offW(ks) + 128u : kOut);       // (JW off: derived from cW[1] where it is used - twelve registers)
        if constexpr (JW) { if (nbc > 2) cW[2][ks] = bload(rw, offW(ks) + 256u); }
        else cW[2][ks] = bload(rw, nbc > 2 ? offW(ks) + 256u : kOut);
      } else {
        jW[1][ks - 3] = bload(rv, offJ(ks) + 128u);
        jW[2][ks - 3] = bload(rv, nbc > 2 ? offJ(ks) + 256u : kOut);        // (unconditional: a conditionally defined value that is completed every stage ends up in scratch)
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned o = (lk + 4 * r < nt) ? gM + RS * (unsigned)r : kOut;
      cM[1][r] = bload(rm, o + 128u);
      if (nbc > 2) cM[2][r] = bload(rm, o + 256u);
      cQ01[r] = bload(rq, lxe ? gQ + 1024u * r + 128u : kOut);
      if (r > RQL || (r == RQL && 16 + 4 * RQL >= NX)) cQ11[r] = 0.0;
      else if (r == RQL) cQ11[r] = bload(rq, lxe ? gQl + 128u : kOut);
      else cQ11[r] = bload(rq, lxe ? gQ + 1024u * (4 + r) + 128u : kOut);
    }
  };
  auto load_bt = [&](int k, int nt) {         // B~ row-major in the A-operand
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bool in = 4 * ks + lk < nt;
      cBt[0][ks] = bload(rw, in ? oBt[0] + 32u * ks : kOut);        // (rows 16.. of B~ row-major are not needed: Acl rows 3..11 only)
    }
  };
  auto load_pu = [&](int k, int nt) {         // Pu row-major (joint rows), [Px | Pe]
    const __amdgpu_buffer_rsrc_t rv = rsrc(io.Vt + (size_t)k * (NJ * WP), NJ * WP);
    const __amdgpu_buffer_rsrc_t rp = rsrc(io.base.Pe + (size_t)k * NU, NU);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bool in = 4 * ks + lk < nt;
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) cPu[bi][ks] = bload(rv, in ? oPu[bi] + 32u * ks : kOut);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { cPI[0][0][r] = 0.0; cPI[0][1][r] = bload(rp, gPe + 32u * r); }
    cPI[0][0][3] = bload(rv, gV); cPI[0][1][3] = bload(rv, lxe ? gV + 128u : kOut);
    cPI[1][0][0] = bload(rv, gV + RS); cPI[1][1][0] = bload(rv, lxe ? gV + RS + 128u : kOut);
    cPI[1][0][1] = bload(rv, gVl); cPI[1][1][1] = bload(rv, lxe ? gVl + 128u : kOut);
    cPI[1][0][2] = 0.0; cPI[1][0][3] = 0.0; cPI[1][1][2] = 0.0; cPI[1][1][3] = 0.0;
  };

  auto flush = [&](int hk) {
    double2* A2 = reinterpret_cast<double2*>(io.Acl + (size_t)hk * NXX);
    double2* K2 = reinterpret_cast<double2*>(io.Kfull + (size_t)hk * NXU);
    const double2* a2 = reinterpret_cast<const double2*>(oA);
    const double2* k2 = reinterpret_cast<const double2*>(oK);
    constexpr int NIT = (NXX / 2 + kWave - 1) / kWave;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = l + it * kWave;
      if (it + 1 < NIT || idx < NXX / 2) K2[idx] = k2[idx];
      if (idx >= 3 * NX / 2 && idx < 12 * NX / 2) A2[idx] = a2[idx];       // rows 3..11 of Acl
    }
    if (l < NX) {
      if (l >= 3 && l < 12) io.bcl[(size_t)hk * NX + l] = ws.ob[l];
      io.kff[(size_t)hk * NU + l] = ws.ok[l];
      io.mvec[(size_t)hk * NX + l] = ws.om[l];
    }
    if (l == NX) io.mscal[hk] = ws.om[NX];
  };

  auto stage_nt = [&](int k) { return __builtin_amdgcn_readfirstlane((int)ws.nut[k >= io.k_lo ? k : io.k_lo]); };
  auto stage_mode = [&](int k) { return __builtin_amdgcn_readfirstlane((int)ws.mode[k >= io.k_lo ? k : io.k_lo]); };
  int nt_c = stage_nt(k_top), mode_c = stage_mode(k_top);
  int nt_n = stage_nt(k_top - 1), mode_n = stage_mode(k_top - 1);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { cW[2][ks] = 0.0; if (ks >= 3) jW[2][ks - 3] = 0.0; }
  cM[2] = v4d{0.0, 0.0, 0.0, 0.0};
  load_g1(k_top, nt_c);
  load_g2(k_top, nt_c);

#ifdef BPMPC_RICCATI_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define RW2PROF(slot) do { const long long tn_ = clock64(); tacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define RW2PROF(slot) ((void)0)
#endif
  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = nt_c;
    const int nbc = (BC + nt + 15) >> 4;
    const int ksn = (nt + 3) >> 2;
    const bool more = k > io.k_lo;
    const int nt_nn = stage_nt(k - 2), mode_nn = stage_mode(k - 2);
    // The lane indices are made opaque once per stage.  Everything derived from them inside the stage - LDS addresses, masked load
    // offsets, 64-bit store addresses - is then recomputed per stage (one VALU instruction each) instead of being hoisted out of the loop
    // as an invariant: the compiler hoisted ~45 such registers and, at 256 registers, spilled them to scratch (a reload waits for every
    // load in flight).
    asm volatile("" : "+v"(l));
    li = l & 15; lk = l >> 4; lx = li < XR; lxe = li <= XR;
    if constexpr (!JW) {              // the joint rows of W from what was loaded of Vt: [I | b | 0] + dt Vt
      const double dtk = io.gdt[k];
      if (l < 12) ws.bq[l] = bjl;
      lds_wave_sync();
#pragma unroll
      for (int ks = 3; ks < KS; ++ks) {
        const int row = 4 * ks + lk;
        // (identity and b only in rows of the state: see riccati_wave.h)
        jW[0][ks - 3] = dtk * jW[0][ks - 3] + ((row < NX && li == row) ? 1.0 : 0.0);
        const double bcol = (row < NX && li == XR) ? ws.bq[4 * (ks - 3) + lk] : 0.0;
        jW[1][ks - 3] = (dtk * jW[1][ks - 3] + ((row < NX && 16 + li == row) ? 1.0 : 0.0)) + bcol;     // state columns 16..; the b column carries dt Pe + b
        jW[2][ks - 3] = dtk * jW[2][ks - 3];
        jB[ks - 3] = dtk * jB[ks - 3];
      }
    }
    RW2PROF(0);
    // ---- the products of the stage, one block column at a time
    double Sa1[KS];                                                     // S(4 ks + lk, 16 + li), state columns only
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Sa1[ks] = lx ? S[ks >> 2][1][ks & 3] : 0.0;
    v4d m[NB], sn00, sn01, sn11;
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) {
      if (bj == 2 && nbc <= 2) { m[2] = cM[2]; continue; }
      v4d sw0 = {0.0, 0.0, 0.0, 0.0}, sw1 = {0.0, 0.0, 0.0, 0.0};
      if (bj == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { sw0[r] = li == XR ? S[0][1][r] : 0.0; sw1[r] = li == XR ? S[1][1][r] : 0.0; }
        if (li == XR) {                                               // r~ and q~ as loaded (m = q~ - Y' r~ below)
#pragma unroll
          for (int r = 0; r < 4; ++r) { ws.rv[lk + 4 * r] = cM[1][r]; ws.qv[lk + 4 * r] = cQ01[r]; ws.qv[16 + lk + 4 * r] = cQ11[r]; }
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sw0 = __builtin_amdgcn_mfma_f64_16x16x4f64(S[ks >> 2][0][ks & 3], BP_W(bj, ks), sw0, 0, 0, 0);
        sw1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa1[ks], BP_W(bj, ks), sw1, 0, 0, 0);
      }
      v4d acc = cM[bj];
      if (nt > 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(BP_B(ks), ks < 4 ? sw0[ks & 3] : sw1[ks & 3], acc, 0, 0, 0);
      }
      m[bj] = acc;
      if (bj == 0) {
        v4d a = cQ00;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f64_16x16x4f64(BP_W(0, ks), ks < 4 ? sw0[ks & 3] : sw1[ks & 3], a, 0, 0, 0);
        sn00 = a;
      }
      if (bj == 1) {
        v4d a = cQ01, b = cQ11;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a = __builtin_amdgcn_mfma_f64_16x16x4f64(BP_W(0, ks), ks < 4 ? sw0[ks & 3] : sw1[ks & 3], a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f64_16x16x4f64(JW ? cWT[ks] : (lx ? BP_W(1, ks) : 0.0), ks < 4 ? sw0[ks & 3] : sw1[ks & 3], b, 0, 0, 0);
        }
        sn01 = a; sn11 = b;
      }
    }
    // the outputs of the stage above leave here, behind the products (whose operand loads they would otherwise queue behind) and before
    // the tiles they lie over are written again
    if (k < k_top) flush(k + 1);
    RW2PROF(1);
    // the registers of W are the initial values of [Acl | bcl]
    v4d acl[2];                                                         // first block row only (rows 3..11 are what the roll-out reads)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
      for (int r = 0; r < 4; ++r) acl[bj][r] = BP_W(bj, r);
    // ---- elimination (riccati_wave.h): [G g H] through the tile, forward elimination + back substitution, Z / Yn for the update of S
    load_bt(k, nt);
    if (nt > 0) {
#pragma unroll
      for (int bj = 0; bj < NB; ++bj)
        if (bj < 2 || nbc > 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Mx[lk + 4 * r][16 * bj + li] = m[bj][r];
        }
    }
    lds_wave_sync();
    bool ok = true;
    if (nt > 0) {
      const int rpr = 16 - nt;
      const bool rows_layout = 4 * rpr >= NX + 1;
      const int rid = rows_layout ? lk * rpr + (li - nt) : l - nt;
      const bool is_h = rows_layout ? li < nt : l < nt;
      const bool rhs = !is_h && rid < NX + 1;
      const bool used = is_h || rhs;
      const int col = is_h ? BC + (rows_layout ? li : l) : (rhs ? rid : 0);
      if (rhs) {
        for (int i = nt; i < 4 * ksn; ++i) { Zt[i][col] = 0.0; Yn[i][col] = 0.0; }
      }
      const int ecol = rhs ? col : NX + 2 + (l & 3);
      auto emit = [&](int p, double z, double y) { Zt[p][ecol] = z; Yn[p][ecol] = y; };
#define BP_GJ_CASE(ROWS, FWD, BWD)                                                                             \
      {                                                                                                        \
        double v[ROWS];                                                                                        \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) { const double t = Mx[i][col]; v[i] = (used && i < nt) ? t : 0.0; } \
        lds_wave_sync();                                                                                       \
        ok = FWD;                                                                                              \
        BWD<ROWS>(v, nt);                                                                                      \
        double mt = (rhs && rid < NX) ? ws.qv[rid] : 0.0;                                                      \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i)                                                       \
          if (rhs && i < nt) { Mx[i][col] = v[i]; mt -= v[i] * ws.rv[i]; }                                     \
        if (rhs) ws.om[rid] = mt;                                                                              \
      }
      if (rows_layout) {
        if (nt <= 8) BP_GJ_CASE(8, forward_eliminate_rows<8>(v, nt, emit), back_substitute_rows)
        else if (nt == 9) BP_GJ_CASE(9, (forward_eliminate_rows<9, true>(v, nt, emit)), back_substitute_rows)
        else BP_GJ_CASE(10, (forward_eliminate_rows<10, true>(v, nt, emit)), back_substitute_rows)
      } else {
        if (nt <= 12) BP_GJ_CASE(12, forward_eliminate_wave<12>(v, nt, emit), back_substitute_wave)
        else BP_GJ_CASE(16, forward_eliminate_wave<16>(v, nt, emit), back_substitute_wave)
      }
#undef BP_GJ_CASE
    } else {
      if (l <= NX) ws.om[l] = l < NX ? ws.qv[l] : 0.0;
    }
    if (!__builtin_amdgcn_readfirstlane((int)ok)) status = 1;
    lds_wave_sync();
    RW2PROF(2);
    // ---- updates, one after the other
    double yb[4][2], zA[2][4], yn[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        yb[ks][b2] = -Mx[4 * ks + lk][16 * b2 + li];
        const double z = -Zt[4 * ks + lk][16 * b2 + li];
        zA[b2][ks] = (b2 == 0 || lx) ? z : 0.0;
        yn[ks][b2] = Yn[4 * ks + lk][16 * b2 + li];
      }
    load_pu(k, nt);
    // [S | s] = Sn - Z' Yn: blocks (0, 0), (0, 1), (1, 1); block (1, 0) is the mirror image of (0, 1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      if (ks < ksn) {
        sn00 = __builtin_amdgcn_mfma_f64_16x16x4f64(zA[0][ks], yn[ks][0], sn00, 0, 0, 0);
        sn01 = __builtin_amdgcn_mfma_f64_16x16x4f64(zA[0][ks], yn[ks][1], sn01, 0, 0, 0);
        sn11 = __builtin_amdgcn_mfma_f64_16x16x4f64(zA[1][ks], yn[ks][1], sn11, 0, 0, 0);
      }
    {
      double (*St)[34] = Zt;                                          // 32 rows: Zt and Yn are adjacent
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        St[lk + 4 * r][li] = sn00[r];
        St[lk + 4 * r][16 + li] = sn01[r];
        St[16 + lk + 4 * r][16 + li] = sn11[r];
      }
      lds_wave_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (the tile has 32 x 32 valid entries: no index is clamped, what lies beyond the matrix is read and masked - every clamp is a
        //  lane-dependent address of its own, and this kernel has no register to spare for it)
        const int row = lk + 4 * r;
        S[0][0][r] = 0.5 * (sn00[r] + St[li][row]);
        S[0][1][r] = sn01[r];
        const int row1 = 16 + row, col1 = 16 + li;
        const double t10 = St[li][row1];                              // S(16 + row, li) = S(li, 16 + row)
        S[1][0][r] = (16 + 4 * r < NX && row1 < NX) ? t10 : 0.0;
        const double t11 = St[col1][row1];
        S[1][1][r] = (row1 < NX && col1 < NX) ? 0.5 * (sn11[r] + t11) : sn11[r];
      }
      lds_wave_sync();
    }
    RW2PROF(3);
    // [Acl | bcl] = [A | b] - B Y  -> LDS (over the tiles of the elimination): rows 3..11 only - the first block row, registers 0..2 -
    // riccati_rollout_sparse derives the other rows of the closed-loop dynamics from du
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      v4d a = acl[bj];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < ksn) a = __builtin_amdgcn_mfma_f64_16x16x4f64(cBt[0][ks], yb[ks][bj], a, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int row = lk + 4 * r, col = 16 * bj + li;
        const bool ain = r > 0 || lk == 3;
        if (bj == 0) { if (ain) oA[row * NX + col] = a[r]; }
        else {
          if (ain && lx) oA[row * NX + col] = a[r];
          if (ain && li == XR) ws.ob[row] = a[r];
        }
      }
    }
    RW2PROF(4);
    // [K | kff] = [Px | Pe] - Pu Y  -> LDS; the force rows of Pu are generated from the contact mode
    {
      const int c0s = wave_stance_first(mode_c), nsf = wave_stance_count(mode_c);
      const int s = li - c0s;
      const bool stance = s >= 0 && s < nsf;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) cPu[0][ks] = li < 12 ? ((stance && s == 4 * ks + lk) ? 1.0 : 0.0) : cPu[0][ks];
    }
    RW2PROF(5);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        v4d a = cPI[bi][bj];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks < ksn) a = __builtin_amdgcn_mfma_f64_16x16x4f64(cPu[bi][ks], yb[ks][bj], a, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (16 * bi + 4 * r >= NX) continue;
          const int row = 16 * bi + lk + 4 * r, col = 16 * bj + li;
          const bool rin = 16 * bi + 4 * r + 3 < NX || row < NX;
          if (bj == 0) { if (rin) oK[row * NX + col] = a[r]; }
          else {
            if (rin && lx) oK[row * NX + col] = a[r];
            if (rin && li == XR) ws.ok[row] = a[r];
          }
        }
      }
    RW2PROF(6);
    lds_wave_sync();
    if (more) { load_g1(k - 1, nt_n); load_g2(k - 1, nt_n); }
    nt_c = nt_n; mode_c = mode_n; nt_n = nt_nn; mode_n = mode_nn;
    RW2PROF(7);
  }
#ifdef BPMPC_RICCATI_PROFILE
  if (io.prof && l == 0)
    for (int i = 0; i < 8; ++i) io.prof[i] = (double)tacc[i];
#endif
  flush(io.k_lo);
  if (io.k_lo > 0) {
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * bi + lk + 4 * r, col = 16 * bj + li;
          if (row < NX) {
            if (col < NX) io.carry[row * NX + col] = S[bi][bj][r];
            else if (col == NX) io.carry[NXX + row] = S[bi][bj][r];
          }
        }
  }
  if (l == 0) io.carry[NXX + NX] = (double)status;
#undef BP_W
#undef BP_B
}

}  // namespace bpmpc
