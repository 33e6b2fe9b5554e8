// Riccati sweep for batches of more than two problems per CU: ONE wavefront owns a problem from the terminal stage down (HIP only; same
// mathematics as riccati_mfma.h / riccati_mfma8.h, which share a problem among four or eight waves).
//
// With that many problems the sweep is a throughput problem, and the workgroup-per-problem kernels spend it badly: 66 KB of LDS per problem
// allow two workgroups per CU, every stage crosses five workgroup barriers, and a stage costs ~6.8 k cycles of a CU per problem against
// ~1.8 k of matrix-core time (114 v_mfma_f64_16x16x4_f64 over four SIMDs).  Loads, the elimination and HBM are not what binds there
// (measured at batch 4096: no prefetch 4.41 ms, no elimination 4.56 ms, no stores 3.67 ms against 4.56 ms; 3.2 TB/s).
// Here a wave runs alone on its SIMD with the whole register file (428 of 512 registers), four problems per CU, no workgroup barrier:
//   * the value function [S | s], the stage operands and all products live in REGISTERS in the accumulator layout of the matrix core
//     (lane l, register r <-> row (l / 16) + 4 r, column l % 16 of a 16x16 block).  That layout is at the same time the B-operand
//     layout of the block for four k-steps (k = row) and the A-operand layout of its transpose, so
//         SW = S W  (+ s in the b column)       A: S blocks (S is kept symmetric)              B: W as loaded
//         M += B' SW                            A: the B~ columns of W as loaded              B: SW as computed
//         Sn = Qq + A' SW                       A: W as loaded                                B: SW as computed
//     need no transposition and no LDS at all; only Acl = [A b] - B Y and [K kff] = [Px Pe] - Pu Y want B~ and Pu row-major in the
//     A-operand - those are loaded from HBM a second time in that orientation (2 KB per stage, L2 hits);
//   * the elimination of [H | G g] is forward elimination + back substitution in the column-per-lane layouts of riccati_fast.h /
//     riccati_mfma8.h (Gauss-Jordan lost 1e-9 of K per stage on the 24-state robot); its pivot rows Z (before) and Yn (after the division)
//     give S = Sn - Z' Yn, and S is made symmetric through LDS every stage (block (1, 0) is never computed: it is the mirror of (0, 1));
//   * LDS (25 KB per problem) carries those changes of layout and the output tiles: Acl, K, bcl, kff, m are assembled in their HBM
//     layout and leave as 16-byte chunks of consecutive addresses at the top of the NEXT stage (stores retire in order with the loads: a
//     store issued at the end of a stage would put its latency in front of the next stage's operands);
//   * every operand is BUFFER-loaded straight into the registers it is used from, one stage ahead, masked by its offset, and re-loaded
//     immediately after its last use (only W, which is needed from the first to the last product of a stage, is double buffered): the
//     loads of stage k-1 are spread over stage k and are consumed in the order they were issued;
//   * the force rows of [Px | Pe | Pu] are generated from the contact mode of the stage as in riccati_mfma.h (PwVtLoader).
// What a stage costs: a lone wave issues an instruction every ~5 cycles whatever it is, and the FP64 matrix instructions share their pipe
// with the FP64 VALU, so (instructions x 5) + (matrix instructions x 64) = 17 k cycles (tools/riccati_wave_phase_profile.py).  Every select
// around a load that the compiler turns into a branch with a full wait shows directly: keep loads unconditional, mask by address.
// The roll-out runs as a second launch (k_riccati_rollout, the routine of riccati_mfma.h): the status travels in the carry record.
#pragma once
#include <hip/hip_runtime.h>

#include "riccati_mfma.h"
#include "riccati_mfma8.h"

namespace bpmpc {

template <int NJ>
struct RiccatiWaveWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int LDM = PackedLq<NJ>::WP + 2;
  alignas(16) double Mx[16][LDM];          // [G | g | H] of the stage, then Y in its first nx + 1 columns
  alignas(16) double oA[NX * NX];          // outputs of the stage in their HBM layout
  alignas(16) double oK[NU * NX];
  double ob[NX], ok[NU], om[NX + 2];
  alignas(16) double Zt[16][34];           // Zt and Yn stay adjacent: together they are the 32-row tile of the symmetrisation of S
  alignas(16) double Yn[16][34];   // pivot rows of the elimination before / after their division (columns <= nx; nx + 2 .. spare)
  unsigned char nut[kMaxRiccatiStages];    // reduced input dimensions and contact modes of all stages (a global load per stage would make
  unsigned char mode[kMaxRiccatiStages];   // the wave wait for every operand load in flight)
  double rv[16];                           // r~ (column nx of Mt as loaded)
  double qv[32];                           // q~ (column nx of Qp as loaded)
};

__device__ __forceinline__ int wave_stance_first(int mode) { return mode == 2 ? 6 : 0; }
__device__ __forceinline__ int wave_stance_count(int mode) { return mode == 3 ? 12 : ((mode == 1 || mode == 2) ? 6 : 0); }   // 0 and kModeEvent: none

// JW off: the joint rows 12.. of Wt = [A~ | b~ | B~] are not in HBM (k_project_fast<.., WJ = false>): they are [I | b | 0] + dt Vt, loaded from
// Vt in the place of the rows of Wt and completed when the operands of the next stage become the current ones (riccati_wave2.h has the same).
template <int NJ, bool JW = true>
__device__ __forceinline__ void riccati_wave(RiccatiWaveWorkspace<NJ>& ws, const RiccatiFastIO& io) {
  using WS = RiccatiWaveWorkspace<NJ>;
  using PL = PackedLq<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, WP = PL::WP, QP = PL::QP, BC = NX + 1, NXX = NX * NX, NXU = NX * NU;
  constexpr int KS = (NX + 3) / 4;            // k-steps over the state dimension
  constexpr int NB = 3;                       // block columns of the packed width nx + 1 + nt, nt <= 16
  constexpr int XR = NX - 16;                 // rows / columns of the second block that belong to the state; column XR of it is the vector column
  static_assert(KS == 6 && NX + 1 <= 32 && NX + 1 + 16 <= 48 && WP >= 48 && NX == NU, "two block rows, three block columns");
  const int l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const int N = io.base.N;
  const int k_top = (io.k_hi < N ? io.k_hi : N) - 1;
  const bool resumed = io.k_hi < N;
  const double* const zero = io.zero_one + 2;

  // stages with more reduced inputs than a block row holds: fail loudly (status in the carry record, the roll-out reports it)
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

  // ---- the value function [S | s]: block (bi, bj), register r <-> row 16 bi + lk + 4 r, column 16 bj + li
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
  if (k_top < io.k_lo) {                     // nothing to sweep
    if (l == 0) io.carry[NXX + NX] = (double)status;
    return;
  }

  // ---- operand loads: BUFFER loads with a descriptor per matrix of the stage (num_records = the matrix), so that a lane is masked by
  //      its OFFSET: an offset beyond the matrix returns 0.0 without touching memory.  Masks that never change (rows >= nx, columns > nx,
  //      the vector column) are baked into the lane offsets once; masks of the stage (rows / columns >= nt) are one select on a 32-bit
  //      offset before the load.  The loaded value is final: it goes straight into the operand or accumulator registers of the matrix
  //      core, a stage ahead, with no instruction between the load and its use.  (First version: plain loads with clamped addresses and
  //      v_cndmask on the values at the head of each phase - ~170 selects on register pairs per stage plus the copies between the two
  //      register files that they forced.)
  constexpr unsigned kOut = 0x80000000u;      // beyond every matrix
  auto rsrc = [](const double* p, int doubles) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, doubles * 8, 0x00020000); };
  auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
  };
  // Few offset registers: rows that are a fixed number of bytes apart share one register, the distance goes into the instruction's
  // 12-bit offset field (kOut plus such a distance is still out of range).
  constexpr unsigned RS = 8u * 4 * WP;         // bytes between the rows of two k-steps / two accumulator registers of W, Mt, Vt
  constexpr int PKS = (2 * RS + 8 * 48 < 4096) ? 3 : 2;          // k-steps per offset register (their distances must fit 12 bits)
  constexpr int NG = (KS + PKS - 1) / PKS;
  unsigned gW[NG], gWT[NG];                   // W[4 ks + lk][li] (+ 8 BC for column BC + li), W[4 ks + lk][16 + li] for state columns only
  unsigned gWl, gWTl;                         // the k-step whose rows reach beyond nx has registers of its own
  constexpr int KL = (4 * KS > NX) ? KS - 1 : KS;                 // k-steps with all four rows inside the matrix
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    gW[g] = 8u * (unsigned)((4 * PKS * g + lk) * WP + li);
    gWT[g] = li < XR ? gW[g] + 128u : kOut;
  }
  gWl = (4 * (KS - 1) + lk < NX) ? 8u * (unsigned)((4 * (KS - 1) + lk) * WP + li) : kOut;
  gWTl = (4 * (KS - 1) + lk < NX && li < XR) ? gWl + 128u : kOut;
  auto offW = [&](int ks) { return ks < KL ? gW[ks / PKS] + RS * (unsigned)(ks % PKS) : gWl; };
  auto offWT = [&](int ks) { return ks < KL ? gWT[ks / PKS] + RS * (unsigned)(ks % PKS) : gWTl; };
  unsigned oBt[2], oPu[2];                    // W[16 bi + li][BC + lk] (+ 4 ks);  Vt[16 bi + li - 12][BC + lk] (+ 4 ks)
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
    const int row = 16 * bi + li, j = row - 12;
    oBt[bi] = row < (JW ? NX : 12) ? 8u * (unsigned)(row * WP + BC + lk) : kOut;      // (operand rows >= 12 only feed rows of Acl that are not kept)
    oPu[bi] = (j >= 0 && row < NX) ? 8u * (unsigned)(j * WP + BC + lk) : kOut;
  }
  // Qp[16 bi + lk + 4 r][li] and [16 + li] (columns <= nx): rows are 4 * 32 * 8 = 1024 bytes apart
  unsigned gQ0[2], gQ1[2], gQ0l, gQ1l;        // per block row; the register r whose rows reach beyond nx has its own
  constexpr int RQL = (NX - 16) / 4;          // first register of the second block row that is not complete (rows 16 + 4 RQL + lk)
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
    gQ0[bi] = 8u * (unsigned)((16 * bi + lk) * QP + li);
    gQ1[bi] = li <= XR ? gQ0[bi] + 128u : kOut;
  }
  gQ0l = (16 + 4 * RQL + lk < NX) ? gQ0[1] + 1024u * RQL : kOut;
  gQ1l = (16 + 4 * RQL + lk < NX && li <= XR) ? gQ0l + 128u : kOut;
  unsigned gM[2];                             // Mt[lk + 4 r][li]: registers 0, 1 and 2, 3
  gM[0] = 8u * (unsigned)(lk * WP + li);
  gM[1] = gM[0] + 2 * RS;
  // rows of [Px | Pe] in the accumulator layout: (0, r < 3) force rows lk + 4 r (zero except Pe in column nx); (0, 3), (1, 0), (1, 1) joint
  // rows jr = lk, 4 + lk, 8 + lk of Vt
  unsigned gV0, gV1, gV0l, gV1l, gPe;
  gV0 = 8u * (unsigned)(lk * WP + li);
  gV1 = li <= XR ? gV0 + 128u : kOut;
  gV0l = (8 + lk < NJ) ? gV0 + 2 * RS : kOut;
  gV1l = (8 + lk < NJ && li <= XR) ? gV0l + 128u : kOut;
  gPe = li == XR ? 8u * (unsigned)lk : kOut;

  // ---- operand registers
  double cW[KS][NB], nW[KS][NB];              // W = [A~ | b~ | B~]: B-operand of S W, A-operand of A' SW, initial value of Acl (double buffered)
  double cWT[KS];                             // its second block column restricted to the state columns: A-operand of the second block row of A' SW
  double cB[KS];                              // columns BC + li of W: A-operand of B' SW
  v4d cM[NB];                                 // Mt = [P~ | r~ | R~], rows < nt
  v4d cQ[2][2];                               // Qp = [Q~ | q~]
  double cBt[2][4];                           // B~ row-major in the A-operand: W[16 bi + li][BC + 4 ks + lk]
  double cPu[2][4];                           // Pu likewise (force rows generated)
  v4d cPI[2][2];                              // [Px | Pe] (force rows generated)

  // JW off: offsets of the joint rows in Vt (row 4 (ks - 3) + lk; the last k-step masked beyond nj), b of the joint rows of the stage being
  // loaded in the lanes of column nx, dt of that stage
  auto offJ = [&](int ks) { return ks == 3 ? gV0 : (ks == 4 ? gV0 + RS : gV0l); };
  double nbj[3] = {0.0, 0.0, 0.0};
  auto load_W = [&](double (&w)[KS][NB], int k, int nt) {
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
    const __amdgpu_buffer_rsrc_t rv = rsrc(io.Vt + (size_t)k * (NJ * WP), NJ * WP);
    const int nbc = (BC + nt + 15) >> 4;      // block columns >= nbc are not written by the change of variables: neither loaded nor used
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int bj = 0; bj < NB; ++bj) {
        if (JW || ks < 3) { if (bj < 2 || bj < nbc) w[ks][bj] = bload(rw, offW(ks) + 128u * bj); }
        else w[ks][bj] = bload(rv, (bj < 2 || bj < nbc) ? offJ(ks) + 128u * bj : kOut);
      }
    if constexpr (!JW) {
      const __amdgpu_buffer_rsrc_t rb = rsrc(io.lqb + (size_t)k * NX, NX);
#pragma unroll
      for (int q = 0; q < 3; ++q) nbj[q] = bload(rb, li == XR ? 8u * (unsigned)(12 + 4 * q + lk) : kOut);
    }
  };
  // completes the joint rows of a freshly loaded W, cB, cWT of stage k (JW off)
  auto complete_joint_rows = [&](double (&w)[KS][NB], int k) {
    const double dtk = io.gdt[k];
#pragma unroll
    for (int ks = 3; ks < KS; ++ks) {
      const int row = 4 * ks + lk;
      // (identity and b only in rows of the state: a padding row >= nx would put a 1 into the b column - harmless only while every operand of the padding rows is masked)
      w[ks][0] = dtk * w[ks][0] + ((row < NX && li == row) ? 1.0 : 0.0);
      const double w1 = dtk * w[ks][1] + ((row < NX && 16 + li == row) ? 1.0 : 0.0);
      cWT[ks] = li < XR ? w1 : 0.0;
      w[ks][1] = w1 + (row < NX ? nbj[ks - 3] : 0.0);
      w[ks][2] = dtk * w[ks][2];
      cB[ks] = dtk * cB[ks];
    }
  };
  auto load_BM = [&](int k, int nt) {
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
    const __amdgpu_buffer_rsrc_t rm = rsrc(io.Mt + (size_t)k * PL::M_SIZE, PL::M_SIZE);
    const int nbc = (BC + nt + 15) >> 4;
    const bool in = li < nt;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (JW || ks < 3) cB[ks] = bload(rw, in ? offW(ks) + 8u * BC : kOut);
      else cB[ks] = bload(rsrc(io.Vt + (size_t)k * (NJ * WP), NJ * WP), in ? offJ(ks) + 8u * BC : kOut);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned o = (lk + 4 * r < nt) ? gM[r >> 1] + RS * (unsigned)(r & 1) : kOut;
#pragma unroll
      for (int bj = 0; bj < NB; ++bj)
        if (bj < 2 || bj < nbc) cM[bj][r] = bload(rm, o + 128u * bj);
    }
  };
  auto load_Q = [&](int k) {
    const __amdgpu_buffer_rsrc_t rq = rsrc(io.Qp + (size_t)k * PL::Q_SIZE, PL::Q_SIZE);
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
#pragma unroll
    for (int ks = 0; ks < (JW ? KS : 3); ++ks) cWT[ks] = bload(rw, offWT(ks));      // (JW off: the joint rows come from nW when it is completed)
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (bi == 1 && r > RQL) { cQ[bi][0][r] = 0.0; cQ[bi][1][r] = 0.0; }
        else if (bi == 1 && r == RQL && 16 + 4 * RQL < NX) { cQ[bi][0][r] = bload(rq, gQ0l); cQ[bi][1][r] = bload(rq, gQ1l); }
        else if (bi == 1 && r == RQL) { cQ[bi][0][r] = 0.0; cQ[bi][1][r] = 0.0; }
        else { cQ[bi][0][r] = bload(rq, gQ0[bi] + 1024u * r); cQ[bi][1][r] = bload(rq, gQ1[bi] + 1024u * r); }
      }
  };
  auto load_late = [&](int k, int nt, int mode) {
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
    const __amdgpu_buffer_rsrc_t rv = rsrc(io.Vt + (size_t)k * (NJ * WP), NJ * WP);
    const __amdgpu_buffer_rsrc_t rp = rsrc(io.base.Pe + (size_t)k * NU, NU);
    const int c0s = wave_stance_first(mode), nsf = wave_stance_count(mode);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int col = 4 * ks + lk;                         // reduced input
      const bool in = col < nt;
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        if (bi == 0) cBt[bi][ks] = bload(rw, in ? oBt[bi] + 32u * ks : kOut);      // (Acl rows 3..11 only: the first block row)
        cPu[bi][ks] = bload(rv, in ? oPu[bi] + 32u * ks : kOut);
      }
    }
    // [Px | Pe]: force rows are zero except Pe in column nx; joint rows from Vt (written up to the last block column of the stage)
#pragma unroll
    for (int r = 0; r < 3; ++r) { cPI[0][0][r] = 0.0; cPI[0][1][r] = bload(rp, gPe + 32u * r); }
    cPI[0][0][3] = bload(rv, gV0); cPI[0][1][3] = bload(rv, gV1);                     // joints lk
    cPI[1][0][0] = bload(rv, gV0 + RS); cPI[1][1][0] = bload(rv, gV1 + RS);           // joints 4 + lk
    cPI[1][0][1] = bload(rv, gV0l); cPI[1][1][1] = bload(rv, gV1l);                   // joints 8 + lk (< nj)
    cPI[1][0][2] = 0.0; cPI[1][0][3] = 0.0; cPI[1][1][2] = 0.0; cPI[1][1][3] = 0.0;
    (void)c0s; (void)nsf;
  };
  // the force rows of Pu (first block row of the A-operand, lanes li < 12): a single one in the component's own reduced-input column
  auto finish_late = [&](int mode) {
    const int c0s = wave_stance_first(mode), nsf = wave_stance_count(mode);
    const int s = li - c0s;
    const bool stance = s >= 0 && s < nsf;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) cPu[0][ks] = li < 12 ? ((stance && s == 4 * ks + lk) ? 1.0 : 0.0) : cPu[0][ks];
  };

  // outputs of stage hk, assembled in LDS by its last phase, to HBM in 16-byte chunks
  auto flush = [&](int hk) {
    double2* A2 = reinterpret_cast<double2*>(io.Acl + (size_t)hk * NXX);
    double2* K2 = reinterpret_cast<double2*>(io.Kfull + (size_t)hk * NXU);
    const double2* a2 = reinterpret_cast<const double2*>(ws.oA);
    const double2* k2 = reinterpret_cast<const double2*>(ws.oK);
    // all LDS reads first (read - wait - store per chunk exposes the LDS latency eight times), the last chunk clamped instead of predicated
    constexpr int NIT = (NXX / 2 + kWave - 1) / kWave;
    double2 ta[NIT], tk[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = l + it * kWave, ic = idx < NXX / 2 ? idx : NXX / 2 - 1;
      ta[it] = a2[ic]; tk[it] = k2[ic];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = l + it * kWave;
      if (it + 1 < NIT || idx < NXX / 2) K2[idx] = tk[it];
      if (idx >= 3 * NX / 2 && idx < 12 * NX / 2) A2[idx] = ta[it];       // rows 3..11 of Acl
    }
    if (l < NX) {
      if (l >= 3 && l < 12) io.bcl[(size_t)hk * NX + l] = ws.ob[l];
      io.kff[(size_t)hk * NU + l] = ws.ok[l];
      io.mvec[(size_t)hk * NX + l] = ws.om[l];
    }
    if (l == NX) io.mscal[hk] = ws.om[NX];
  };

  // ---- prologue: everything of the top stage; contact modes and input dimensions two stages ahead (scalar loads)
  auto stage_nt = [&](int k) { return __builtin_amdgcn_readfirstlane((int)ws.nut[k >= io.k_lo ? k : io.k_lo]); };
  auto stage_mode = [&](int k, int) { return __builtin_amdgcn_readfirstlane((int)ws.mode[k >= io.k_lo ? k : io.k_lo]); };
  int nt_c = stage_nt(k_top), mode_c = stage_mode(k_top, nt_c);
  int nt_n = stage_nt(k_top - 1), mode_n = stage_mode(k_top - 1, nt_n);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) { cW[ks][bj] = 0.0; nW[ks][bj] = 0.0; }
#pragma unroll
  for (int bj = 0; bj < NB; ++bj) cM[bj] = v4d{0.0, 0.0, 0.0, 0.0};
  load_W(cW, k_top, nt_c);
  load_BM(k_top, nt_c);
  load_Q(k_top);
  load_late(k_top, nt_c, mode_c);
  if constexpr (!JW) complete_joint_rows(cW, k_top);

#ifdef BPMPC_RICCATI_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define RWPROF(slot) do { const long long tn_ = clock64(); tacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define RWPROF(slot) ((void)0)
#endif
  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = nt_c;
    const int nbc = (BC + nt + 15) >> 4;
    const int ksn = (nt + 3) >> 2;
    const bool more = k > io.k_lo;
    const int nt_nn = stage_nt(k - 2), mode_nn = stage_mode(k - 2, nt_nn);
    // ---- outputs of the stage above, then the first loads of the stage below
    if (k < k_top) flush(k + 1);
    if (more) load_W(nW, k - 1, nt_n);
    RWPROF(0);
    // ---- P1: SW = S' W, s added to the b column
    v4d sw[2][NB];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < NB; ++bj) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        if (bj < nbc) {
          if (bj == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = li == XR ? S[bi][1][r] : 0.0;
          }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            double a = S[ks >> 2][bi][ks & 3];                        // S(4 ks + lk, 16 bi + li)
            if (bi == 1) a = li < XR ? a : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, cW[ks][bj], acc, 0, 0, 0);
          }
        }
        sw[bi][bj] = acc;
      }
    RWPROF(1);
    // r~ and q~ as loaded (m = q~ - Y' r~ below).  Here, not at the top of the stage: a use of a loaded register is a wait, and at the top
    // the loads issued at the end of the stage above are still on their way.
    if (li == XR) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ws.rv[lk + 4 * r] = cM[1][r];
        ws.qv[lk + 4 * r] = cQ[0][1][r];
        ws.qv[16 + lk + 4 * r] = cQ[1][1][r];
      }
    }
    // ---- P2: [G | g | H] = [P | r | R] + B' SW
    v4d m[NB];
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) {
      v4d acc = cM[bj];
      if (bj < nbc && nt > 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cB[ks], sw[ks >> 2][bj][ks & 3], acc, 0, 0, 0);
      }
      m[bj] = acc;
    }
    if (more) load_BM(k - 1, nt_n);
    RWPROF(2);
    // ---- P3: [Sn | sn] = [Q | q] + A' SW
    // (Issued two per pivot step inside the elimination - hooks in forward_eliminate_rows - these 24 instructions made the stage SLOWER,
    //  16.9 k -> 18.4 k cycles: the FP64 matrix instructions and the FP64 VALU of a SIMD do not overlap, a matrix instruction in the
    //  dependent chain of the elimination costs its 64 cycles plus the chain's re-start.)
    v4d sn[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        v4d acc = cQ[bi][bj];
        if (bi == 1 && bj == 0) { sn[bi][bj] = acc; continue; }     // block (1, 0) of S is the transpose of block (0, 1): mirrored below
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bi == 0 ? cW[ks][0] : cWT[ks], sw[ks >> 2][bj][ks & 3], acc, 0, 0, 0);      // A(4 ks + lk, 16 bi + li)
        }
        sn[bi][bj] = acc;
      }
    if (more) load_Q(k - 1);
    RWPROF(3);
    // ---- elimination: Y = H^-1 [G g], one column per lane (riccati_fast.h), through the LDS tile
    if (nt > 0) {
#pragma unroll
      for (int bj = 0; bj < NB; ++bj)
        if (bj < nbc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ws.Mx[lk + 4 * r][16 * bj + li] = m[bj][r];
        }
    }
    lds_wave_sync();
    bool ok = true;
    if (nt > 0) {
      // Forward elimination and back substitution (riccati_fast.h / riccati_mfma8.h), NOT Gauss-Jordan: on the 24-state robot H has a
      // condition number of 2.6e5 and gains of 6e3, where Gauss-Jordan loses 1e-9 of K in the first stage and 4e-6 over fifty (measured
      // against the oracle; elimination + substitution: 1e-12).  Z (pivot rows before their division) and Yn (after) give the value
      // function in the symmetric form S = Sn - Z' Yn.
      // Lane layout: with 4 (16 - nt) >= nx + 1 right-hand sides the DPP rows (every 16-lane row holds H in its lanes 0..nt-1 and 16 - nt
      // right-hand sides), otherwise one column of [H | G g] per lane (v_readlane).
      const int rpr = 16 - nt;
      const bool rows_layout = 4 * rpr >= NX + 1;
      const int rid = rows_layout ? lk * rpr + (li - nt) : l - nt;
      const bool is_h = rows_layout ? li < nt : l < nt;
      const bool rhs = !is_h && rid < NX + 1;
      const bool used = is_h || rhs;
      const int col = is_h ? BC + (rows_layout ? li : l) : (rhs ? rid : 0);
      if (rhs) {
        for (int i = nt; i < 4 * ksn; ++i) { ws.Zt[i][col] = 0.0; ws.Yn[i][col] = 0.0; }
      }
      // every lane stores, the ones without a right-hand side into spare columns (never read as Z; as Yn they only make columns > nx of S)
      const int ecol = rhs ? col : NX + 2 + (l & 3);
      auto emit = [&](int p, double z, double y) { ws.Zt[p][ecol] = z; ws.Yn[p][ecol] = y; };
#define BP_GJ_CASE(ROWS, FWD, BWD)                                                                             \
      {                                                                                                        \
        double v[ROWS];                                                                                        \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) { const double t = ws.Mx[i][col]; v[i] = (used && i < nt) ? t : 0.0; } \
        lds_wave_sync();                                                                                       \
        ok = FWD;                                                                                              \
        BWD<ROWS>(v, nt);                                                                                      \
        double mt = (rhs && rid < NX) ? ws.qv[rid] : 0.0;                                                      \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i)                                                       \
          if (rhs && i < nt) { ws.Mx[i][col] = v[i]; mt -= v[i] * ws.rv[i]; }                                  \
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
      if (l <= NX) ws.om[l] = l < NX ? ws.qv[l] : 0.0;               // event node: m = q~, m0 = 0
    }
    if (!__builtin_amdgcn_readfirstlane((int)ok)) status = 1;
    lds_wave_sync();
    RWPROF(4);
    // ---- P4: [S | s] = Sn - G' Y, [Acl | bcl] = [A | b] - B Y, [K | kff] = [Px | Pe] - Pu Y
    finish_late(mode_c);
    double yb[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) yb[ks][bj] = -ws.Mx[4 * ks + lk][16 * bj + li];   // unconditional (a select on ksn compiles to a branch and a full LDS wait per read); k-steps >= ksn are not issued, rows >= nt of the tile are zero
    double zA[2][4], yn[4][2];                                         // -Z' in the A-operand (Z(4 ks + lk, 16 bi + li)), Yn in the B-operand
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const double z = -ws.Zt[4 * ks + lk][16 * b2 + li];
        zA[b2][ks] = (b2 == 0 || li < XR) ? z : 0.0;
        yn[ks][b2] = ws.Yn[4 * ks + lk][16 * b2 + li];
      }
    v4d acl[2][2], kf[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        v4d a0 = sn[bi][bj], a1, a2 = cPI[bi][bj];
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[r] = (4 * bi + r < KS) ? cW[(4 * bi + r < KS) ? 4 * bi + r : 0][bj] : 0.0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks < ksn) {
            if (!(bi == 1 && bj == 0)) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(zA[bi][ks], yn[ks][bj], a0, 0, 0, 0);
            if (bi == 0) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(cBt[bi][ks], yb[ks][bj], a1, 0, 0, 0);   // the roll-out reads the rows 3..11 of Acl only
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(cPu[bi][ks], yb[ks][bj], a2, 0, 0, 0);
          }
        S[bi][bj] = a0; acl[bi][bj] = a1; kf[bi][bj] = a2;
      }
    RWPROF(5);
    if (more) load_late(k - 1, nt_n, mode_n);
    // S made symmetric through LDS (the tiles of Z and Yn are free: their operands are in registers): the diagonal blocks become
    // (S + S') / 2, block (1, 0) - never computed: 9 matrix-core instructions less per stage - is the transpose of block (0, 1).  The
    // next stage reads S(k, i) for S(i, k); without this step the rounding asymmetry of a hundred stages shows at the 24-state robot
    // (2e-10 against the other sweeps after three iterations, 1e-11 with it).
    {
      double (*St)[34] = ws.Zt;                                       // 32 rows: Zt and Yn are adjacent
#pragma unroll
      for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
          for (int r = 0; r < 4; ++r) St[16 * bi + lk + 4 * r][16 * bj + li] = S[bi][bj][r];
      lds_wave_sync();
#pragma unroll
      for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (bi == 0 && bj == 1) continue;                         // kept as computed (its mirror image is block (1, 0))
            const int row = 16 * bi + lk + 4 * r, col = 16 * bj + li;
            if (16 * bi + 4 * r >= NX) continue;
            const double t = St[col < NX ? col : 0][row < NX ? row : 0];
            if (bi == 1 && bj == 0) S[bi][bj][r] = row < NX ? t : 0.0;
            else if (row < NX && col < NX) S[bi][bj][r] = 0.5 * (S[bi][bj][r] + t);
          }
    }
    // outputs into their HBM layout
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (16 * bi + 4 * r >= NX) continue;                        // (whole registers beyond the matrix: decided at compile time)
          const int row = 16 * bi + lk + 4 * r, col = 16 * bj + li;
          const bool rin = 16 * bi + 4 * r + 3 < NX || row < NX;
          // [Acl | bcl]: rows 3..11 only (riccati_rollout_sparse derives the other rows from du); column nx is the vector column
          const bool ain = bi == 0 && r < 3 && (r > 0 || lk == 3);
          if (bj == 0) {
            if (rin) ws.oK[row * NX + col] = kf[bi][bj][r];
            if (ain) ws.oA[row * NX + col] = acl[bi][bj][r];
          } else if (rin && li <= XR) {
            double* dk = li < XR ? &ws.oK[row * NX + col] : &ws.ok[row];
            *dk = kf[bi][bj][r];
            if (ain) { double* da = li < XR ? &ws.oA[row * NX + col] : &ws.ob[row]; *da = acl[bi][bj][r]; }
          }
        }
    lds_wave_sync();
    // ---- next stage
    if (more) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int bj = 0; bj < NB; ++bj) cW[ks][bj] = nW[ks][bj];
      if constexpr (!JW) complete_joint_rows(cW, k - 1);
    }
    nt_c = nt_n; mode_c = mode_n; nt_n = nt_nn; mode_n = mode_nn;
    RWPROF(6);
  }
#ifdef BPMPC_RICCATI_PROFILE
  if (io.prof && l == 0)
    for (int i = 0; i < 8; ++i) io.prof[i] = (double)tacc[i];
#endif
  flush(io.k_lo);
  // ---- hand over: S, s and the status to the launch that sweeps the earlier stages; the status alone to the roll-out
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
}

// Roll-out, step norms and the opening of the line search after riccati_wave: the routine of the workgroup kernels on its own.
static_assert(offsetof(RiccatiWaveWorkspace<10>, Yn) == offsetof(RiccatiWaveWorkspace<10>, Zt) + sizeof(double) * 16 * 34, "Zt, Yn adjacent");

template <int NJ>
struct RiccatiRolloutWorkspace {
  static constexpr int NX = 12 + NJ;
  static constexpr int kCap = 152;                                                   // stages of history per pass
  static constexpr int kDoubles = (kCap + 4 + 8) * NX + kStepNormsScratch * kRiccatiThreads / kWave;
  static constexpr int kCapSparse = (kDoubles - 200 - 64 - 4 * NX) / (2 * NX);          // riccati_rollout_sparse: dx and du histories of a pass
  alignas(16) double hist[kDoubles];
};
// (riccati_rollout_sparse: riccati_mfma.h)
template <int NJ>
__device__ __forceinline__ void riccati_rollout_only(RiccatiRolloutWorkspace<NJ>& ws, const RiccatiFastIO& io) {
  constexpr int NX = 12 + NJ, NXX = NX * NX;
  const int st = (int)io.carry[NXX + NX];
  __syncthreads();
  riccati_rollout_sparse<NJ>(ws.hist, RiccatiRolloutWorkspace<NJ>::kCapSparse, st, io);
}
// The same roll-out in workgroups of TWO waves (recurrence; opening of the line search) on a quarter of a CU's LDS: the recurrence is a
// chain of ~1 600 cycles per stage in which the SIMD idles, so what a batch of problems needs is more recurrences in flight - four per CU
// instead of two.  The history holds ~100 stages; a longer horizon takes another pass (riccati_rollout_sparse).
constexpr int kRolloutPairThreads = 2 * kWave;
template <int NJ>
struct RiccatiRolloutPairWorkspace {
  static constexpr int NX = 12 + NJ;
  static constexpr int kDoubles = 5040;                                              // 39.4 KB: four workgroups per CU
  static constexpr int kCapSparse = (kDoubles - 200 - 64 - 4 * NX) / (2 * NX);       // 106 stages at nx = 22, 97 at nx = 24
  alignas(16) double hist[kDoubles];
};
template <int NJ>
__device__ __forceinline__ void riccati_rollout_pair(RiccatiRolloutPairWorkspace<NJ>& ws, const RiccatiFastIO& io) {
  constexpr int NX = 12 + NJ, NXX = NX * NX;
  const int st = (int)io.carry[NXX + NX];
  __syncthreads();
  riccati_rollout_sparse<NJ, kRolloutPairThreads>(ws.hist, RiccatiRolloutPairWorkspace<NJ>::kCapSparse, st, io);
}

}  // namespace bpmpc
