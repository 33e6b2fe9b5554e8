// Riccati sweep, eight wavefronts per problem, with the stage data staged by LDS-DMA (HIP only; gfx950).
//
// Same mathematics, roles and phases as riccati_mfma8.h.  What changed is how a stage's operands reach LDS.  There the loader waves
// prefetch the packed projected model into registers a stage ahead and write it to LDS in phase P0 - forty ds_write_b128 per lane,
// ~1.2 k of the ~6.1 k cycles a stage takes, ON the chain (the staging barrier B0 waits for it).  Here
//   * W = [A~ | b~ | B~], Qq = [Q~ | q~], M = [P~ | r~ | R~] travel HBM -> LDS by `global_load_lds_dwordx4` (16 bytes per lane, the
//     destination is wave-uniform base + 16 lane, the SOURCE address is per lane): every 16-byte chunk of the LDS image names its
//     chunk of the packed HBM layout (PackedLq, row stride 48 instead of 50 doubles), and chunks that must read as zero - padding
//     columns, block columns >= nbc and rows >= nut that the projection kernel does not write - name a 16-byte zero page.  No
//     register, no ds_write, no select;
//   * the requests of stage k - 1 are issued in phase P1 of stage k and waited for (`s_waitcnt vmcnt(0)`, by the loader waves only)
//     right before the staging barrier of stage k - 1: a whole stage of latency hiding.  That needs the destination to be free a
//     stage early, while the outputs of stage k + 1 (Acl = A~ - B~ Y, K = Px - Pu Y, finished beside the chain in P3 of stage k) still
//     read W and [Px | Pe | Pu] of stage k + 1: W and PW are TRIPLE buffered (index k mod 3), Qq and M double buffered, and what
//     the late readers need of those two lives elsewhere - the gain Y goes to its own matrix Yb instead of back into M, r~ and q~
//     are copied out in P1 (r~ to LDS, q~ into a register of the wave that computes m);
//   * [Px | Pe | Pu] still goes through registers (three separate arrays with 22-double rows: Pu's rows start on an odd column of
//     the LDS image, a 16-byte DMA chunk cannot land there).
// LDS: 151 KB at nx = 22 (one workgroup per CU, as before).  nx = 24 does not fit three copies and keeps riccati_mfma8.h.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "riccati_mfma8.h"

namespace bpmpc {

template <int NJ>
struct RiccatiDma8Workspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int RB = 32, RE = 16, LDN = 34;
  static constexpr int WC = NX + 1 + NU;
  static constexpr int LDW = ((WC + 15) / 16) * 16 + 2;
  static_assert(NX + 1 <= 32 && NU <= 32, "two block rows / columns");
  alignas(16) double S[RB][LDN];        // [S | s], not symmetrised
  alignas(16) double Sn[RB][LDN];       // [Sn | sn]
  alignas(16) double Zt[RE][LDN];       // pivot rows of the forward elimination of [G | g]
  alignas(16) double Yn[RE][LDN];       // the same rows divided by their pivots
  alignas(16) double Yb[RE][LDN];       // -Y = -H^-1 [G g] of the stage that was eliminated last (read by the output blocks a stage later)
  alignas(16) double SW[RB][LDW];       // sym(S) W
  alignas(16) double Qq[2][RB][LDN];    // [Q~ | q~]                               (DMA)
  alignas(16) double M[2][RE][LDW];     // [P~ | r~ | R~] -> [G | g | H]           (DMA)
  alignas(16) double W[3][RB][LDW];     // [A~ | b~ | B~]                          (DMA)
  alignas(16) double PW[3][RB][LDW];    // [Px | Pe | Pu]
  double r[2][RE];                      // r~ of the stage, copied out of M before G overwrites it
  int status;
  unsigned char nut[kMaxRiccatiStages];
  unsigned char mode[kMaxRiccatiStages];
};

// One LDS-DMA request of 16 bytes per lane: LDS[lds_dst + 16 lane] <- *gsrc.  M0 carries the destination base and belongs to the
// compiler: saved and restored inside the statement (cdna_hip_programming.md, inline assembly notes).  hipcc does not count this
// request: the issuing wave waits with wait_dma() before the barrier that publishes the data.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst /* wave-uniform LDS byte address */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
#ifndef BPMPC_DMA8_ISSUERS
#define BPMPC_DMA8_ISSUERS 2
#endif
#ifndef BPMPC_DMA8_ABLATE
#define BPMPC_DMA8_ABLATE 0      // timing experiments (wrong results): 1 no requests, 2 no wait for them
#endif
__device__ __forceinline__ void wait_dma() {
#if BPMPC_DMA8_ABLATE != 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ unsigned lds_address(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// The requests of one stage.  The rows 0..nx-1 of W, the rows 0..nx-1 of Qq and the RE rows of M are runs of consecutive 16-byte
// chunks of LDS; request i of a matrix covers its chunks 64 i .. 64 i + 63 (rows >= nx of W and Qq stay zero from the start, a request
// that runs past row nx - 1 writes zeros there).  The NI requests of a stage are dealt round robin to the NLW loader waves; what a
// lane needs per request is fixed for the whole sweep (byte offset of the chunk inside the node's packed array, its first column, its
// row) and kept in registers.
template <int NJ, int NLW, int LDW, int LDN, int RE>
struct DmaStageRequests {
  using PL = PackedLq<NJ>;
  static constexpr int NX = PL::NX, NU = PL::NU, WP = PL::WP, QP = PL::QP, BC = NX + 1;
  static constexpr int CRW = LDW / 2, CRQ = LDN / 2;                       // chunks per LDS row
  static constexpr int HQU = (NX + 2) / 2;                                 // pairs per row of Qp that carry anything ([Q~ | q~]: nx + 1 columns)
  static constexpr int IW = (NX * CRW + 63) / 64, IQ = (NX * CRQ + 63) / 64, IM = (RE * CRW + 63) / 64;
  static constexpr int NI = IW + IQ + IM, JMAX = (NI + NLW - 1) / NLW;
  static_assert(LDW % 2 == 0 && LDN % 2 == 0 && WP % 2 == 0 && QP % 2 == 0 && WP <= LDW && QP <= LDN, "16-byte chunks stay inside the rows");
  static_assert(IW * 64 <= 32 * CRW && IQ * 64 <= 32 * CRQ, "a request never runs past its matrix (M: predicated)");
  int goff[JMAX];      // >= 0: byte offset in the node's array; -1: zero page; -2: no request for this lane
  int gcol[JMAX];      // first column of the chunk (W, M: compared with the first unwritten column of the stage)
  int grow[JMAX];      // row (M: compared with nut)
  int lw;              // loader wave 0..NLW-1

  __device__ __forceinline__ void init(int lw_, int l) {
    lw = lw_;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int i = lw + j * NLW;
      goff[j] = -2; gcol[j] = 0; grow[j] = 0;
      if (i < IW) {
        const int c = 64 * i + l, row = c / CRW, p = c % CRW;
        goff[j] = (row < NX && p < WP / 2) ? (row * WP + 2 * p) * 8 : -1;
        gcol[j] = 2 * p;
      } else if (i < IW + IQ) {
        const int c = 64 * (i - IW) + l, row = c / CRQ, p = c % CRQ;
        goff[j] = (row < NX && p < HQU) ? (row * QP + 2 * p) * 8 : -1;
      } else if (i < NI) {
        const int c = 64 * (i - IW - IQ) + l, row = c / CRW, p = c % CRW;
        goff[j] = c < RE * CRW ? ((p < WP / 2) ? (row * WP + 2 * p) * 8 : -1) : -2;
        gcol[j] = 2 * p; grow[j] = row;
      }
    }
  }
  // requests J0 .. J1 - 1 of this wave for stage k (nt reduced inputs) into the LDS matrices at the given byte addresses
  template <int J0, int J1>
  __device__ __forceinline__ void issue(const RiccatiFastIO& io, size_t k, int nt, const double* zero_page, unsigned ldsW, unsigned ldsQ, unsigned ldsM) const {
    const int cend = 16 * ((BC + nt + 15) >> 4);                           // first column the projection kernel does not write
    const char* gW = reinterpret_cast<const char*>(io.Wt + k * PL::W_SIZE);
    const char* gQ = reinterpret_cast<const char*>(io.Qp + k * PL::Q_SIZE);
    const char* gM = reinterpret_cast<const char*>(io.Mt + k * PL::M_SIZE);
    const char* zp = reinterpret_cast<const char*>(zero_page);
#pragma unroll
    for (int j = J0; j < (J1 < JMAX ? J1 : JMAX); ++j) {
      const int i = lw + j * NLW;                                          // wave-uniform
      if (i < NI) {
        const bool isW = i < IW, isQ = !isW && i < IW + IQ;
        const char* base = isW ? gW : (isQ ? gQ : gM);
        const unsigned dst = isW ? ldsW + 1024u * i : (isQ ? ldsQ + 1024u * (i - IW) : ldsM + 1024u * (i - IW - IQ));
        const bool live = goff[j] >= 0 && (isQ || gcol[j] < cend) && (isW || isQ || grow[j] < nt);
        const char* src = live ? base + goff[j] : zp;
#if BPMPC_DMA8_ABLATE != 1
        if (goff[j] != -2) glds16(src, __builtin_amdgcn_readfirstlane(dst));
#else
        (void)src; (void)dst;
#endif
      }
    }
  }
};

template <int NJ>
__device__ __forceinline__ void riccati_dma8(RiccatiDma8Workspace<NJ>& ws, const RiccatiFastIO& io, const double* zero_page) {
  using WS = RiccatiDma8Workspace<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, NT = kRiccati8Threads, LDN = WS::LDN, LDW = WS::LDW, RE = WS::RE;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  constexpr int KS = (NX + 3) / 4;          // k-steps over the state dimension
  constexpr int BC = NX + 1;                // first column of B~ / Pu / R~ in the packed layouts
  static_assert(NX == NU, "packed layouts assume nx == nu");
  static_assert(NX + 1 + RE <= kWave, "one lane per column of [H | G g]");
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int li = l & 15, lk = l >> 4;       // operand row/column index and k index of this lane
  const int N = io.base.N;
  constexpr int NLW = 3;                    // loader waves 4, 5, 6 (L4, L5, F)
  const bool role_c = w < 4, role_l = w >= 4 && w < 4 + NLW, role_f = w == 6, role_e = w == 7;

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

  // The requests are issued by wave 4 alone: it never stores to global memory, so its `s_waitcnt vmcnt(0)` before the staging barrier
  // waits for the requests and nothing else.  (First version: dealt to the three loader waves - waves 5 and 6 then also waited for the
  // output stores they issue in P3, whose acknowledgement takes longer than the rest of the stage: 0.334 -> 0.428 ms.)
  constexpr int NLD = NLW * kWave;
  constexpr int NDW = BPMPC_DMA8_ISSUERS;   // waves 4 .. that issue the requests; none of them stores to global memory
  const bool role_d = w >= 4 && w < 4 + NDW;
  DmaStageRequests<NJ, NDW, LDW, LDN, RE> dma;
  PwVtLoader<NJ, NLD, LDW> pw;
  dma.init(role_d ? w - 4 : 0, l);
  pw.init(io, tid - 4 * kWave, role_l, (size_t)(k_top > 0 ? k_top : 0));
  const unsigned ldsW0 = lds_address(&ws.W[0][0][0]), ldsQ0 = lds_address(&ws.Qq[0][0][0]), ldsM0 = lds_address(&ws.M[0][0][0]);
  constexpr unsigned kWBytes = sizeof(ws.W[0]), kQBytes = sizeof(ws.Qq[0]), kMBytes = sizeof(ws.M[0]);
  using DMA = DmaStageRequests<NJ, NDW, LDW, LDN, RE>;
  constexpr int J1 = (DMA::JMAX + 2) / 3, J2 = (2 * DMA::JMAX + 2) / 3;      // the requests of a stage are issued in three parts: P1, P2, P3
  auto issue_part = [&](int k, int nt, auto part) {   // wave 4: a third of the requests of stage k (nt reduced inputs) into its buffers (k mod 3, k mod 2)
    constexpr int P = decltype(part)::value;
    dma.template issue<(P == 0 ? 0 : (P == 1 ? J1 : J2)), (P == 0 ? J1 : (P == 1 ? J2 : DMA::JMAX))>(io, (size_t)k, nt, zero_page, ldsW0 + kWBytes * (unsigned)(k % 3), ldsQ0 + kQBytes * (unsigned)(k & 1),
              ldsM0 + kMBytes * (unsigned)(k & 1));
  };
  if (role_l && k_top >= io.k_lo) { const int n0 = io.base.nut[k_top]; pw.prefetch(n0, n0 > 0 ? (io.mode[k_top] & 3) : kModeEvent); }
  using Part0 = std::integral_constant<int, 0>; using Part1 = std::integral_constant<int, 1>; using Part2 = std::integral_constant<int, 2>;
  if (role_d && k_top >= io.k_lo) {                                        // (the LDS copy of nut may not be visible yet)
    const int nt0 = io.base.nut[k_top];
    issue_part(k_top, nt0, Part0{}); issue_part(k_top, nt0, Part1{}); issue_part(k_top, nt0, Part2{});
  }
  __syncthreads();

  // Outputs of a stage that are not on the chain (block bw of each): [Acl | bcl] = [A | b] - B Y, [K | kff] = [Px | Pe] - Pu Y,
  // from the buffers that stage was staged into (k mod 3) and the gain matrix Yb.
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

  // m = q~ - Y' r~, m0 = -r~' H^-1 g of a finished stage (one wave; q~ of that stage in `qv`, lane l = component l).  Rows >= nt of
  // r~ are zero, the matching rows of Yb hold finite leftovers of wider stages: all loads first, no branch per row.
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

  constexpr int kMWave = NDW >= 2 ? 6 : 5;   // the wave that computes m: wave 5 unless it issues requests (then F, after its output block)
  int pend_k = -1, pend_nt = 0;          // stage whose outputs are still to be finished (uniform)
  double q_pend = 0.0, q_cur = 0.0;      // wave kMWave: q~ of the pending / the current stage (lane l = component l)
  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = ws.nut[k];            // max_nodes <= kMaxRiccatiStages is checked when the solver is created
    const int b2 = k & 1, b3 = k % 3;
    double (*const W)[LDW] = ws.W[b3];
    double (*const PW)[LDW] = ws.PW[b3];
    double (*const Qq)[LDN] = ws.Qq[b2];
    double (*const M)[LDW] = ws.M[b2];
    const int ksn = (nt + 3) >> 2;                   // k-steps over the reduced input
    const int nbc = (BC + nt + 15) >> 4;             // block columns of the packed width nx + 1 + nt
    auto sn_block = [&](int sid) {     // [Sn | sn] = [Q | q] + A' SW(:, 0..nx), block sid of four
      const int r0 = 16 * (sid >> 1), c0 = 16 * (sid & 1);
      v4d acc = blk_load<LDN, 32, 0>(&Qq[0][0], r0, c0, l);
      const int acol = r0 + li < NX ? r0 + li : LDW - 1;             // the last padding column of W is always zero
      double a[KS], b[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 4 * ks + lk;
        a[ks] = W[kk][acol];                                        // A'(i, kk)
        b[ks] = ws.SW[kk][c0 + li];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
      blk_store<LDN, 32>(&ws.Sn[0][0], r0, c0, l, acc);
    };
    // ---- P0 (L): the requests of this stage have landed; [Px | Pe | Pu] registers -> LDS
    if (role_d) wait_dma();
    if (role_l) pw.stage(PW);
    lds_barrier();                     // B0
    // ---- P1: SW = sym(S) W, s added to the b column: up to six blocks on C0..C3, F, E
    //      L: the requests of the next stage (their buffers were last read in P3 of the stage before this one); wave 5 keeps r~, q~
    if (role_l && k > io.k_lo) pw.prefetch(ws.nut[k - 1], ws.mode[k - 1]);                              // never beyond the chunk: earlier stages may not be projected yet
    const bool ahead = role_d && k > io.k_lo;                              // wave 4 issues the requests of stage k - 1, a third here, a third in P2, a third in P3:
    const int nt_next = k > io.k_lo ? ws.nut[k - 1] : 0;                   // all of them at once kept it ~3 k cycles and the barrier B1 waited for it (0.385 ms)
    if (ahead) issue_part(k - 1, nt_next, Part0{});
    if (w == kMWave) {
      if (l < RE) ws.r[b2][l] = M[l][NX];                                  // rows >= nt came from the zero page
      q_cur = l < NX ? Qq[l][NX] : 0.0;
    }
    if (w != 4 && w != 5) {
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
        const double smask = (c0 + li == NX) ? 1.0 : 0.0;              // s rides in the b column; rows >= nx of S are zero
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
    // ---- P2: [G | g | H] = [P | r | R] + B' SW: nbc <= 3 blocks on C0..C2 (the elimination waits for them); C3: block 3 of Sn
    if (role_c) {
      if (w < nbc) {
        const int c0 = 16 * w;
        v4d acc = blk_load<LDW, 32, 0>(&M[0][0], 0, c0, l);
        double a[KS], b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int kk = 4 * ks + lk;
          a[ks] = W[kk][BC + li];                                    // B'(i, kk); columns >= nt and rows >= nx of B~ are zero
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
    if (ahead) issue_part(k - 1, nt_next, Part1{});
    lds_barrier();                     // B2
    // ---- P3 (E): forward elimination of [H | G g] -> Z, Yn;  B3;  back substitution -> Yb (beside the chain)
    //      C0..C2, F: outputs of stage k + 1;  L4: blocks 0, 2 of Sn, L5: block 1, m of stage k + 1;  C3: nothing (shares its SIMD with E);
    //      B3;  C0..C3: [S | s] = Sn - Z' Yn
    if (role_e) {
      const int rpr = 16 - nt;
      const bool rows_layout = BPMPC_RICCATI_GJ_DPP && 4 * rpr >= NX + 1;
      const int c16 = l & 15;
      const int rid = rows_layout ? (l >> 4) * rpr + (c16 - nt) : l - nt;          // right-hand side of this lane
      const bool is_h = rows_layout ? c16 < nt : l < nt;
      const bool rhs = !is_h && rid < NX + 1;
      const bool used = is_h || rhs;
      const int col = is_h ? BC + (rows_layout ? c16 : l) : (rhs ? rid : 0);
      bool ok;
      // rows nt .. of Z and Yn up to the k-step boundary must read as zero (an earlier stage may have had more reduced inputs)
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
        /* -Y: saves the negations of the four output blocks; rows nt.. up to the k-step boundary as zeros (v is zero there) */ \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) if (rhs && i < 4 * ksn) ws.Yb[i][col] = -v[i];                       \
        if (ROWS < 4 * ksn && rhs) for (int i = ROWS; i < 4 * ksn; ++i) ws.Yb[i][col] = 0.0;                                  \
      }
      if (rows_layout) {
        if (nt <= 8) BP_GJ_CASE(8, forward_eliminate_rows, back_substitute_rows)
        else if (nt == 9) BP_GJ_CASE(9, forward_eliminate_rows, back_substitute_rows)          // single support of this robot class: 14 rows of rank 13
        else BP_GJ_CASE(10, forward_eliminate_rows, back_substitute_rows)
      } else {
        if (nt <= 12) BP_GJ_CASE(12, forward_eliminate_wave, back_substitute_wave)
        else BP_GJ_CASE(RE, forward_eliminate_wave, back_substitute_wave)
      }
#undef BP_GJ_CASE
    } else {
      if (ahead) issue_part(k - 1, nt_next, Part2{});
      if (w == 4 || w == 5) sn_block(w - 4);
      if (w == 4) sn_block(2);
      if ((w < 3 || role_f) && pend_k >= 0) finish_outputs(pend_k, pend_nt, w < 3 ? w : 3);
      if (w == kMWave && pend_k >= 0) finish_m(pend_k, q_pend);
      lds_barrier();                   // B3
      if (role_c) {
        const int r0 = 16 * (w >> 1), c0 = 16 * (w & 1);
        v4d acc = blk_load<LDN, 32, 0>(&ws.Sn[0][0], r0, c0, l);
        const int gcol = r0 + li < NX ? r0 + li : LDN - 1;              // the last padding column of Z is always zero
        double ag[4], yb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kk = 4 * ks + lk;
          ag[ks] = -ws.Zt[kk][gcol];                                    // -Z'(i, kk)
          yb[ks] = ws.Yn[kk][c0 + li];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks < ksn) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[ks], yb[ks], acc, 0, 0, 0);
        blk_store<LDN, 32>(&ws.S[0][0], r0, c0, l, acc);    // S was last read in P1, two barriers ago
      }
    }
    pend_k = k; pend_nt = nt; q_pend = q_cur;
    // no barrier: the next staging writes other buffers, and its barrier orders S, Yb and the status
  }
  __syncthreads();
  if ((w < 3 || role_f) && pend_k >= 0) finish_outputs(pend_k, pend_nt, w < 3 ? w : 3);
  if (w == kMWave && pend_k >= 0) finish_m(pend_k, q_pend);
  __syncthreads();
  if (io.k_lo > 0) {                                   // hand over to the launch that sweeps the earlier stages
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
