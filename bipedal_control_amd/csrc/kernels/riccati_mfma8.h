// Riccati sweep on the FP64 matrix cores, eight wavefronts per problem with fixed roles (HIP only; same mathematics as
// riccati.h / riccati_mfma.h, used when every problem of the batch gets a CU of its own).
//
// The sweep is one dependent chain per problem: S_k needs S_{k+1}.  riccati_mfma.h walks a stage in five barrier-separated
// phases on four wavefronts and everything a stage produces sits on that chain.  Only a part of it has to:
//     SW = sym(S) W,  [G | g | H] = [P | r | R] + B' SW,  Sn = [Q | q] + A' SW          (as before)
//     forward elimination of [H | G g]:  rows Z (pivot rows as they are eliminated) and Yn = D^-1 Z (divided by their pivots)
//     [S | s] = Sn - Z' Yn                                                                 (= Sn - G' H^-1 [G g])
// The gain Y = H^-1 [G g] (back substitution), [Acl | bcl] = [A | b] - B Y, [K | kff] = [Px | Pe] - Pu Y and m are outputs,
// not inputs of the next stage: they are finished one stage later, beside the chain, from the other half of the double
// buffered stage data.  Roles of the eight waves (w = wave index; waves w and w + 4 share a SIMD and its matrix core):
//     C  w = 0..3   one per SIMD: blocks of SW and their parts of G, the S update (C0, C1, C3); C0..C2 finish Acl, K of the previous stage while E eliminates
//     L  w = 4, 5   prefetch (global -> registers, a whole stage ahead), staging (registers -> LDS); Sn blocks beside the elimination;
//                   L5 also m of the previous stage (its store is younger than the loads it waits for next, so it delays nothing)
//     F  w = 6      third loader; a block of SW; finishes the fourth block of Acl, K
//     E  w = 7      forward elimination (on the chain) with SIMD 3 to itself - C3 idles meanwhile: the matrix core and the
//                   issue port of a SIMD are shared by its waves, and a busy neighbour doubled the elimination time -,
//                   then back substitution (beside the chain); a block of SW
// vmcnt retires in order, so a wave's wait for a load also waits for every store it issued BEFORE that load (riccati_mfma.h
// holds results in registers for a stage to get around that).  Here C and F only store, L4 only loads, and L5 stores (in P3) long before it
// prefetches again (in P1 of the next stage) and waits for that (a stage later still).
// Per stage: staging | B0 | SW and, from the accumulators, G | B2 | forward elimination (others: Sn, outputs of stage k + 1) | B3 |
// S update (E: back substitution) - three barriers, the S update needs none before the next staging barrier.
// Round 4 (0.339 -> 0.316 ms at batch 256): (1) G = [P | r | R] + B' SW had a phase of its own - a barrier, SW back from LDS, six matrix
// instructions on three waves.  The accumulator layout of v_mfma_f64_16x16x4_f64 IS its B-operand layout (lane (li, lk), register r <->
// row lk + 4 r, column li), so the wave that holds a block of SW multiplies it from the registers: block row 0 gives the state rows 0..15 of
// B' SW (added to M), block row 1 the rows 16.. (to Mb), the elimination wave adds the two parts when it loads its columns.  (2) Block (1, 0)
// of S is the mirror of (0, 1) and is neither updated nor formed in Sn (one block of Sn less beside the elimination, whose side work had
// become the longer part of its phase); sym(S) reads it from (0, 1), as the wave-per-problem sweeps do.
#pragma once
#include <hip/hip_runtime.h>

#include "riccati_mfma.h"

namespace bpmpc {

typedef unsigned int bp8_u32x2 __attribute__((ext_vector_type(2)));
constexpr int kRiccati8Threads = 512;

// LDS matrices in the swizzled layout LdsSwz (riccati_mfma.h): every operand load of the sweep, whichever of the two lane patterns it follows, meets 32
// different bank pairs per lane group.  Flat arrays; element (r, c) of a matrix with nx + 1 (+ spare) columns is at LN::ix(r, c), of a packed-width matrix at LW::ix(r, c).
template <int NJ>
struct RiccatiMfma8Workspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int RB = 32;                                  // two full 16-row blocks: plain block loads / stores
  static constexpr int RE = 16;                                  // rows of the reduced-input matrices (nut <= 16, checked)
  static constexpr int WC = NX + 1 + NU;
  using LN = LdsSwz<2>;                                          // [.. | vector] blocks: nx + 1 <= 32 columns; column 31 is never written (a zero column)
  using LW = LdsSwz<(WC + 15) / 16>;                             // packed width [A | b | B]
  static constexpr int ZN = 31;
  static_assert(NX + 1 <= 32 && NU <= 32, "two block rows / columns");
  alignas(16) double S[LN::size(RB)];        // [S | s], not symmetrised
  alignas(16) double Qq[2][LN::size(RB)];    // [Q~ | q~]
  alignas(16) double Sn[LN::size(RB)];       // [Sn | sn]
  alignas(16) double Zt[LN::size(RE)];       // pivot rows of the forward elimination of [G | g]
  alignas(16) double Yn[LN::size(RE)];       // the same rows divided by their pivots
  alignas(16) double W[2][LW::size(RB)];     // [A~ | b~ | B~]
  alignas(16) double PW[2][LW::size(RB)];    // [Px | Pe | Pu]
  alignas(16) double SW[LW::size(RB)];       // sym(S) W
  alignas(16) double M[2][LW::size(RE)];     // [P~ | r~ | R~] -> its sum with the first part of B' SW -> Y in the first nx + 1 columns
  alignas(16) double Mb[LW::size(RE)];       // the second part of B' SW (state rows 16 ..): [G | g | H] = M + Mb
  double r[2][NU];
  int status;
  unsigned char nut[kMaxRiccatiStages];
  unsigned char mode[kMaxRiccatiStages];
};
// block of the accumulator layout (lane (li, lk), register r <-> row r0 + lk + 4 r, column c0 + li) from / to a swizzled matrix
template <class L>
__device__ __forceinline__ v4d sblk_load(const double* Mx, int r0, int c0, int l) {
  v4d c;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = lds1(Mx[L::ix(r0 + (l >> 4) + 4 * r, c0 + (l & 15))]);
  return c;
}
template <class L>
__device__ __forceinline__ void sblk_store(double* Mx, int r0, int c0, int l, v4d c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) Mx[L::ix(r0 + (l >> 4) + 4 * r, c0 + (l & 15))] = c[r];
}

// Forward elimination of [H | G g], one column per lane (layout of gauss_jordan_wave).  After step p row p is divided by its
// pivot and eliminated from the rows below; emit(p, z, y) sees the pivot row before (z) and after (y) the division.
template <int ROWS, class Emit>
__device__ __forceinline__ bool forward_eliminate_wave(double (&v)[ROWS], int nt, Emit&& emit) {
  bool ok = true;
#pragma unroll
  for (int p = 0; p < ROWS; ++p) {
    if (p < nt) {  // wave-uniform
      double f[ROWS];
#pragma unroll
      for (int i = p; i < ROWS; ++i) f[i] = readlane_f64(v[i], p);
      ok = ok && (f[p] > 0.0);
      const double row = v[p] * fast_reciprocal(f[p]);
      emit(p, v[p], row);
#pragma unroll
      for (int i = p + 1; i < ROWS; ++i) v[i] -= f[i] * row;
      v[p] = row;
    }
  }
  return ok;
}
// Back substitution on the rows left by forward_eliminate_wave (unit upper triangular in the H lanes): v <- H^-1 [.. | G g].
template <int ROWS>
__device__ __forceinline__ void back_substitute_wave(double (&v)[ROWS], int nt) {
#pragma unroll
  for (int p = ROWS - 1; p >= 1; --p) {
    if (p < nt) {
#pragma unroll
      for (int i = 0; i < p; ++i) {
        const double u = readlane_f64(v[i], p);
        v[i] -= u * v[p];
      }
    }
  }
}

template <int NJ, bool JW = true>     // JW: Wt holds its joint rows (off: the loaders complete them from Vt, PackedStageLoader JR)
__device__ __forceinline__ void riccati_mfma8(RiccatiMfma8Workspace<NJ>& ws, const RiccatiFastIO& io) {
  using WS = RiccatiMfma8Workspace<NJ>;
  using LN = typename WS::LN; using LW = typename WS::LW;
  constexpr int NX = WS::NX, NU = WS::NU, NT = kRiccati8Threads, RE = WS::RE, ZN = WS::ZN;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  constexpr int KS = (NX + 3) / 4;          // k-steps over the state dimension
  constexpr int BC = NX + 1;                // first column of B~ / Pu / R~ in the packed layouts
  static_assert(NX == NU, "packed layouts assume nx == nu");
  static_assert(NX + 1 + RE <= kWave, "one lane per column of [H | G g]");
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int li = l & 15, lk = l >> 4;       // operand row/column index and k index of this lane
  const int N = io.base.N;
  constexpr int kLoaders = 3;          // loader waves: L4, L5, F (a fourth, C3, was measured in round 5: no gain)
  const bool role_c = w < 4, role_l = w >= 4 && w < 4 + kLoaders, role_f = w == 6, role_e = w == 7;

  const int k_top = (io.k_hi < N ? io.k_hi : N) - 1;
  const bool resumed = io.k_hi < N;
  {
    double* z = &ws.S[0];
    constexpr int total = (int)(offsetof(WS, status) / sizeof(double));
    for (int idx = tid; idx < total; idx += NT) z[idx] = 0.0;     // every matrix and its padding
  }
  __syncthreads();
  if (tid == 0) ws.status = resumed ? (int)io.carry[NXX + NX] : 0;
  if (!resumed && io.reg != 0.0 && tid < NX) ws.S[LN::ix(tid, tid)] = io.reg;
  if (resumed) {
    for (int idx = tid; idx < NXX; idx += NT) ws.S[LN::ix(idx / NX, idx % NX)] = io.carry[idx];
    if (tid < NX) ws.S[LN::ix(tid, NX)] = io.carry[NXX + tid];
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
    // the roll-out, which normally opens the line search of this problem, is skipped: open it here, otherwise done / alpha / base keep the
    // previous iteration's values and k_ls_decide would skip the problem instead of reporting the failure (advisor r02)
    if (io.k_lo == 0 && io.with_ls && tid < kWave) linesearch_begin_wave<NJ>(&ws.S[0], io.ls, tid);
    return;
  }

  // Prefetch registers and staging of the loader waves (PackedStageLoader, riccati_mfma.h): 128 threads, pairs t, t + 128, ..
  constexpr int NLD = kLoaders * kWave;
  PackedStageLoader<NJ, NLD, RE, LW::kCols, LN::kCols, true, !JW, LW, LN> ld;
  ld.init(io, (w - 4) * kWave + l, role_l, (size_t)(k_top > 0 ? k_top : 0));
  if (role_l && k_top >= io.k_lo) {     // the stage the loader's pointers stand on (the LDS copies of nut and mode may not be visible yet)
    const int kt = k_top > 0 ? k_top : 0, n0 = io.base.nut[kt];
    ld.prefetch(n0, n0 > 0 ? (io.mode[kt] & 3) : kModeEvent);
  }
  __syncthreads();
#ifdef BPMPC_RICCATI_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define RM8PROF(slot) do { const long long tn_ = clock64(); tacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#define RM8OWN(phase) do { if (BPMPC_RICCATI_PROFILE == 10 + (phase)) tacc[7] += clock64() - tprev; } while (0)   /* own work of a phase, before its barrier */
#else
#define RM8PROF(slot) ((void)0)
#define RM8OWN(phase) ((void)0)
#endif

  // Lane parts of the swizzled addresses (element offsets, formed once: an access is lane part + a constant that fits the instruction's offset field).
  //   pattern (2): rows R + lk, columns C + li (R a multiple of 4, C of 16): (R / 2) PS + 2 C + p2[(C / 16) & 1]
  //   pattern (1): rows R + li, columns C + lk (R a multiple of 16, C of 4, no chunk boundary inside C .. C + 3): (R / 2) PS + 2 (C & ~15) + (C & 15) + p1[(C / 16) & 1]
  constexpr int PSN = LN::PS, PSW = LW::PS;
  const int hk = lk >> 1, pk = lk & 1, hl = li >> 1, pl = li & 1;
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  const int nP2a = opaque(hk * PSN + 16 * pk + li), nP2b = opaque(hk * PSN + 16 * (pk ^ 1) + li);     // LN matrices, pattern (2), even / odd block column
  const int wP2a = opaque(hk * PSW + 16 * pk + li), wP2b = opaque(hk * PSW + 16 * (pk ^ 1) + li);     // LW matrices
  const int nP1a = opaque(hl * PSN + 16 * pl + lk), nP1b = opaque(hl * PSN + 16 * (pl ^ 1) + lk);     // LN matrices, pattern (1)
  const int nSV = opaque(hk * PSN + 16 * (pk ^ 1) + 32 + (NX & 15));                                   // column nx (s) of rows R + lk
  static_assert(NX >= 16 && NX < 32, "column nx lies in block column 1");
  int wBC;                                                                                             // rows R + lk, columns BC + li of an LW matrix
  { const int c = BC + li, cb = c >> 4; wBC = opaque(hk * PSW + 32 * cb + 16 * ((pk ^ cb) & 1) + (c & 15)); }
  // per wave, fixed for the sweep: the output block bw (C0..C2: 0..2, F: 3), the Sn block of L4 / L5 (sid = w - 4: block row 0), block 3 of Sn, the S-update block (blk = w)
  int foW, foY, foB[4], snN, snP, wA3, suN, suZ, suY;
  {
    const int bw = w < 3 ? w : 3, r0 = 16 * (bw >> 1), c0 = 16 * (bw & 1);
    foW = opaque((r0 / 2) * PSW + 2 * c0 + ((bw & 1) ? wP2b : wP2a));
    foY = opaque(2 * c0 + ((bw & 1) ? wP2b : wP2a));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foB[ks] = opaque(LW::ix(r0 + li, BC + 4 * ks + lk));
    const int sid = w & 1;                                   // (w = 4, 5)
    snN = opaque(32 * sid + (sid ? nP2b : nP2a));
    snP = opaque(32 * sid + (sid ? wP2b : wP2a));
    const int ac3 = 16 + li < NX ? 16 + li : NX - 1;         // block row 1 of A': rows >= nx do not exist - those lanes read a column that does (and multiply it by zero)
    wA3 = opaque(hk * PSW + 32 + 16 * (pk ^ 1) + (ac3 & 15));
    const int blk = w & 3, ur0 = 16 * (blk >> 1), uc0 = 16 * (blk & 1);
    suN = opaque((ur0 / 2) * PSN + 2 * uc0 + ((blk & 1) ? nP2b : nP2a));
    suY = opaque(2 * uc0 + ((blk & 1) ? nP2b : nP2a));
    const int gcol = ur0 + li < NX ? ur0 + li : ZN;           // column 31 of Z is never written: always zero
    suZ = opaque(hk * PSN + 32 * (gcol >> 4) + 16 * ((pk ^ (gcol >> 4)) & 1) + (gcol & 15));
  }
  const double amask3 = 16 + li < NX ? 1.0 : 0.0;

  // Outputs of a stage that are not on the chain (block bw of each): [Acl | bcl] = [A | b] - B Y, [K | kff] = [Px | Pe] - Pu Y,
  // from the buffer set `buf` that stage was staged into.  A wave always forms the same block: C0..C2 the blocks 0..2, F block 3.
  constexpr unsigned kOob = 0x80000000u;                       // beyond num_records of the resources below
  auto out_rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); };
  const __amdgpu_buffer_rsrc_t rAcl = out_rsrc(io.Acl), rKf = out_rsrc(io.Kfull), rbcl = out_rsrc(io.bcl), rkff = out_rsrc(io.kff);
  unsigned oom[4], oov[4];
  {
    const int bw = w < 3 ? w : 3, r0 = 16 * (bw >> 1), col = 16 * (bw & 1) + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = r0 + lk + 4 * r;
      oom[r] = (rr < NX && col < NX) ? (unsigned)(rr * NX + col) * 8u : kOob;
      oov[r] = (rr < NX && col == NX) ? (unsigned)rr * 8u : kOob;
    }
  }
  auto finish_outputs = [&](int k, int buf, int nt, int bw) {
    const double* const W = ws.W[buf];
    const double* const PW = ws.PW[buf];
    const double* const M = ws.M[buf];
    const int ksn = (nt + 3) >> 2;
    const int c0 = 16 * (bw & 1);
    v4d acl, kf;
#pragma unroll
    for (int r = 0; r < 4; ++r) { acl[r] = lds1(W[foW + 2 * r * PSW]); kf[r] = lds1(PW[foW + 2 * r * PSW]); }      // rows r0 + lk + 4 r, columns c0 + li
    double ab[4], ap[4], yb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      yb[ks] = lds1(M[foY + 2 * ks * PSW]);                      // -Y(4 ks + lk, c0 + li) (E stores the gain negated); rows >= nt are zero
      ab[ks] = lds1(W[foB[ks]]);                                 // B(i, kk) = W(r0 + li, BC + 4 ks + lk); rows >= nx of W and PW are zero
      ap[ks] = lds1(PW[foB[ks]]);                                // Pu(i, kk)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < ksn) {                                            // wave-uniform
        acl = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[ks], yb[ks], acl, 0, 0, 0);
        kf = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[ks], yb[ks], kf, 0, 0, 0);
      }
    }
    // buffer stores: where a register of the block goes is a byte offset that is fixed for the whole sweep (Acl and K share it; column nx goes to
    // bcl / kff), an element that goes nowhere has an offset beyond the resource, the stage is the scalar offset of the instruction.  (As predicated
    // plain stores with 64-bit addresses the eight stores of a block were most of the 2.2 k cycles a block took beside the elimination.)
    const unsigned ku = (unsigned)__builtin_amdgcn_readfirstlane(k);
    const unsigned sm = ku * (unsigned)(NXX * 8), sv = ku * (unsigned)(NX * 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bp8_u32x2 va, vk;
      va.x = (unsigned)__double2loint(acl[r]); va.y = (unsigned)__double2hiint(acl[r]);
      vk.x = (unsigned)__double2loint(kf[r]); vk.y = (unsigned)__double2hiint(kf[r]);
      __builtin_amdgcn_raw_buffer_store_b64(va, rAcl, oom[r], sm, 0);
      __builtin_amdgcn_raw_buffer_store_b64(vk, rKf, oom[r], sm, 0);
      if (c0 != 0) {                                             // wave-uniform: the block column that holds column nx
        __builtin_amdgcn_raw_buffer_store_b64(va, rbcl, oov[r], sv, 0);
        __builtin_amdgcn_raw_buffer_store_b64(vk, rkff, oov[r], sv, 0);
      }
    }
  };

  // m = q~ - Y' r~, m0 = -r~' H^-1 g of a finished stage (one wave).  Rows >= nt of Y and of r~ are zero (the projection kernel pads
  // with zeros): all loads first, no branch per row.
  auto finish_m = [&](int k, int buf) {
    if (l <= NX) {
      const int fm0 = 32 * (l >> 4) + 16 * ((l >> 4) & 1) + (l & 15);      // column l of an even row of an LW matrix (odd rows: bit 4 flipped)
      double yv[RE], rv[RE];
#pragma unroll
      for (int i = 0; i < RE; ++i) { yv[i] = ws.M[buf][(i >> 1) * PSW + ((i & 1) ? (fm0 ^ 16) : fm0)]; rv[i] = ws.r[buf][i]; }
      double m0 = l < NX ? ws.Qq[buf][LN::ix(l, NX)] : 0.0, m1 = 0.0;
#pragma unroll
      for (int i = 0; i < RE; i += 2) { m0 += yv[i] * rv[i]; m1 += yv[i + 1] * rv[i + 1]; }     // yv: -Y
      if (l < NX) io.mvec[(size_t)k * NX + l] = m0 + m1; else io.mscal[k] = m0 + m1;
    }
  };

  int pend_k = -1, pend_nt = 0;          // stage whose outputs are still to be finished (uniform)
  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = ws.nut[k];            // max_nodes <= kMaxRiccatiStages is checked when the solver is created
    const int cur = k & 1;
    double* const W = ws.W[cur];
    double* const PW = ws.PW[cur];
    double* const Qq = ws.Qq[cur];
    double* const M = ws.M[cur];
    double* const rvec = ws.r[cur];
    const int ksn = (nt + 3) >> 2;                   // k-steps over the reduced input
    const int nbc = (BC + nt + 15) >> 4;             // block columns of the packed width nx + 1 + nt
    // [Sn | sn] = [Q | q] + A' SW(:, 0..nx), one block of four: TOP: block row 0 (block column w - 4 on L4 / L5), else block (1, 1); block (1, 0) is the mirror of (0, 1)
    auto sn_block = [&](auto top_c) {
      constexpr bool TOP = decltype(top_c)::value;
      const int nb = TOP ? snN : 8 * PSN + 32 + nP2b;                // rows r0 + lk + 4 r, columns c0 + li of Qq / Sn
      const int wa = TOP ? wP2a : wA3;                                // A'(i, kk) = W(4 ks + lk, r0 + li)
      const int wb = TOP ? snP : 32 + wP2b;                           // SW(4 ks + lk, c0 + li)
      v4d acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = lds1(Qq[nb + 2 * r * PSN]);
      double a[KS], b[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a[ks] = lds1(W[wa + 2 * ks * PSW]);
        b[ks] = lds1(ws.SW[wb + 2 * ks * PSW]);
      }
      if constexpr (!TOP) {          // rows >= nx of A' do not exist: their lanes have read a column that does (finite numbers; x 1.0 is exact)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] *= amask3;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) ws.Sn[nb + 2 * r * PSN] = acc[r];
    };
    // ---- P0 (L): registers -> packed LDS layouts; what the projection kernel does not write (block columns >= nbc, rows >= nt) is staged as zero
    if (role_l) ld.stage(W, PW, Qq, M, rvec, nt);
    RM8OWN(0);
    lds_barrier();                     // B0
    RM8PROF(0);
    // ---- P1: SW = sym(S) W, s added to the b column: up to six blocks on C0..C3, F, E (L: the loads of the next stage); and from each block,
    //      while it is still in the accumulators (their layout is the B-operand layout: lane (li, lk), register r <-> SW[r0 + lk + 4 r][c0 + li]),
    //      its part of [G | g | H](:, bj) = [P | r | R] + B' SW(:, bj): state rows 0..15 (block row 0, added to M) or 16.. (block row 1, to Mb).
    //      Round 3 had a phase of its own for G (a barrier, SW back from LDS, six matrix instructions on three waves); the elimination adds the two parts.
    if (role_l && k > io.k_lo) ld.prefetch(ws.nut[k - 1], ws.mode[k - 1]);     // never beyond the chunk: earlier stages may not be projected yet
    if (w != 4 && w != 5) {
      const int id = w < 4 ? w : w - 2;
      if (id < 2 * nbc) {
        const int bi = id >= nbc ? 1 : 0;
        const int r0 = 16 * bi, c0 = 16 * (id - bi * nbc);
        const int row = r0 + li;
        const double half = row < NX ? 0.5 : 0.0;
        constexpr int KG1 = KS - 4;                                      // k-steps of the second block row (state rows 16 .. nx - 1)
        const int odd = (c0 >> 4) & 1;
        const int wb = 2 * c0 + (odd ? wP2b : wP2a);                      // rows 4 ks + lk (or lk + 4 r), columns c0 + li of W, M, Mb, SW
        const int sx0 = 8 * bi * PSN + nP1a, sx1 = 8 * bi * PSN + 32 + nP1b;     // S(row, 4 ks + lk): ks < 4 / ks >= 4
        const int sy = 32 * bi + (bi ? nP2b : nP2a);                      // S(4 ks + lk, row)
        double a[KS], b[KS], sv[4], ga[4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          // sym(S): the mean of the two triangles inside the diagonal blocks; block (1, 0) of S is never computed (the S update leaves it out, as
          // the wave-per-problem sweeps do) - its elements are read from block (0, 1): both terms of the mean are then the same element
          const bool up = (bi == 0 && ks >= 4), lo = (bi != 0 && ks < 4);
          const int ex = (ks < 4 ? sx0 : sx1) + 4 * (ks & 3), ey = sy + 2 * ks * PSN;
          a[ks] = half * (lds1(ws.S[lo ? ey : ex]) + lds1(ws.S[up ? ex : ey]));
          b[ks] = lds1(W[wb + 2 * ks * PSW]);
        }
        const double smask = (c0 + li == NX) ? 1.0 : 0.0;              // s rides in the b column; rows >= nx of S are zero
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = smask * lds1(ws.S[nSV + (8 * bi + 2 * r) * PSN]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ga[ks] = (bi == 0 || ks < KG1) ? lds1(W[wBC + (8 * bi + 2 * ks) * PSW]) : 0.0;     // B'(i, kk) = W(r0 + 4 ks + lk, BC + li); columns >= nt and rows >= nx of B~ are zero
        v4d g;
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = lds1(M[wb + 2 * r * PSW]);
        if (bi != 0) g = v4d{0.0, 0.0, 0.0, 0.0};
        __builtin_amdgcn_sched_barrier(0);
        v4d acc = {sv[0], sv[1], sv[2], sv[3]};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) ws.SW[wb + (8 * bi + 2 * r) * PSW] = acc[r];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (bi == 0 || ks < KG1) g = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[ks], acc[ks], g, 0, 0, 0);
        double* const gd = bi == 0 ? M : ws.Mb;
#pragma unroll
        for (int r = 0; r < 4; ++r) gd[wb + 2 * r * PSW] = g[r];
      }
    }
    RM8OWN(1);
    lds_barrier();                     // B2 (there is no B1 any more)
    RM8PROF(1);
    RM8PROF(2);
    // ---- P3 (E): forward elimination of [H | G g] -> Z, Yn;  B3;  back substitution -> Y (beside the chain)
    //      C0..C2, F: outputs of stage k + 1;  L4: blocks 0, 2 of Sn, L5: block 1;  C3: nothing (shares its SIMD with E);  B3;
    //      C0..C3: [S | s] = Sn - Z' Yn
    if (role_e) {
      // lane layout: with at most 4 (16 - nt) >= nx + 1 right-hand sides the DPP row layout of riccati_fast.h (every 16-lane row
      // holds H in its lanes 0..nt-1 and 16 - nt right-hand sides), otherwise one column per lane of the wave (v_readlane)
      const int rpr = 16 - nt;
      const bool rows_layout = 4 * rpr >= NX + 1;
      const int c16 = l & 15;
      const int rid = rows_layout ? (l >> 4) * rpr + (c16 - nt) : l - nt;          // right-hand side of this lane
      const bool is_h = rows_layout ? c16 < nt : l < nt;
      const bool rhs = !is_h && rid < NX + 1;
      const bool used = is_h || rhs;
      const int col = is_h ? BC + (rows_layout ? c16 : l) : (rhs ? rid : 0);
      bool ok;
      // rows nt .. of Z and Yn up to the k-step boundary must read as zero (an earlier stage may have had more reduced inputs)
      if (rhs) {
        for (int i = nt; i < 4 * ksn; ++i) { ws.Zt[LN::ix(i, col)] = 0.0; ws.Yn[LN::ix(i, col)] = 0.0; }
      }
      // every lane stores, the ones without a right-hand side into four spare columns (nx + 2 ..: never read as Z; as Yn they only make
      // columns > nx + 1 of S, which no product reads): a lane predicate here is an s_and_saveexec / branch pair per pivot on the critical wave
      static_assert(NX + 2 + 3 < ZN && 4 * KS <= NX + 2, "spare columns of Z / Yn");
      const int ecol = rhs ? col : NX + 2 + (l & 3);
      // element (i, column of this lane): the lane part for even rows; odd rows exchange the halves of the chunk pair (bit 4 - the lane parts are below 32 + 16 + 16)
      const int eM0 = 32 * (col >> 4) + 16 * ((col >> 4) & 1) + (col & 15), eM1 = eM0 ^ 16;
      const int eN0 = 32 * (ecol >> 4) + 16 * ((ecol >> 4) & 1) + (ecol & 15), eN1 = eN0 ^ 16;
      auto emit = [&](int p, double z, double y) { const int at = (p >> 1) * PSN + ((p & 1) ? eN1 : eN0); ws.Zt[at] = z; ws.Yn[at] = y; };
#define BP_EM(i) (((i) >> 1) * PSW + (((i) & 1) ? eM1 : eM0))
#define BP_GJ_CASE(ROWS, FWD, BWD)                                                            \
      {                                                                                       \
        double v[ROWS];                                                                       \
        {   /* both parts of [H | G g].  The loads are pinned (as conditional sums they had become ten load - wait - add branches).  nx = 24: the second \
               part arrives in two groups - 15 instead of 20 values in flight, or the kernel needs scratch -, which starts the elimination an LDS round trip later */ \
          constexpr int H1 = NJ <= 10 ? ROWS : (ROWS + 1) / 2;                                \
          double ta[ROWS], tb[H1];                                                            \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i) ta[i] = lds1(M[BP_EM(i)]);         \
          _Pragma("unroll") for (int i = 0; i < H1; ++i) tb[i] = lds1(ws.Mb[BP_EM(i)]);       \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i) asm volatile("" : "+v"(ta[i]));    \
          _Pragma("unroll") for (int i = 0; i < H1; ++i) asm volatile("" : "+v"(tb[i]));      \
          /* (no masks: rows >= nt of M and Mb are zero - they are staged as zeros and B~ has no columns there -, and a lane without a column eliminates column 0 into a spare column) */ \
          _Pragma("unroll") for (int i = 0; i < H1; ++i) v[i] = ta[i] + tb[i];              \
          if constexpr (H1 < ROWS) {                                                          \
            _Pragma("unroll") for (int i = H1; i < ROWS; ++i) tb[i - H1] = ws.Mb[BP_EM(i)];       \
            _Pragma("unroll") for (int i = H1; i < ROWS; ++i) asm volatile("" : "+v"(tb[i - H1])); \
            _Pragma("unroll") for (int i = H1; i < ROWS; ++i) v[i] = ta[i] + tb[i - H1];     \
          }                                                                                   \
        }                                                                                     \
        ok = FWD(v, nt, emit);                                                          \
        if (l == 0 && !ok) ws.status = 1;                                                     \
        RM8PROF(6);                                                                           \
        lds_barrier();                 /* B3 */                                               \
        RM8PROF(3);                                                                           \
        BWD<ROWS>(v, nt);                                                                     \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) if (rhs && i < nt) M[BP_EM(i)] = -v[i];    /* -Y: saves the negations of the four output blocks */ \
      }
      // the elimination is the longest dependent chain of a stage: instantiate it for the actual number of rows
      if (rows_layout) {
        if (nt <= 8) BP_GJ_CASE(8, forward_eliminate_rows<8>, back_substitute_rows)
        else if (nt == 9) BP_GJ_CASE(9, (forward_eliminate_rows<9, true>), back_substitute_rows)          // single support of this robot class: 14 rows of rank 13
        else BP_GJ_CASE(10, (forward_eliminate_rows<10, true>), back_substitute_rows)
      } else {
        if (nt <= 12) BP_GJ_CASE(12, forward_eliminate_wave<12>, back_substitute_wave)
        else BP_GJ_CASE(RE, forward_eliminate_wave<RE>, back_substitute_wave)
      }
#undef BP_GJ_CASE
#undef BP_EM
    } else {
      if (w == 4 || w == 5) sn_block(std::true_type{});
      if (w == 4) sn_block(std::false_type{});         // (not C3: a block of Sn beside E on SIMD 3 slowed the elimination from 2290 to 2720 cycles; block 2 = (1, 0) is not needed)
      if (w == 5 && pend_k >= 0) finish_m(pend_k, cur ^ 1);      // (not E after its back substitution: the staging barrier waited for it)
      if ((w < 3 || role_f) && pend_k >= 0) finish_outputs(pend_k, cur ^ 1, pend_nt, w < 3 ? w : 3);
      RM8PROF(6);
      lds_barrier();                   // B3
      RM8PROF(3);
      if (role_c && w != 2) {          // block (1, 0) of S is the mirror of (0, 1): nobody reads it
        v4d acc;                                                         // block blk = w: rows r0 + lk + 4 r, columns c0 + li of Sn
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = lds1(ws.Sn[suN + 2 * r * PSN]);
        double ag[4], yb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          ag[ks] = -lds1(ws.Zt[suZ + 2 * ks * PSN]);                          // -Z'(i, kk) = -Zt(4 ks + lk, r0 + li)
          yb[ks] = lds1(ws.Yn[suY + 2 * ks * PSN]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks < ksn) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[ks], yb[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) ws.S[suN + 2 * r * PSN] = acc[r];      // S was last read in P1, two barriers ago
      }
    }
    pend_k = k; pend_nt = nt;
    RM8OWN(4);
    RM8PROF(4);
    // no barrier: the next staging writes the other buffer set, and its barrier orders S, Y and the status
  }
#ifdef BPMPC_RICCATI_PROFILE
#if BPMPC_RICCATI_PROFILE == 2      // own work of every wave between B2 and B3
  if (io.prof && l == 0) io.prof[w] = (double)tacc[6];
#elif BPMPC_RICCATI_PROFILE >= 10   // own work of every wave in phase BPMPC_RICCATI_PROFILE - 10 (0: staging, 1: SW, 2: G, 4: S update incl. E's back substitution)
  if (io.prof && l == 0) io.prof[w] = (double)tacc[7];
#else
  if (io.prof && tid == 0)
    for (int i = 0; i < 5; ++i) io.prof[i] = (double)tacc[i];     // [5]: roll-out + [6]: step norms, written behind the sweep
  if (io.prof && tid == 7 * kWave) io.prof[7] = (double)tacc[6];
#endif
#endif
  __syncthreads();
  if ((w < 3 || role_f) && pend_k >= 0) finish_outputs(pend_k, pend_k & 1, pend_nt, w < 3 ? w : 3);
  if (w == 5 && pend_k >= 0) finish_m(pend_k, pend_k & 1);
  __syncthreads();
  if (io.k_lo > 0) {                                   // hand over to the launch that sweeps the earlier stages
    for (int idx = tid; idx < NXX; idx += NT) { const int r = idx / NX, c = idx % NX; io.carry[idx] = (r >= 16 && c < 16) ? ws.S[LN::ix(c, r)] : ws.S[LN::ix(r, c)]; }
    if (tid < NX) io.carry[NXX + tid] = ws.S[LN::ix(tid, NX)];
    if (tid == 0) io.carry[NXX + NX] = (double)ws.status;
    return;
  }
  {
    const int st = ws.status;
    __syncthreads();                                   // the workspace is dead from here on: it holds the state history
#ifdef BPMPC_RICCATI_PROFILE
    const long long tr0 = clock64();
#endif
    constexpr int kRingChunk = 4;                      // stages per chunk of the roll-out's ring (six: 20 stages less history, no faster)
    using RR = RolloutRing<NJ, NT, kRingChunk>;
    constexpr int kRingCap = RR::cap((int)(offsetof(WS, status) / sizeof(double)));
    static_assert(kRingCap >= 128, "roll-out history");
    riccati_rollout_ring<NJ, NT, kRingChunk>(reinterpret_cast<double*>(&ws), kRingCap, st, io);
#ifdef BPMPC_RICCATI_PROFILE
    if (BPMPC_RICCATI_PROFILE == 1 && io.prof && tid == 0) io.prof[5] = (double)(clock64() - tr0);     // the roll-out behind the sweep, whole horizon ([6]: its recurrence alone)
#endif
  }
}

}  // namespace bpmpc
