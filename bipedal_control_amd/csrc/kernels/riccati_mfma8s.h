// Riccati sweep, eight wavefronts per problem, the value function never leaves the registers between two stages (HIP only; same
// mathematics and the same roles as riccati_mfma8.h, bit-identical results; nx = 22: three sets of stage data have to fit the LDS).
//
// riccati_mfma8.h walks a stage in four steps on the chain: S update | staging | S W, G | forward elimination - three LDS-only
// barriers, and between the S update and S W the new S makes a round trip through the LDS (4 stores, a barrier, 12 loads, the
// symmetrisation).  Two facts remove that round trip, the S-update phase and the staging phase:
//   (1) the accumulator layout of v_mfma_f64_16x16x4_f64 (lane (li, lk), register r  <->  row lk + 4 r, column li) is the A-operand
//       layout of the TRANSPOSED block: register ks of the block X = S(kb, bi) (rows of block row kb, columns of block row bi) is
//       the k-step ks of the operand sym(S)(bi, kb) - S is symmetric.  So the wave that forms the block (bi, bj) of S W computes the
//       blocks of S = Sn - Z' Yn it needs itself, straight into its operand registers:
//           block row 0:  X1 = S(0,0), X2 = S(0,0)' (the same products with the operands exchanged: exactly the transposed block,
//                         0.5 (X1 + X2) is the symmetrised diagonal block of riccati_mfma8.h bit for bit), XT = S(0,1)'
//           block row 1:  X01 = S(0,1) (block (1,0) is its mirror, as before), X1 = S(1,1), X2 = S(1,1)'
//       nine matrix instructions instead of three, but no S in the LDS, no barrier, no loads of S, no symmetrisation arithmetic; s
//       (column nx of S) is column nx - 16 of S(1,1) / S(0,1) in exactly the lanes that add it to the b column of S W (block row 0
//       fetches its sixteen values from XT through a 128-byte LDS slot of its own).
//   (2) with a third set of stage data in the LDS the loaders stage stage k - 1 while the chain waves work on stage k: nothing of
//       the staging is between two phases of the chain.
// Per stage two phases, two barriers:
//     C phase   C0..C3: S blocks -> S W (stored for Sn) -> G parts (M, Mb);   E: back substitution of the stage before (its rows stayed
//               in registers) -> -Y;   L4, L5, F: registers -> LDS of stage k - 1 (third set), requests of stage k - 2
//     E phase   E: forward elimination of [H | G g] -> Z, Yn;   L4, L5: Sn = Q + A' S W (blocks (0,0), (0,1), (1,1)), m of stage k + 1;
//               C0..C2, F: [Acl bcl], [K kff] of stage k + 1;   C3 idles (it shares its SIMD with E)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "riccati_mfma8.h"

namespace bpmpc {

#ifndef BPMPC_RS8_ABLATE
#define BPMPC_RS8_ABLATE 0      // timing experiments (wrong results): bit 1: S = Sn (no S blocks formed), bit 2: no G, bit 3: no S W
#endif

template <int NJ>
struct RiccatiMfma8sWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int RB = 32;                                  // rows of the matrices that are written as whole blocks
  static constexpr int RW = NX % 4 == 0 ? NX + 2 : ((NX + 3) / 4) * 4;   // rows of the staged operands: nx and at least one row of zeros (block rows beyond are read there)
  static constexpr int ZR = RW - 1;
  static constexpr int RE = 16;
  static constexpr int LDN = 34;
  static constexpr int WC = NX + 1 + NU;
  static constexpr int LDW = ((WC + 15) / 16) * 16 + 2;
  static constexpr int NSET = 3;
  static_assert(NX + 1 <= 32 && NU <= 32 && ((NX + 3) / 4) * 4 <= RW, "two block rows / columns; every k-step stays inside the rows");
  alignas(16) double Sn[RB][LDN];           // [Sn | sn]; before the first stage: the terminal [S | s]; after the last one of a chunk: [S | s] for the hand-over
  alignas(16) double Zt[RE][LDN];
  alignas(16) double Yn[RE][LDN];
  alignas(16) double SW[RB][LDW];
  alignas(16) double Mb[RE][LDW];
  alignas(16) double W[NSET][RW][LDW];
  alignas(16) double PW[NSET][RW][LDW];
  alignas(16) double M[NSET][RE][LDW];
  alignas(16) double sx[4][32];             // per chain wave: s of block row 0 on its way from XT to the accumulator lanes
  alignas(16) double rm[16];                // r~ of the stage whose m is formed, on its way to every lane of that wave
  int status;
  unsigned char nut[kMaxRiccatiStages];
  unsigned char mode[kMaxRiccatiStages];
};

// block load of the accumulator layout, transposed: lane (li, lk), register r  <-  Mx[r0 + li][c0 + lk + 4 r]
template <int LD>
__device__ __forceinline__ v4d blk_load_t(const double* Mx, int r0, int c0, int l) {
  v4d c;
  const double* p = Mx + (r0 + (l & 15)) * LD + c0 + (l >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = p[4 * r];
  return c;
}


// Stage loader of this sweep: the streams of PackedStageLoader / PwVtLoader (riccati_mfma.h: the same pairs, the same LDS offsets, the same values),
// requested with BUFFER loads.  A loader wave of riccati_mfma8.h spent 1.8 k cycles per stage on its thirteen requests: 64-bit pointer selects per
// slot (a pair that the projection kernel does not write is masked by the address of its request), 64-bit pointer decrements per stream.  Here a
// stream is a buffer resource (scalar registers), the stage is the scalar offset of the instruction, a slot is a 32-bit byte offset that never
// changes, and a masked slot is an offset beyond the resource: the hardware returns zeros for it.
typedef unsigned int bp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bp_u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kBufOob = 0x80000000u;                  // beyond num_records of every resource below
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bp_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void bp_buf_pair(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double& x, double& y) {
  const bp_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  x = __hiloint2double((int)v.y, (int)v.x); y = __hiloint2double((int)v.w, (int)v.z);
}
__device__ __forceinline__ double bp_buf_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const bp_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}

template <int NJ, int NLD, int LDW, bool JR>
struct BufStageLoader {
  using PL = PackedLq<NJ>;
  static constexpr int NX = PL::NX, NU = PL::NU, WP = PL::WP, QP = PL::QP, BC = NX + 1;
  static constexpr int HW = WP / 2, HQ = QP / 2, HQU = (NX + 2) / 2;
  static constexpr int NPW = (JR ? 12 : NX) * HW, NPV = NJ * HW, NPF = 12 * HW;
  static constexpr int SW = (NPW + NLD - 1) / NLD, SV = (NPV + NLD - 1) / NLD, SF = (NPF + NLD - 1) / NLD;
  static_assert(LDW % 2 == 0 && NX % 2 == 0 && WP <= LDW, "pairs stay aligned and inside the rows");
  // what a request returns stays in the registers it arrived in until it is staged (taken apart into doubles at request time, every request
  // was followed by its own s_waitcnt and a handful of register moves: the requests of a stage went out one memory round trip after the other)
  bp_u32x4 w4[SW], v4[SV];
  bp_u32x2 pe2[SF], jb2[SV];
  double dtk;
  int fmode;                                     // contact mode code of the requested stage (force rows are generated when they are staged)
  const double *gW, *gV, *gPe, *gB, *gDt;
  // per slot, fixed for the whole sweep: byte offset of the pair inside the node's block (kBufOob: no pair), LDS element offset (-1: none), column, row
  unsigned wv[SW], vv[SV], pev[SF], jbv[SV];
  int wo[SW], wc[SW], vo[SV], vc[SV], fo[SF], ux[SF], uy[SF];
  int jid[SV];                                   // bit 0 / 1: the first / second value of the pair sits on the diagonal of its joint row

  __device__ __forceinline__ void init(const RiccatiFastIO& io, int tl, bool loader) {
    const int tp = loader ? tl : 0;
    gW = io.Wt; gV = io.Vt; gPe = io.base.Pe; gB = io.lqb; gDt = io.gdt; dtk = 0.0; fmode = 0;
#pragma unroll
    for (int e = 0; e < SW; ++e) { const int p = tp + e * NLD; const bool ok = loader && p < NPW; wv[e] = ok ? (unsigned)p * 16u : kBufOob; wo[e] = ok ? (p / HW) * LDW + 2 * (p % HW) : -1; wc[e] = 2 * (p % HW); }
#pragma unroll
    for (int e = 0; e < SV; ++e) {
      const int p = tp + e * NLD, row = 12 + p / HW, col = 2 * (p % HW);
      const bool ok = loader && p < NPV;
      vv[e] = ok ? (unsigned)p * 16u : kBufOob; vo[e] = ok ? row * LDW + col : -1; vc[e] = col;
      jid[e] = ((ok && col == row) ? 1 : 0) | ((ok && col + 1 == row) ? 2 : 0);
      jbv[e] = (JR && ok && col == NX) ? (unsigned)row * 8u : kBufOob;
    }
#pragma unroll
    for (int e = 0; e < SF; ++e) {
      const int p = tp + e * NLD, c = p / HW, col = 2 * (p % HW);
      const bool ok = loader && p < NPF;
      fo[e] = ok ? c * LDW + col : -1;
      pev[e] = (ok && col == NX) ? (unsigned)c * 8u : kBufOob;
      ux[e] = 0; uy[e] = 0;
#pragma unroll
      for (int m = 1; m <= 3; ++m) {               // modes with stance components: LF (0..5), RF (6..11), STANCE (all)
        const int c0s = m == 2 ? 6 : 0, nsf = m == 3 ? 12 : 6;
        const int sc = c - c0s, ucol = (ok && sc >= 0 && sc < nsf) ? BC + sc : -1;
        ux[e] |= (col == ucol ? 1 : 0) << m;
        uy[e] |= (col + 1 == ucol ? 1 : 0) << m;
      }
    }
  }
  // requests of stage k (nt reduced inputs, contact mode code `mode`).  The stage offsets are scalar: k is uniform, but it is used under a
  // wave-level role predicate, where the compiler treats everything as divergent and wraps every buffer load in a waterfall loop otherwise
  __device__ __forceinline__ void prefetch(int k, int nt, int mode) {
    const int cend = 16 * ((BC + nt + 15) >> 4);       // the projection kernel wrote the columns below this one
    const unsigned ku = (unsigned)__builtin_amdgcn_readfirstlane(k);
    const unsigned sW = ku * (unsigned)(PL::W_SIZE * 8), sV = ku * (unsigned)(NJ * WP * 8),
                   sPe = ku * (unsigned)(NU * 8), sB = ku * (unsigned)(NX * 8);
    const __amdgpu_buffer_rsrc_t rW = bp_rsrc(gW), rV = bp_rsrc(gV), rPe = bp_rsrc(gPe), rB = bp_rsrc(gB);
#pragma unroll
    for (int e = 0; e < SW; ++e) w4[e] = __builtin_amdgcn_raw_buffer_load_b128(rW, wc[e] < cend ? wv[e] : kBufOob, sW, 0);
#pragma unroll
    for (int e = 0; e < SV; ++e) v4[e] = __builtin_amdgcn_raw_buffer_load_b128(rV, vc[e] < cend ? vv[e] : kBufOob, sV, 0);
#pragma unroll
    for (int e = 0; e < SF; ++e) pe2[e] = __builtin_amdgcn_raw_buffer_load_b64(rPe, pev[e], sPe, 0);
    fmode = mode;
    if constexpr (JR) {
      dtk = gDt[ku];
#pragma unroll
      for (int e = 0; e < SV; ++e) jb2[e] = __builtin_amdgcn_raw_buffer_load_b64(rB, jbv[e], sB, 0);
    }
  }
  // registers -> LDS: W = [A~ | b~ | B~], PW = [Px | Pe | Pu].  ([Q~ | q~] and [P~ | r~ | R~] have one reader each: they go straight from HBM
  // into the accumulator registers of the waves that start from them - a 16-byte store of a loader wave took ~150 cycles beside the LDS traffic
  // of the other waves, and the staging of everything was the longest job of its phase wherever it was put)
  __device__ __forceinline__ void stage(double (*W)[LDW], double (*PW)[LDW]) const {
    double* Wf = &W[0][0]; double* PWf = &PW[0][0];
#pragma unroll
    for (int e = 0; e < SW; ++e)
      if ((e + 1) * NLD <= NPW || wo[e] >= 0) *reinterpret_cast<bp_u32x4*>(Wf + wo[e]) = w4[e];
#pragma unroll
    for (int e = 0; e < SV; ++e)
      if ((e + 1) * NLD <= NPV || vo[e] >= 0) {
        *reinterpret_cast<bp_u32x4*>(PWf + vo[e]) = v4[e];
        if constexpr (JR) {
          const double x = __hiloint2double((int)v4[e].y, (int)v4[e].x), y = __hiloint2double((int)v4[e].w, (int)v4[e].z);
          const double jb = __hiloint2double((int)jb2[e].y, (int)jb2[e].x);
          double2 t; t.x = __builtin_fma(dtk, x, ((jid[e] & 1) ? 1.0 : 0.0) + jb); t.y = __builtin_fma(dtk, y, (jid[e] & 2) ? 1.0 : 0.0);
          *reinterpret_cast<double2*>(Wf + vo[e]) = t;
        }
      }
#pragma unroll
    for (int e = 0; e < SF; ++e)
      if ((e + 1) * NLD <= NPF || fo[e] >= 0) {
        double2 v;
        v.x = pev[e] != kBufOob ? __hiloint2double((int)pe2[e].y, (int)pe2[e].x) : (((ux[e] >> fmode) & 1) ? 1.0 : 0.0);
        v.y = ((uy[e] >> fmode) & 1) ? 1.0 : 0.0;
        *reinterpret_cast<double2*>(PWf + fo[e]) = v;
      }
  }
};

template <int NJ, bool JW = true>
__device__ __forceinline__ void riccati_mfma8s(RiccatiMfma8sWorkspace<NJ>& ws, const RiccatiFastIO& io) {
  using WS = RiccatiMfma8sWorkspace<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, NT = kRiccati8Threads, LDN = WS::LDN, LDW = WS::LDW, RE = WS::RE, RW = WS::RW, ZR = WS::ZR;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  constexpr int KS = (NX + 3) / 4;          // k-steps over the state dimension
  constexpr int BC = NX + 1;                // first column of B~ / Pu / R~ in the packed layouts
  static_assert(NX == NU, "packed layouts assume nx == nu");
  static_assert(NX + 1 + RE <= kWave, "one lane per column of [H | G g]");
  static_assert(NX > 16 && NX < 32 && KS > 4, "column nx of S sits in the second block column");
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int li = l & 15, lk = l >> 4;
  const int N = io.base.N;
  const bool role_c = w < 4, role_l = w >= 4 && w < 7, role_f = w == 6, role_e = w == 7;

  const int k_top = (io.k_hi < N ? io.k_hi : N) - 1;
  const bool resumed = io.k_hi < N;
  {
    double* z = &ws.Sn[0][0];
    constexpr int total = (int)(offsetof(WS, status) / sizeof(double));
    for (int idx = tid; idx < total; idx += NT) z[idx] = 0.0;     // every matrix and its padding
  }
  __syncthreads();
  if (tid == 0) ws.status = resumed ? (int)io.carry[NXX + NX] : 0;
  if (!resumed && io.reg != 0.0 && tid < NX) ws.Sn[tid][tid] = io.reg;
  if (resumed) {
    for (int idx = tid; idx < NXX; idx += NT) ws.Sn[idx / NX][idx % NX] = io.carry[idx];
    if (tid < NX) ws.Sn[tid][NX] = io.carry[NXX + tid];
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
    if (io.k_lo == 0 && io.with_ls && tid < kWave) linesearch_begin_wave<NJ>(&ws.Sn[0][0], io.ls, tid);
    return;
  }

  constexpr int NLD = 3 * kWave;
  BufStageLoader<NJ, NLD, LDW, !JW> ld;
  ld.init(io, (w - 4) * kWave + l, role_l);
  if (role_l && k_top >= io.k_lo) {     // the first stage: requested and staged into set 0 (nut / mode of ws are visible: barrier above)
    ld.prefetch(k_top, ws.nut[k_top], ws.mode[k_top]);
    ld.stage(ws.W[0], ws.PW[0]);
  }
  __syncthreads();
#ifdef BPMPC_RICCATI_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define RS8PROF(slot) do { const long long tn_ = clock64(); tacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define RS8PROF(slot) ((void)0)
#endif

  // [Acl | bcl] = [A | b] - B Y, [K | kff] = [Px | Pe] - Pu Y of a finished stage (block bw of each), from the set it was staged into.
  // The results leave through buffer stores: where an element of the block goes (Acl and K share the offset; the column nx goes to bcl / kff)
  // is a byte offset per register that is fixed for the whole sweep, an element that goes nowhere has an offset beyond the resource and the
  // stage is the scalar offset of the instruction - one instruction per store.  (As predicated plain stores with 64-bit addresses the
  // eight stores of a block were most of the 2.1 k cycles a block took beside the elimination.)
  const __amdgpu_buffer_rsrc_t rAcl = bp_rsrc(io.Acl), rKf = bp_rsrc(io.Kfull), rbcl = bp_rsrc(io.bcl), rkff = bp_rsrc(io.kff);
  auto out_offsets = [&](int bw, unsigned* om, unsigned* ov) {
    const int r0 = 16 * (bw >> 1), col = 16 * (bw & 1) + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = r0 + lk + 4 * r;
      om[r] = (rr < NX && col < NX) ? (unsigned)(rr * NX + col) * 8u : kBufOob;
      ov[r] = (rr < NX && col == NX) ? (unsigned)rr * 8u : kBufOob;
    }
  };
  auto finish_outputs = [&](int k, int set, int nt, int bw, const unsigned* om, const unsigned* ov) {
    double (*const W)[LDW] = ws.W[set];
    double (*const PW)[LDW] = ws.PW[set];
    double (*const M)[LDW] = ws.M[set];
    const int ksn = (nt + 3) >> 2;
    const int r0 = 16 * (bw >> 1), c0 = 16 * (bw & 1);
    const int row = r0 + li, rowc = row < RW ? row : ZR;
    v4d acl = blk_load<LDW, RW, ZR>(&W[0][0], r0, c0, l);
    v4d kf = blk_load<LDW, RW, ZR>(&PW[0][0], r0, c0, l);
    double ab[4], ap[4], yb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = 4 * ks + lk;
      yb[ks] = M[kk][c0 + li];                                   // -Y (E stores the gain negated); rows >= nt are zero
      ab[ks] = W[rowc][BC + kk];                                 // B(i, kk)
      ap[ks] = PW[rowc][BC + kk];                                // Pu(i, kk)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < ksn) {                                            // wave-uniform
        acl = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[ks], yb[ks], acl, 0, 0, 0);
        kf = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[ks], yb[ks], kf, 0, 0, 0);
      }
    }
    const unsigned ku = (unsigned)__builtin_amdgcn_readfirstlane(k);
    const unsigned sm = ku * (unsigned)(NXX * 8), sv = ku * (unsigned)(NX * 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bp_u32x2 va, vk;
      va.x = (unsigned)__double2loint(acl[r]); va.y = (unsigned)__double2hiint(acl[r]);
      vk.x = (unsigned)__double2loint(kf[r]); vk.y = (unsigned)__double2hiint(kf[r]);
      __builtin_amdgcn_raw_buffer_store_b64(va, rAcl, om[r], sm, 0);
      __builtin_amdgcn_raw_buffer_store_b64(vk, rKf, om[r], sm, 0);
      if (c0 != 0) {                                             // wave-uniform: the block column that holds column nx
        __builtin_amdgcn_raw_buffer_store_b64(va, rbcl, ov[r], sv, 0);
        __builtin_amdgcn_raw_buffer_store_b64(vk, rkff, ov[r], sv, 0);
      }
    }
  };
  // m = q~ - Y' r~, m0 = -r~' H^-1 g of a finished stage (one wave).  q~ and r~ come straight from HBM (m_request, a phase ahead): lane l holds q~[l]
  // and, l < 16, r~[l] (zero beyond the reduced inputs: masked by the offset of the request), which every lane needs - through a 128-byte LDS slot.
  const __amdgpu_buffer_rsrc_t rQp = bp_rsrc(io.Qp), rMt = bp_rsrc(io.Mt);
  auto m_request = [&](int k, int nt, v4d& qr /* [0]: q~, [1]: r~ */) {
    const unsigned ku = (unsigned)__builtin_amdgcn_readfirstlane(k);
    const bp_u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rQp, l < NX ? (unsigned)(l * PackedLq<NJ>::QP + NX) * 8u : kBufOob, ku * (unsigned)(PackedLq<NJ>::Q_SIZE * 8), 0);
    const bp_u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rMt, l < nt ? (unsigned)(l * PackedLq<NJ>::WP + NX) * 8u : kBufOob, ku * (unsigned)(PackedLq<NJ>::M_SIZE * 8), 0);
    qr[0] = __hiloint2double((int)a.y, (int)a.x); qr[1] = __hiloint2double((int)b.y, (int)b.x);
  };
  auto finish_m = [&](int k, int set, double qm, double rr) {
    if (l < RE) ws.rm[l] = rr;
    lds_wave_sync();
    if (l <= NX) {
      double yv[RE], rv[RE];
#pragma unroll
      for (int i = 0; i < RE; ++i) { yv[i] = ws.M[set][i][l]; rv[i] = ws.rm[i]; }
      double m0 = qm, m1 = 0.0;                                  // (lane nx: no q~, its request returned zero)
#pragma unroll
      for (int i = 0; i < RE; i += 2) { m0 += yv[i] * rv[i]; m1 += yv[i + 1] * rv[i + 1]; }     // yv: -Y
      if (l < NX) io.mvec[(size_t)k * NX + l] = m0 + m1; else io.mscal[k] = m0 + m1;
    }
    lds_wave_sync();
  };
  // [Q~ | q~] block of a wave that forms a block of Sn, [P~ | r~ | R~] block column of a chain wave of block row 0: the accumulators these
  // waves start from, requested straight from HBM in the accumulator layout (lane (li, lk), register r <-> row r0 + lk + 4 r, column c0 + li)
  auto acc_request = [&](__amdgpu_buffer_rsrc_t rs, unsigned soff, const unsigned* off, v4d& acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bp_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off[r], soff, 0);
      acc[r] = __hiloint2double((int)v.y, (int)v.x);
    }
  };
  // Registers that live across the stage loop are kept ONCE for all roles (a wave has one role, but the register allocator does not know: a
  // register set per role went into scratch memory, and a scratch reload waits for every request in flight):
  //   ro[0..3]   chain waves: where the rows of the wave's output block go in Acl / K;   loaders: where the rows of the wave's block of [Q~ | q~] come from
  //   ro[4..7]   chain waves: the same for column nx of the block (bcl / kff)
  //   accn       C0, C1: [P~ | r~ | R~] block column of the next stage;   loaders: [Q~ | q~] block of this stage
  //   acc2       C0: third block column of [P~ | r~ | R~];   L5: q~, r~ of the stage whose m it forms next
  unsigned ro[8];
  v4d accn = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
  const int sn_sid = w == 4 ? 0 : (w == 5 ? 1 : 3);             // block of Sn of a loader wave
  if (role_l) {
    const int r0 = 16 * (sn_sid >> 1), c0 = 16 * (sn_sid & 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + lk + 4 * r, col = c0 + li;
      ro[r] = (row < NX && col < 2 * ((NX + 2) / 2)) ? (unsigned)(row * PackedLq<NJ>::QP + col) * 8u : kBufOob;     // (the pairs of [Q~ | q~] that carry anything)
      ro[4 + r] = kBufOob;
    }
  }
  auto g_request = [&](int k, int nt, int c0, v4d& g) {          // rows >= nt and block columns that the projection kernel does not write: zeros
    const int cend = 16 * ((BC + nt + 15) >> 4);
    unsigned off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int row = lk + 4 * r, col = c0 + li; off[r] = (row < nt && col < cend) ? (unsigned)(row * PackedLq<NJ>::WP + col) * 8u : kBufOob; }
    acc_request(rMt, (unsigned)__builtin_amdgcn_readfirstlane(k) * (unsigned)(PackedLq<NJ>::M_SIZE * 8), off, g);
  };

  // The blocks of S = Sn - Z' Yn a chain wave of block row bi needs, as A operands of sym(S) W (see the header); ksp: k-steps over the
  // reduced inputs of the stage that produced Z, Yn (0: S = Sn).  sv: s in the lanes of column nx of the second block column (masked by the caller).
  auto s_operands = [&](int bi, int ksp, bool want_s, double (&a)[KS], double (&sv)[4], v4d* x_keep /* bi == 0: S(0,0); bi == 1: {S(0,1), S(1,1)}; may be null */) {
    const double* Snf = &ws.Sn[0][0];
    double z0[4], y0[4], z1[4], y1[4];
    if (bi == 0) {
      v4d x1 = blk_load<LDN, 32, 0>(Snf, 0, 0, l), x2 = blk_load_t<LDN>(Snf, 0, 0, l), xt = blk_load_t<LDN>(Snf, 0, 16, l);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) { const int p = 4 * ps + lk; z0[ps] = -ws.Zt[p][li]; y0[ps] = ws.Yn[p][li]; y1[ps] = ws.Yn[p][16 + li]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
        if (ps < ksp) xt = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[ps], z0[ps], xt, 0, 0, 0);      // (k, m) <- S[m][16 + k]
      if (want_s) {      // s[m] = S[m][nx] = xt(k = nx - 16, m): lanes lk == (nx - 16) % 4, register (nx - 16) / 4  ->  lanes (nx - 16, lk'), register r: s[lk' + 4 r]
        double* sx = ws.sx[w & 3];
        sx[lk == (NX - 16) % 4 ? li : 16 + li] = xt[(NX - 16) / 4];
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
        if (ps < ksp) {
          x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0[ps], y0[ps], x1, 0, 0, 0);
          x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[ps], z0[ps], x2, 0, 0, 0);
        }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[ks] = 0.5 * (x1[ks] + x2[ks]);
#pragma unroll
      for (int ks = 4; ks < KS; ++ks) a[ks] = xt[ks - 4];
      if (want_s) {
        lds_wave_sync();
        const double* sx = ws.sx[w & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = sx[lk + 4 * r];
      }
      if (x_keep) x_keep[0] = x1;
    } else {
      v4d x01 = blk_load<LDN, 32, 0>(Snf, 0, 16, l), x1 = blk_load<LDN, 32, 0>(Snf, 16, 16, l), x2 = blk_load_t<LDN>(Snf, 16, 16, l);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) { const int p = 4 * ps + lk; z0[ps] = -ws.Zt[p][li]; z1[ps] = -ws.Zt[p][16 + li]; y1[ps] = ws.Yn[p][16 + li]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
        if (ps < ksp) x01 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0[ps], y1[ps], x01, 0, 0, 0);     // (k, m) <- S[k][16 + m]
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
        if (ps < ksp) {
          x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1[ps], y1[ps], x1, 0, 0, 0);
          x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[ps], z1[ps], x2, 0, 0, 0);
        }
      const double one = 16 + li < NX ? 1.0 : 0.0, half = 0.5 * one;            // rows >= nx of sym(S) are zero
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[ks] = one * x01[ks];
#pragma unroll
      for (int ks = 4; ks < KS; ++ks) a[ks] = half * (x1[ks - 4] + x2[ks - 4]);
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[r] = x1[r];
      if (x_keep) { x_keep[0] = x01; x_keep[1] = x1; }
    }
  };

  if (!role_l) out_offsets(w < 3 ? w : 3, ro, ro + 4);     // block w; C0 also forms block 3 (its offsets are made when it gets there)
  if (role_c && (w >> 1) == 0 && k_top >= io.k_lo) { g_request(k_top, ws.nut[k_top], 16 * (w & 1), accn); if ((w & 1) == 0) g_request(k_top, ws.nut[k_top], 32, acc2); }
  int pend_k = -1, pend_nt = 0;          // stage whose outputs are still to be finished (uniform)
  int cur = 0;                           // set of stage k; stage k - 1: (cur + 1) % 3, stage k + 1: (cur + 2) % 3
  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = ws.nut[k];
    const int nxt = cur == 2 ? 0 : cur + 1, prv = cur == 0 ? 2 : cur - 1;
    double (*const W)[LDW] = ws.W[cur];
    double (*const M)[LDW] = ws.M[cur];
    const int ksn = (nt + 3) >> 2;
    const int ksp = (pend_nt + 3) >> 2;
    const int nbc = (BC + nt + 15) >> 4;
    auto sn_block = [&](int sid, v4d acc) {     // [Sn | sn] = [Q | q] + A' SW(:, 0..nx), block sid of four; acc: its block of [Q~ | q~]
      const int r0 = 16 * (sid >> 1), c0 = 16 * (sid & 1);
      const int acol = r0 + li < NX ? r0 + li : LDW - 1;             // the last padding column of W is always zero
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
    // ---- C phase
    if (role_c) {
      // block (bi, bj0) of S W and its part of G in ONE straight line per (k-steps of the stage before, block row): every LDS operand of the
      // phase is requested first - a wave issues in order, and requests that follow the first matrix instructions wait behind them -, then
      // the nineteen matrix instructions with the arithmetic of the symmetrisation in their shadow.  (As a function of ksp with branches
      // around every matrix instruction and the operands of S W requested behind the S blocks the phase took 3.1 k cycles for 1.2 k of matrix time.)
      const int bi = w >> 1, bj0 = w & 1;
      const int c0 = 16 * bj0;
      const double smask = (c0 + li == NX) ? 1.0 : 0.0;
      constexpr int KG1 = KS - 4;
      double a[KS];
      auto chain = [&](auto kspc, auto bic) {
        constexpr int KSP = decltype(kspc)::value, BI = decltype(bic)::value;
        constexpr int KSA = KSP > 0 ? KSP : 1;
        const double* Snf = &ws.Sn[0][0];
        double za[KSA], zb[KSA], ya[KSA], yb[KSA], b[KS], ga[4];
        v4d xa, x1, x2, g;
        if constexpr (BI == 0) { x1 = blk_load<LDN, 32, 0>(Snf, 0, 0, l); x2 = blk_load_t<LDN>(Snf, 0, 0, l); xa = blk_load_t<LDN>(Snf, 0, 16, l); }
        else { xa = blk_load<LDN, 32, 0>(Snf, 0, 16, l); x1 = blk_load<LDN, 32, 0>(Snf, 16, 16, l); x2 = blk_load_t<LDN>(Snf, 16, 16, l); }
#pragma unroll
        for (int ps = 0; ps < KSP; ++ps) {
          const int p = 4 * ps + lk;
          za[ps] = ws.Zt[p][li]; yb[ps] = ws.Yn[p][16 + li];
          if constexpr (BI == 0) ya[ps] = ws.Yn[p][li]; else zb[ps] = ws.Zt[p][16 + li];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = W[4 * ks + lk][c0 + li];
#pragma unroll
        for (int ks = 0; ks < (BI == 0 ? 4 : KG1); ++ks) ga[ks] = W[16 * BI + 4 * ks + lk][BC + li];
        if constexpr (BI == 0) g = accn; else g = v4d{0.0, 0.0, 0.0, 0.0};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ps = 0; ps < KSP; ++ps) { za[ps] = -za[ps]; if constexpr (BI != 0) zb[ps] = -zb[ps]; }
        double sv[4];
        if constexpr (BI == 0) {
          // XT(k, m) = S[m][16 + k] first: its row nx - 16 is s, on its way to the lanes of column nx through the wave's own LDS slot
#pragma unroll
          for (int ps = 0; ps < KSP; ++ps) xa = __builtin_amdgcn_mfma_f64_16x16x4f64(yb[ps], za[ps], xa, 0, 0, 0);
          double* sx = ws.sx[w & 3];
          sx[lk == (NX - 16) % 4 ? li : 16 + li] = xa[(NX - 16) / 4];
#pragma unroll
          for (int ps = 0; ps < KSP; ++ps) {
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(za[ps], ya[ps], x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ps], za[ps], x2, 0, 0, 0);
          }
          lds_wave_sync();
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[r] = sx[lk + 4 * r];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) a[ks] = 0.5 * (x1[ks] + x2[ks]);
#pragma unroll
          for (int ks = 4; ks < KS; ++ks) a[ks] = xa[ks - 4];
        } else {
          // X01(k, m) = S[k][16 + m], then the diagonal block (1,1) and its transpose
#pragma unroll
          for (int ps = 0; ps < KSP; ++ps) xa = __builtin_amdgcn_mfma_f64_16x16x4f64(za[ps], yb[ps], xa, 0, 0, 0);
#pragma unroll
          for (int ps = 0; ps < KSP; ++ps) {
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(zb[ps], yb[ps], x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(yb[ps], zb[ps], x2, 0, 0, 0);
          }
          const double one = 16 + li < NX ? 1.0 : 0.0, half = 0.5 * one;            // rows >= nx of sym(S) are zero
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) a[ks] = one * xa[ks];
#pragma unroll
          for (int ks = 4; ks < KS; ++ks) a[ks] = half * (x1[ks - 4] + x2[ks - 4]);
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[r] = x1[r];
        }
        v4d acc = {smask * sv[0], smask * sv[1], smask * sv[2], smask * sv[3]};
#pragma unroll
        for (int ks = 0; ks < ((BPMPC_RS8_ABLATE & 8) ? 1 : KS); ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
        blk_store<LDW, 32>(&ws.SW[0][0], 16 * BI, c0, l, acc);
#pragma unroll
        for (int ks = 0; ks < ((BPMPC_RS8_ABLATE & 4) ? 0 : (BI == 0 ? 4 : KG1)); ++ks) g = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[ks], acc[ks], g, 0, 0, 0);
        blk_store<LDW, 32>(BI == 0 ? &M[0][0] : &ws.Mb[0][0], 0, c0, l, g);
      };
#define BP_CHAIN_CASE(K) case K: if (bi == 0) chain(std::integral_constant<int, K>{}, std::integral_constant<int, 0>{}); else chain(std::integral_constant<int, K>{}, std::integral_constant<int, 1>{}); break;
      switch ((BPMPC_RS8_ABLATE & 2) ? 0 : ksp) { BP_CHAIN_CASE(0) BP_CHAIN_CASE(1) BP_CHAIN_CASE(2) BP_CHAIN_CASE(3) default: BP_CHAIN_CASE(4) }
#undef BP_CHAIN_CASE
      if (nbc > 2 && bj0 == 0) {       // a third block column (more than 32 - nx - 1 reduced inputs): the waves of block column 0 take it too, their S operands are formed
        const int r0 = 16 * bi, c2 = 32;
        double b[KS], ga[4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = W[4 * ks + lk][c2 + li];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ga[ks] = (bi == 0 || ks < KG1) ? W[r0 + 4 * ks + lk][BC + li] : 0.0;
        v4d g = acc2;
        if (bi != 0) g = v4d{0.0, 0.0, 0.0, 0.0};
        __builtin_amdgcn_sched_barrier(0);
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
        blk_store<LDW, 32>(&ws.SW[0][0], r0, c2, l, acc);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (bi == 0 || ks < KG1) g = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[ks], acc[ks], g, 0, 0, 0);
        blk_store<LDW, 32>(bi == 0 ? &M[0][0] : &ws.Mb[0][0], 0, c2, l, g);
      }
    } else if (role_l) {
      // the requests of stage k - 1: its registers were staged in the E phase of stage k + 1 (set of stage k: before the loop)
      acc_request(rQp, (unsigned)__builtin_amdgcn_readfirstlane(k) * (unsigned)(PackedLq<NJ>::Q_SIZE * 8), ro, accn);       // [Q~ | q~] of this stage: used right behind the barrier
      if (w == 5 && pend_k >= 0) m_request(pend_k, pend_nt, acc2);
      if (k > io.k_lo) ld.prefetch(k - 1, ws.nut[k - 1], ws.mode[k - 1]);     // never beyond the chunk: earlier stages may not be projected yet
    }
    RS8PROF(0);
    lds_barrier();                     // Bb
    RS8PROF(1);
    // ---- E phase
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
      if (rhs) {
        for (int i = nt; i < 4 * ksn; ++i) { ws.Zt[i][col] = 0.0; ws.Yn[i][col] = 0.0; }
      }
      static_assert(NX + 2 + 3 < LDN - 1 && 4 * KS <= NX + 2, "spare columns of Z / Yn");
      const int ecol = rhs ? col : NX + 2 + (l & 3);
      auto emit = [&](int p, double z, double y) { ws.Zt[p][ecol] = z; ws.Yn[p][ecol] = y; };
#define BP_GJS_CASE(ROWS, FWD, BWD)                                                           \
      {                                                                                       \
        double v[ROWS];                                                                       \
        {                                                                                     \
          double ta[ROWS], tb[ROWS];                                                          \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i) ta[i] = M[i][col];                 \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i) tb[i] = ws.Mb[i][col];             \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i) asm volatile("" : "+v"(ta[i]));    \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i) asm volatile("" : "+v"(tb[i]));    \
          /* (no masks: rows >= nt of M and Mb are zero - B~ has no columns there -, and a lane without a column eliminates column 0 into a spare column) */ \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i) v[i] = ta[i] + tb[i];             \
        }                                                                                     \
        ok = FWD(v, nt, emit);                                                          \
        if (l == 0 && !ok) ws.status = 1;                                                     \
        RS8PROF(2);                                                                           \
        lds_barrier();                 /* Ba */                                               \
        RS8PROF(3);                                                                           \
        BWD<ROWS>(v, nt);              /* beside the C phase of the next stage */             \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) if (rhs && i < nt) M[i][col] = -v[i];    /* -Y */ \
      }
      if (rows_layout) {
        if (nt <= 8) BP_GJS_CASE(8, forward_eliminate_rows<8>, back_substitute_rows)
        else if (nt == 9) BP_GJS_CASE(9, (forward_eliminate_rows<9, true>), back_substitute_rows)
        else BP_GJS_CASE(10, (forward_eliminate_rows<10, true>), back_substitute_rows)
      } else {
        if (nt <= 12) BP_GJS_CASE(12, forward_eliminate_wave<12>, back_substitute_wave)
        else BP_GJS_CASE(RE, forward_eliminate_wave<RE>, back_substitute_wave)
      }
#undef BP_GJS_CASE
    } else {
      // L4, L5, F: a block of Sn each, then the registers of stage k - 1 -> LDS (third set; the requests went out in the C phase);
      // C0: output blocks 0 and 3, C1, C2: 1, 2; L5: m.  C3 idles beside E.
      // C0, C1: [P~ | r~ | R~] of stage k - 1 for the wave's block column (C0: and the third one), first thing in the phase: used behind S W of the next
      // C phase.  (Requested at the start of the C phase into a second register set, the wait for the first set - vmcnt retires in order and the
      // compiler cannot count across the loop edge - became a wait for the requests just issued: C phase 2.7 k -> 4.4 k cycles.)
      if (w < 2 && k > io.k_lo) { g_request(k - 1, ws.nut[k - 1], 16 * w, accn); if (w == 0) g_request(k - 1, ws.nut[k - 1], 32, acc2); }
      if (role_l) sn_block(sn_sid, accn);
      if (w == 5 && pend_k >= 0) finish_m(pend_k, prv, acc2[0], acc2[1]);
#ifndef BPMPC_RS8_OUT3
#define BPMPC_RS8_OUT3 0      // who forms output block 3 beside the elimination: 0: C0 (behind its block 0), 1: C3 (on the SIMD of the elimination wave)
#endif
      if (w < (BPMPC_RS8_OUT3 == 1 ? 4 : 3) && pend_k >= 0) {
        finish_outputs(pend_k, prv, pend_nt, w, ro, ro + 4);
        if (BPMPC_RS8_OUT3 == 0 && w == 0) { unsigned ob[8]; out_offsets(3, ob, ob + 4); finish_outputs(pend_k, prv, pend_nt, 3, ob, ob + 4); }
      }
#ifdef BPMPC_RICCATI_PROFILE
      if (BPMPC_RICCATI_PROFILE == 3 && role_l) { RS8PROF(4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RS8PROF(5); }
#endif
      if (role_l && k > io.k_lo) ld.stage(ws.W[nxt], ws.PW[nxt]);
#ifdef BPMPC_RICCATI_PROFILE
      if (BPMPC_RICCATI_PROFILE == 3 && role_l) { RS8PROF(6); }
#endif
      RS8PROF(2);
      lds_barrier();                   // Ba
      RS8PROF(3);
    }
    pend_k = k; pend_nt = nt;
    cur = nxt;
  }
  const int last = cur == 0 ? 2 : cur - 1;      // set of the last stage
#ifdef BPMPC_RICCATI_PROFILE
  if (io.prof && tid == 0) for (int i = 0; i < 4; ++i) io.prof[i] = (double)tacc[i];
  if (io.prof && tid == 7 * kWave) io.prof[7] = (double)tacc[2];
  if (io.prof && tid == 4 * kWave) { io.prof[4] = (double)tacc[4]; io.prof[5] = (double)tacc[5]; io.prof[6] = (double)tacc[6]; }
#endif
  __syncthreads();                              // -Y of the last stage
  if (role_f) out_offsets(3, ro, ro + 4);
  if ((w < 3 || role_f) && pend_k >= 0) finish_outputs(pend_k, last, pend_nt, w < 3 ? w : 3, ro, ro + 4);
  if (w == 5 && pend_k >= 0) { m_request(pend_k, pend_nt, acc2); finish_m(pend_k, last, acc2[0], acc2[1]); }
  if (io.k_lo > 0) {                                   // hand over to the launch that sweeps the earlier stages: [S | s] of stage k_lo
    v4d xk[2];
    double a[KS], sv[4] = {0.0, 0.0, 0.0, 0.0};
    if (w == 0 || w == 2) s_operands(w >> 1, (pend_nt + 3) >> 2, false, a, sv, xk);
    __syncthreads();                                   // every read of Sn, Z, Yn is done
    if (w == 0) blk_store<LDN, 32>(&ws.Sn[0][0], 0, 0, l, xk[0]);
    if (w == 2) { blk_store<LDN, 32>(&ws.Sn[0][0], 0, 16, l, xk[0]); blk_store<LDN, 32>(&ws.Sn[0][0], 16, 16, l, xk[1]); }
    __syncthreads();
    for (int idx = tid; idx < NXX; idx += NT) { const int r = idx / NX, c = idx % NX; io.carry[idx] = (r >= 16 && c < 16) ? ws.Sn[c][r] : ws.Sn[r][c]; }
    if (tid < NX) io.carry[NXX + tid] = ws.Sn[tid][NX];
    if (tid == 0) io.carry[NXX + NX] = (double)ws.status;
    return;
  }
  __syncthreads();
  {
    const int st = ws.status;
    __syncthreads();                                   // the workspace is dead from here on: it holds the state history
    constexpr int kHistCap = ((int)(offsetof(WS, status) / sizeof(double)) - kStepNormsScratch * NT / kWave) / NX - 8;
    static_assert(kHistCap >= 64, "roll-out history");
#ifdef BPMPC_RICCATI_PROFILE
    const long long tr0 = clock64();
#endif
    riccati_rollout_deep<NJ, NT>(reinterpret_cast<double*>(&ws), kHistCap, st, io);
#ifdef BPMPC_RICCATI_PROFILE
    if (BPMPC_RICCATI_PROFILE == 1 && io.prof && tid == 0) { const long long te = clock64(); io.prof[5] = (double)(te - tr0); io.prof[6] = (double)te - io.prof[6]; }
#endif
  }
}

}  // namespace bpmpc
