// Riccati sweep with TWO wavefronts per problem, each owning block columns of the packed width (HIP only; same mathematics and layouts as
// riccati_wave.h / riccati_wave2.h, which give a problem one wavefront).
//
// Between two and eight problems per CU neither of those fills the chip: one wave per problem leaves SIMDs idle (two problems per CU) or
// works them at the issue rate of a lone wave (four), and two waves per SIMD need eight problems per CU.  The products of a stage split by
// BLOCK COLUMN without any exchange - SW(:, bj) = S W(:, bj), M(:, bj) += B' SW(:, bj), Sn(:, bj) += A' SW(:, bj) touch nothing of another
// column - and so do the updates ([S | s], [Acl | bcl], [K | kff] block column by block column).  So:
//     wave 0: columns 0 (and 2, when the stage has more than 9 reduced inputs): S W, G, Sn(0,0);  the elimination;  S(0,0), Acl(:,0), K(:,0)
//     wave 1: column 1: S W (+ s), [g | H head], Sn(0,1), Sn(1,1), r~ / q~;  while wave 0 eliminates: the output stores of the stage above and
//             the loads it has registers for;  S(0,1), S(1,1), Acl(:,1), K(:,1) and the vector columns bcl, kff
// Three LDS-only barriers per stage (after the tile of [G g H] is written; after the elimination; after the updated S and the outputs are in
// LDS), against five of riccati_mfma.h; both waves read the whole symmetrised S back (the A-operand of S W needs all of it).
// Everything else - buffer loads masked by their offsets, registers as operands, forward elimination + back substitution, symmetric update
// S = Sn - Z' Yn, output tiles in the HBM layout, lane indices opaque per stage - is riccati_wave2.h.
#pragma once
#include <hip/hip_runtime.h>

#include "riccati_wave2.h"

namespace bpmpc {

constexpr int kRiccatiPairThreads = 2 * kWave;

template <int NJ>
struct RiccatiPairWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int LDM = PackedLq<NJ>::WP + 2;
  alignas(16) double Mx[16][LDM];          // [G | g | H] -> Y
  alignas(16) double Zt[16][34];
  alignas(16) double Yn[16][34];
  alignas(16) double St[32][34];           // the updated [S | s] on its way to both waves
  alignas(16) double oA[NX * NX];          // outputs of a stage in their HBM layout (stored by wave 1 during the next stage's elimination)
  alignas(16) double oK[NU * NX];
  double ob[NX], ok[NU], om[2][NX + 2];
  double rv[16], qv[32];
  unsigned char nut[kMaxRiccatiStages], mode[kMaxRiccatiStages];
};

template <int NJ, int W>
__device__ __forceinline__ void riccati_pair_wave(RiccatiPairWorkspace<NJ>& ws, const RiccatiFastIO& io) {
  using WS = RiccatiPairWorkspace<NJ>;
  using PL = PackedLq<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, WP = PL::WP, QP = PL::QP, BC = NX + 1, NXX = NX * NX, NXU = NX * NU, LDM = WS::LDM;
  constexpr int KS = (NX + 3) / 4, XR = NX - 16;
  static_assert(KS == 6 && NX + 1 <= 32 && NX + 1 + 16 <= 48 && WP >= 48 && NX == NU, "two block rows, three block columns");
  int l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const int N = io.base.N;
  const int k_top = (io.k_hi < N ? io.k_hi : N) - 1;
  const bool resumed = io.k_hi < N;
  const double* const zero = io.zero_one + 2;

  {
    int tw = 0;
    for (int idx = l; idx < N && idx < kMaxRiccatiStages; idx += kWave) {          // both waves: the same values
      const int n = io.base.nut[idx];
      tw |= n > 16 ? 1 : 0;
      ws.nut[idx] = (unsigned char)n;
      ws.mode[idx] = (unsigned char)(n > 0 ? (io.mode[idx] & 3) : kModeEvent);
    }
    lds_barrier();
    if (__any(tw)) {
      if (W == 0 && l == 0) io.carry[NXX + NX] = 1.0;
      return;
    }
  }
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
    if (W == 0 && l == 0) io.carry[NXX + NX] = (double)status;
    return;
  }

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
    oBt[bi] = row < NX ? 8u * (unsigned)(row * WP + BC + lk) : kOut;
    oPu[bi] = (j >= 0 && row < NX) ? 8u * (unsigned)(j * WP + BC + lk) : kOut;
  }
  const unsigned gQ = 8u * (unsigned)(lk * QP + li);
  constexpr int RQL = (NX - 16) / 4;
  const unsigned gQl = (16 + 4 * RQL + lk < NX) ? gQ + 1024u * (4 + RQL) : kOut;
  const unsigned gM = 8u * (unsigned)(lk * WP + li);
  const unsigned gV = 8u * (unsigned)(lk * WP + li);
  const unsigned gVl = (8 + lk < NJ) ? gV + 2 * RS : kOut;
  const unsigned gPe = li == XR ? 8u * (unsigned)lk : kOut;

  // ---- operand registers of this wave
  double cWa[KS];                             // own column of W (0 / 1): B-operand of S W, initial value of [Acl | bcl](:, own)
  double cWb[KS];                             // wave 0: column 2 (stages with three block columns); wave 1: column 0 (A-operand of Sn(0, 1))
  double cWT[KS];                             // wave 1: column 1 restricted to the state columns (A-operand of Sn(1, 1))
  double cB[KS];                              // W[4 ks + lk][BC + li], li < nt
  v4d cMa, cMb;                               // Mt blocks of the own column (wave 0: also column 2)
  v4d cQa, cQb;                               // wave 0: Qp(0,0);  wave 1: Qp(0,1), Qp(1,1)
  double cBt[2][4], cPu[2][4];
  v4d cPI[2];                                 // [Px | Pe](:, own column)

  auto load_products = [&](int k, int nt) {
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
    const __amdgpu_buffer_rsrc_t rm = rsrc(io.Mt + (size_t)k * PL::M_SIZE, PL::M_SIZE);
    const __amdgpu_buffer_rsrc_t rq = rsrc(io.Qp + (size_t)k * PL::Q_SIZE, PL::Q_SIZE);
    const int nbc = (BC + nt + 15) >> 4;
    const bool in = li < nt;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      cB[ks] = bload(rw, in ? offW(ks) + 8u * BC : kOut);
      if (W == 0) {
        cWa[ks] = bload(rw, offW(ks));
        if (nbc > 2) cWb[ks] = bload(rw, offW(ks) + 256u);
      } else {
        cWa[ks] = bload(rw, offW(ks) + 128u);
        cWb[ks] = bload(rw, offW(ks));
        cWT[ks] = bload(rw, lx ? offW(ks) + 128u : kOut);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned o = (lk + 4 * r < nt) ? gM + RS * (unsigned)r : kOut;
      if (W == 0) {
        cMa[r] = bload(rm, o);
        if (nbc > 2) cMb[r] = bload(rm, o + 256u);
        cQa[r] = bload(rq, gQ + 1024u * r);
      } else {
        cMa[r] = bload(rm, o + 128u);
        cQa[r] = bload(rq, lxe ? gQ + 1024u * r + 128u : kOut);
        if (r > RQL || (r == RQL && 16 + 4 * RQL >= NX)) cQb[r] = 0.0;
        else if (r == RQL) cQb[r] = bload(rq, lxe ? gQl + 128u : kOut);
        else cQb[r] = bload(rq, lxe ? gQ + 1024u * (4 + r) + 128u : kOut);
      }
    }
  };
  auto load_late = [&](int k, int nt) {       // B~ and Pu row-major in the A-operand, [Px | Pe] of the own column
    const __amdgpu_buffer_rsrc_t rw = rsrc(io.Wt + (size_t)k * PL::W_SIZE, PL::W_SIZE);
    const __amdgpu_buffer_rsrc_t rv = rsrc(io.Vt + (size_t)k * (NJ * WP), NJ * WP);
    const __amdgpu_buffer_rsrc_t rp = rsrc(io.base.Pe + (size_t)k * NU, NU);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bool in = 4 * ks + lk < nt;
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        cBt[bi][ks] = bload(rw, in ? oBt[bi] + 32u * ks : kOut);
        cPu[bi][ks] = bload(rv, in ? oPu[bi] + 32u * ks : kOut);
      }
    }
    if (W == 0) {
      cPI[0] = v4d{0.0, 0.0, 0.0, 0.0}; cPI[1] = v4d{0.0, 0.0, 0.0, 0.0};
      cPI[0][3] = bload(rv, gV); cPI[1][0] = bload(rv, gV + RS); cPI[1][1] = bload(rv, gVl);
    } else {
      cPI[1] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 3; ++r) cPI[0][r] = bload(rp, gPe + 32u * r);
      cPI[0][3] = bload(rv, lxe ? gV + 128u : kOut); cPI[1][0] = bload(rv, lxe ? gV + RS + 128u : kOut); cPI[1][1] = bload(rv, lxe ? gVl + 128u : kOut);
    }
  };
  auto flush = [&](int hk) {                  // wave 1: outputs of stage hk, assembled in LDS by both waves, to HBM in 16-byte chunks
    double2* A2 = reinterpret_cast<double2*>(io.Acl + (size_t)hk * NXX);
    double2* K2 = reinterpret_cast<double2*>(io.Kfull + (size_t)hk * NXU);
    const double2* a2 = reinterpret_cast<const double2*>(ws.oA);
    const double2* k2 = reinterpret_cast<const double2*>(ws.oK);
    constexpr int NIT = (NXX / 2 + kWave - 1) / kWave;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = l + it * kWave;
      if (it + 1 < NIT || idx < NXX / 2) { A2[idx] = a2[idx]; K2[idx] = k2[idx]; }
    }
    if (l < NX) {
      io.bcl[(size_t)hk * NX + l] = ws.ob[l];
      io.kff[(size_t)hk * NU + l] = ws.ok[l];
      io.mvec[(size_t)hk * NX + l] = ws.om[hk & 1][l];
    }
    if (l == NX) io.mscal[hk] = ws.om[hk & 1][NX];
  };

  auto stage_nt = [&](int k) { return __builtin_amdgcn_readfirstlane((int)ws.nut[k >= io.k_lo ? k : io.k_lo]); };
  auto stage_mode = [&](int k) { return __builtin_amdgcn_readfirstlane((int)ws.mode[k >= io.k_lo ? k : io.k_lo]); };
  int nt_c = stage_nt(k_top), mode_c = stage_mode(k_top);
  int nt_n = stage_nt(k_top - 1), mode_n = stage_mode(k_top - 1);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { cWb[ks] = 0.0; cWT[ks] = 0.0; }
  cMb = v4d{0.0, 0.0, 0.0, 0.0}; cQb = v4d{0.0, 0.0, 0.0, 0.0};
  load_products(k_top, nt_c);

#ifdef BPMPC_RICCATI_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define RPPROF(slot) do { const long long tn_ = clock64(); tacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define RPPROF(slot) ((void)0)
#endif
  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = nt_c;
    const int nbc = (BC + nt + 15) >> 4;
    const int ksn = (nt + 3) >> 2;
    const bool more = k > io.k_lo;
    const int nt_nn = stage_nt(k - 2), mode_nn = stage_mode(k - 2);
    asm volatile("" : "+v"(l));               // lane indices opaque per stage (riccati_wave2.h)
    li = l & 15; lk = l >> 4; lx = li < XR; lxe = li <= XR;
    // ---- products of the own block column(s)
    double Sa1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Sa1[ks] = lx ? S[ks >> 2][1][ks & 3] : 0.0;
    v4d ma, mb = cMb, sna, snb = cQb;          // own M block (wave 0: + column 2), own Sn blocks (wave 0: (0,0); wave 1: (0,1), (1,1))
    {
      v4d sw0 = {0.0, 0.0, 0.0, 0.0}, sw1 = {0.0, 0.0, 0.0, 0.0};
      if (W == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { sw0[r] = li == XR ? S[0][1][r] : 0.0; sw1[r] = li == XR ? S[1][1][r] : 0.0; }
        if (li == XR) {                                               // r~ and q~ as loaded (m = q~ - Y' r~ in the elimination)
#pragma unroll
          for (int r = 0; r < 4; ++r) { ws.rv[lk + 4 * r] = cMa[r]; ws.qv[lk + 4 * r] = cQa[r]; ws.qv[16 + lk + 4 * r] = cQb[r]; }
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sw0 = __builtin_amdgcn_mfma_f64_16x16x4f64(S[ks >> 2][0][ks & 3], cWa[ks], sw0, 0, 0, 0);
        sw1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa1[ks], cWa[ks], sw1, 0, 0, 0);
      }
      ma = cMa;
      if (nt > 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ma = __builtin_amdgcn_mfma_f64_16x16x4f64(cB[ks], ks < 4 ? sw0[ks & 3] : sw1[ks & 3], ma, 0, 0, 0);
      }
      sna = cQa;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sna = __builtin_amdgcn_mfma_f64_16x16x4f64(W == 0 ? cWa[ks] : cWb[ks], ks < 4 ? sw0[ks & 3] : sw1[ks & 3], sna, 0, 0, 0);
        if (W == 1) snb = __builtin_amdgcn_mfma_f64_16x16x4f64(cWT[ks], ks < 4 ? sw0[ks & 3] : sw1[ks & 3], snb, 0, 0, 0);
      }
    }
    if (W == 0 && nbc > 2) {                   // third block column: S W and G only
      v4d sw0 = {0.0, 0.0, 0.0, 0.0}, sw1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sw0 = __builtin_amdgcn_mfma_f64_16x16x4f64(S[ks >> 2][0][ks & 3], cWb[ks], sw0, 0, 0, 0);
        sw1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa1[ks], cWb[ks], sw1, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) mb = __builtin_amdgcn_mfma_f64_16x16x4f64(cB[ks], ks < 4 ? sw0[ks & 3] : sw1[ks & 3], mb, 0, 0, 0);
    }
    // the registers of W are the initial values of [Acl | bcl](:, own column)
    v4d acl[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) { acl[0][r] = cWa[r]; acl[1][r] = (4 + r < KS) ? cWa[(4 + r < KS) ? 4 + r : 0] : 0.0; }
    if (nt > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ws.Mx[lk + 4 * r][16 * W + li] = ma[r];
        if (W == 0 && nbc > 2) ws.Mx[lk + 4 * r][32 + li] = mb[r];
      }
    }
    load_late(k, nt);
    if (more) load_products(k - 1, nt_n);       // their registers are dead from here on: a lead of the elimination and the updates
    RPPROF(0);
    lds_barrier();                              // B1: [G | g | H], r~, q~ are in LDS; the output tiles of the stage above are complete
    RPPROF(1);
    if (W == 0) {
      // ---- elimination (riccati_wave.h)
      bool ok = true;
      double* const om = ws.om[k & 1];
      if (nt > 0) {
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
        const int ecol = rhs ? col : NX + 2 + (l & 3);
        auto emit = [&](int p, double z, double y) { ws.Zt[p][ecol] = z; ws.Yn[p][ecol] = y; };
#define BP_GJ_CASE(ROWS, FWD, BWD)                                                                             \
        {                                                                                                      \
          double v[ROWS];                                                                                      \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i) { const double t = ws.Mx[i][col]; v[i] = (used && i < nt) ? t : 0.0; } \
          lds_wave_sync();                                                                                     \
          ok = FWD;                                                                                            \
          BWD<ROWS>(v, nt);                                                                                    \
          double mt = (rhs && rid < NX) ? ws.qv[rid] : 0.0;                                                    \
          _Pragma("unroll") for (int i = 0; i < ROWS; ++i)                                                     \
            if (rhs && i < nt) { ws.Mx[i][col] = v[i]; mt -= v[i] * ws.rv[i]; }                                \
          if (rhs) om[rid] = mt;                                                                               \
        }
        if (rows_layout) {
          if (nt <= 8) BP_GJ_CASE(8, forward_eliminate_rows<8>(v, nt, emit), back_substitute_rows)
          else if (nt == 9) BP_GJ_CASE(9, forward_eliminate_rows<9>(v, nt, emit), back_substitute_rows)
          else BP_GJ_CASE(10, forward_eliminate_rows<10>(v, nt, emit), back_substitute_rows)
        } else {
          if (nt <= 12) BP_GJ_CASE(12, forward_eliminate_wave<12>(v, nt, emit), back_substitute_wave)
          else BP_GJ_CASE(16, forward_eliminate_wave<16>(v, nt, emit), back_substitute_wave)
        }
#undef BP_GJ_CASE
      } else {
        if (l <= NX) om[l] = l < NX ? ws.qv[l] : 0.0;
      }
      if (!__builtin_amdgcn_readfirstlane((int)ok)) status = 1;
    } else {
      if (k < k_top) flush(k + 1);              // beside the elimination: the stores of the stage above
    }
    RPPROF(2);
    lds_barrier();                              // B2: Y, Z, Yn
    RPPROF(3);
    // ---- updates of the own block column
    double yb[4], zA0[4], zA1[4], yn[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      yb[ks] = -ws.Mx[4 * ks + lk][16 * W + li];
      zA0[ks] = -ws.Zt[4 * ks + lk][li];
      if (W == 1) { const double z = -ws.Zt[4 * ks + lk][16 + li]; zA1[ks] = lx ? z : 0.0; }
      yn[ks] = ws.Yn[4 * ks + lk][16 * W + li];
    }
    {
      const int c0s = wave_stance_first(mode_c), nsf = wave_stance_count(mode_c);
      const int s = li - c0s;
      const bool stance = s >= 0 && s < nsf;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) cPu[0][ks] = li < 12 ? ((stance && s == 4 * ks + lk) ? 1.0 : 0.0) : cPu[0][ks];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      if (ks < ksn) {
        sna = __builtin_amdgcn_mfma_f64_16x16x4f64(zA0[ks], yn[ks], sna, 0, 0, 0);
        if (W == 1) snb = __builtin_amdgcn_mfma_f64_16x16x4f64(zA1[ks], yn[ks], snb, 0, 0, 0);
        acl[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cBt[0][ks], yb[ks], acl[0], 0, 0, 0);
        acl[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cBt[1][ks], yb[ks], acl[1], 0, 0, 0);
        cPI[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cPu[0][ks], yb[ks], cPI[0], 0, 0, 0);
        cPI[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cPu[1][ks], yb[ks], cPI[1], 0, 0, 0);
      }
    // the updated S and the outputs into LDS
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ws.St[lk + 4 * r][16 * W + li] = sna[r];
      if (W == 1) ws.St[16 + lk + 4 * r][16 + li] = snb[r];
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * bi + 4 * r >= NX) continue;
        const int row = 16 * bi + lk + 4 * r;
        const bool rin = 16 * bi + 4 * r + 3 < NX || row < NX;
        if (W == 0) {
          if (rin) { ws.oA[row * NX + li] = acl[bi][r]; ws.oK[row * NX + li] = cPI[bi][r]; }
        } else {
          if (rin && lx) { ws.oA[row * NX + 16 + li] = acl[bi][r]; ws.oK[row * NX + 16 + li] = cPI[bi][r]; }
          if (rin && li == XR) { ws.ob[row] = acl[bi][r]; ws.ok[row] = cPI[bi][r]; }
        }
      }
    RPPROF(4);
    lds_barrier();                              // B3: S and the outputs are in LDS
    RPPROF(5);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = lk + 4 * r, row1 = 16 + row, col1 = 16 + li;
      const double s00 = ws.St[row][li], s00t = ws.St[li][row];
      const double s01 = ws.St[row][col1], s10 = ws.St[li][row1];       // S(16 + row, li) = S(li, 16 + row)
      const double s11 = ws.St[row1][col1], s11t = ws.St[col1][row1];
      S[0][0][r] = 0.5 * (s00 + s00t);
      S[0][1][r] = s01;
      S[1][0][r] = (16 + 4 * r < NX && row1 < NX) ? s10 : 0.0;
      S[1][1][r] = (row1 < NX && col1 < NX) ? 0.5 * (s11 + s11t) : s11;
    }
    nt_c = nt_n; mode_c = mode_n; nt_n = nt_nn; mode_n = mode_nn;
    RPPROF(6);
  }
#ifdef BPMPC_RICCATI_PROFILE
  if (io.prof && l == 0 && W == 0)
    for (int i = 0; i < 8; ++i) io.prof[i] = (double)tacc[i];
#endif
  lds_barrier();                                // (the last read of St precedes nothing else; the outputs of stage k_lo are complete since B3)
  if (W == 1) flush(io.k_lo);
  if (W == 0) {
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
}

template <int NJ>
__device__ __forceinline__ void riccati_pair(RiccatiPairWorkspace<NJ>& ws, const RiccatiFastIO& io, int role) {
  if (role == 0) riccati_pair_wave<NJ, 0>(ws, io);
  else riccati_pair_wave<NJ, 1>(ws, io);
}

}  // namespace bpmpc
