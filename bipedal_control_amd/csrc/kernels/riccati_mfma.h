// Riccati sweep on the FP64 matrix cores (HIP only; reference with the same mathematics: riccati.h).
//
// One 256-thread workgroup per problem; Gauss-Jordan, barriers and roll-out come from riccati_fast.h.  The
// products of a stage are 16x16 output blocks accumulated with v_mfma_f64_16x16x4_f64:
//   * one wavefront per output block; an MFMA reads one A and one B element per lane (a 16x4 and a 4x16 operand), so a
//     22x22x22 product costs 2 x 64-lane LDS reads per 1024 multiply-adds instead of two 16-byte reads per four - the
//     2x2-register-tile version was bound by LDS bandwidth and by the dependent-FMA latency of a single wave per SIMD;
//     a chain of dependent MFMAs issues back to back (measured 64 cycles each, tools/probes/mfma_f64_probe.hip);
//   * the stage data is staged in packed, zero-padded layouts so that the vectors ride along as extra columns:
//         W  = [A~ | b~ | B~]         PW = [Px | Pe | Pu]        Qq = [Q~ | q~]        M = [P~ | r~ | R~]
//         SW = sym(S) W  (+ s in the b column)                    -> [S A | S b + s | S B]
//         M += B' SW                                              -> [G | g | H]
//         Sn = Qq + A' SW(:, 0..nx)                               -> [Q + A'SA | q + A'(S b + s)]
//         Y  = H^-1 [G g]                 (wave 0, registers; waves 1-3 compute Sn meanwhile)
//         [S | s]     = Sn - G' Y         [Acl | bcl] = [A | b] - B Y         [K | kff] = [Px | Pe] - Pu Y
//     so g, s, bcl, kff cost no extra instructions;
//   * S is stored as computed and symmetrised when it is loaded as an operand (0.5 (S_ik + S_ki)), which saves the
//     separate symmetrisation pass and its barrier.
// Operand rows / columns beyond nx are masked to zero on load, so padding never carries stale values into a product.
#pragma once
#include <hip/hip_runtime.h>

#include "riccati_fast.h"


namespace bpmpc {

typedef double v4d __attribute__((ext_vector_type(4)));

// Where element (row, col) of an LDS matrix lives.  LdsRows<LD>: row major with leading dimension LD.
// LdsSwz<NBC> (round 6, the eight-wave sweep): NBC block columns of 16; rows 2j and 2j + 1 share a PAIR ROW of 32 NBC + 2 doubles in which the
// 16-column chunks of the two rows alternate, the order inside a chunk pair exchanged for odd block columns:
//     ix = (row / 2) PS + 32 (col / 16) + 16 ((row ^ (col / 16)) & 1) + col % 16,   PS = 32 NBC + 2.
// ds_read_b64 serves a wave in two groups of 32 lanes over 64 banks of 4 B, i.e. the 32 doubles of a group must differ mod 32.  The operand loads of
// v_mfma_f64_16x16x4_f64 come in two lane patterns (li = lane % 16, lk = lane / 16; a group = lk in {0, 1} or {2, 3}):
//     (1) rows r0 + li, columns 4 ks + lk   (A operand of a row-stored matrix)          (2) rows 4 ks + lk, columns c0 + li   (B operand; accumulator layout)
// Row major serves (1) with LD = 2 mod 32 and (2) with LD = 16 mod 32, never both: at LD = 34 / 50 / 66 every load of pattern (2) - 250 of the
// sweep's 300 operand loads per stage - was a two-way conflict (SQ_LDS_BANK_CONFLICT 0.24 of SQ_LDS_IDX_ACTIVE, profiles/r05_sq_counters.csv).
// Swizzled: (1) 2 (li / 2) + 16 (li & 1) + lk + const, (2) 16 (lk ^ chunk) + (c0 + li) % 16 + const (also across a chunk boundary) - both hit 32
// different doubles mod 32; a row read by one column per lane (the elimination) separates columns c and c + 16.  Pairs (col even) stay 16-byte aligned.
template <int LD>
struct LdsRows {
  static constexpr int kCols = LD;
  __host__ __device__ static constexpr int ix(int row, int col) { return row * LD + col; }
  static constexpr int size(int rows) { return rows * LD; }
};
template <int NBC>
struct LdsSwz {
  static constexpr int kCols = 16 * NBC, PS = 32 * NBC + 2;
  __host__ __device__ static constexpr int ix(int row, int col) { return (row >> 1) * PS + ((col >> 4) << 5) + (((row ^ (col >> 4)) & 1) << 4) + (col & 15); }
  static constexpr int size(int rows) { return ((rows + 1) / 2) * PS; }
};

// DB: the staged operands are double buffered (stage k-1 is staged while the updates of stage k still read theirs, which
// saves a barrier per stage).  At nx = 22 that costs 98 KB of LDS (one workgroup per CU) against 66 KB single buffered (two
// workgroups per CU): the solver double buffers when the batch does not exceed the number of CUs.
template <int NJ, bool DB>
struct RiccatiMfmaWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int ZR = NX;                                  // a row that is always zero: block rows >= nx are read there
  // rows: two full 16-row blocks when double buffered (plain block loads / stores), otherwise nx + the zero row(s)
  // (block rows >= nx are then read from the zero row and not stored)
  static constexpr int RB = DB ? 32 : ((NX + 2 + 1) / 2) * 2;
  static constexpr int RCL = DB ? 32 : NX;                       // block rows below this are read as they are
  static constexpr int LDN = 34;                                 // leading dimension of the [.. | vector] blocks (nx + 1 <= 32 columns)
  static constexpr int WC = NX + 1 + NU;                         // packed width [A | b | B]
  static constexpr int LDW = ((WC + 15) / 16) * 16 + 2;
  static_assert(NX + 1 <= 32 && NU <= 32 && ((NX + 3) / 4) * 4 <= RB, "two block rows / columns");
  static constexpr int NBUF = DB ? 2 : 1;
  alignas(16) double S[RB][LDN];        // [S | s], not symmetrised
  alignas(16) double Qq[NBUF][RB][LDN]; // [Q~ | q~]
  alignas(16) double Sn[RB][LDN];       // [Sn | sn]
  alignas(16) double G0[RB][LDN];       // [G | g] before the elimination
  alignas(16) double W[NBUF][RB][LDW];  // [A~ | b~ | B~]
  alignas(16) double PW[NBUF][RB][LDW]; // [Px | Pe | Pu]
  alignas(16) double SW[RB][LDW];       // sym(S) W
  // Rows of M = reduced inputs.  The single-buffered nx = 24 variant keeps 16 of them (rows 16.. of [P~ | r~ | R~] are zero whenever a
  // stage has at most 16 reduced inputs - every mode of this problem family leaves at most nu - 10 = 14) so that two workgroups share
  // a CU (79 KB each); a stage with more makes the sweep report a numerical failure instead of computing nonsense.
  static constexpr int RBM = (!DB && NX > 22) ? 16 : RB;
  alignas(16) double M[NBUF][RBM][LDW]; // [P~ | r~ | R~] -> [G | g | H] -> Y in the first nx + 1 columns
  double r[NBUF][NU];
  alignas(16) double dx[2][NX];
  int status;
  unsigned char nut[kMaxRiccatiStages]; // reduced input dimensions of all stages (a global load per stage would sit on the critical path)
  unsigned char mode[kMaxRiccatiStages]; // contact mode of the stages (the force rows of [Px | Pe | Pu] are generated from it)
};

// D-layout of v_mfma_f64_16x16x4_f64: lane l, register r  <->  row (l / 16) + 4 r, column l % 16 of the 16x16 block.
// The LDS matrices hold RB < 32 rows: block rows >= NXR (all zero by construction) are read from the zero row ZR and not stored.
template <int LD, int NXR, int ZR>
__device__ __forceinline__ v4d blk_load(const double* Mx, int r0, int c0, int l) {
  v4d c;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + (l >> 4) + 4 * r;
    if constexpr (NXR >= 32) c[r] = lds1(Mx[row * LD + c0 + (l & 15)]);
    else c[r] = lds1(Mx[(row < NXR ? row : ZR) * LD + c0 + (l & 15)]);
  }
  return c;
}
template <int LD, int RBR>
__device__ __forceinline__ void blk_store(double* Mx, int r0, int c0, int l, v4d c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + (l >> 4) + 4 * r;
    if constexpr (RBR >= 32) Mx[row * LD + c0 + (l & 15)] = c[r];
    else if (row < RBR) Mx[row * LD + c0 + (l & 15)] = c[r];
  }
}

constexpr int kModeEvent = 4;                 // contact mode code of an event node (no inputs: no ones in the force rows)

// [Px | Pe | Pu] of a stage for the sweeps.  Its joint rows arrive packed from the elimination kernel (Vt: row stride WP, the column
// layout of the LDS operand, a 16-byte pair of HBM is a 16-byte pair of LDS, complete rows: zeros beyond the reduced inputs); its
// FORCE rows are not stored anywhere: row c < 12 is zero except Pe_c in column nx and, for a component of a stance contact, a single 1
// in its own reduced-input column nx + 1 + (c - first stance component) (project_lu_s.h) - generated from the contact mode of the stage.
// Loader thread t owns the pairs t, t + NLD, .. of both parts and rewrites all of them at every stage (the buffers are reused across
// modes).  The staging (registers -> LDS) sits on the sweep's critical path - a loader wave issues an instruction every ~8 cycles and
// the staging barrier waits for it -, so everything that can be decided a stage ahead is decided in prefetch(): the first value of a
// force-row pair is LOADED (Pe_c for the pair that opens at column nx, otherwise 0.0 or 1.0 from a two-entry table), the second is a
// register set there; stage() is four ds_write_b128 with fixed addresses.  (First version: masks, mode look-ups and selects in stage(),
// +60 instructions = +0.5 k cycles per stage, 0.334 -> 0.354 ms per sweep.)
template <int NJ, int NLD, int LDW, class LW = LdsRows<LDW>>
struct PwVtLoader {
  using PL = PackedLq<NJ>;
  static constexpr int NX = PL::NX, NU = PL::NU, WP = PL::WP, BC = NX + 1, HW = WP / 2;
  static constexpr int NPV = NJ * HW, NPF = 12 * HW;
  static constexpr int SV = (NPV + NLD - 1) / NLD, SF = (NPF + NLD - 1) / NLD;
  static_assert(NX % 2 == 0 && LDW % 2 == 0 && WP <= LW::kCols, "column nx opens a pair; pairs stay aligned and inside the rows");
  double vx[SV], vy[SV], fx[SF], fy[SF];
  const double2* gV;
  const double* gPe;
  const double* zero_one;                       // {0.0, 1.0, 0.0, 0.0} in global memory
  int vo[SV], fo[SF];                           // LDS element offset of the pair (-1: none)
  // force-row pairs, decided once for the whole sweep: pe_off >= 0: the pair opens at column nx (its first value is Pe of that component);
  // ux / uy: bit m set = under contact mode code m the first / second value of the pair is the 1 of a stance component
  int pe_off[SF], ux[SF], uy[SF];
  int tl;
  __device__ __forceinline__ int vc(int e) const { return 2 * ((tl + e * NLD) % HW); }      // first column of the joint-row pair of slot e
  __device__ __forceinline__ void init(const RiccatiFastIO& io, int tl_, bool loader, size_t k) {
    tl = tl_;
    const int tp = loader ? tl : 0;
    gV = reinterpret_cast<const double2*>(io.Vt + k * (NJ * WP)) + tp;
    gPe = io.base.Pe + k * NU;
    zero_one = io.zero_one;
#pragma unroll
    for (int e = 0; e < SV; ++e) { const int p = tp + e * NLD; vo[e] = p < NPV ? LW::ix(12 + p / HW, 2 * (p % HW)) : -1; }
#pragma unroll
    for (int e = 0; e < SF; ++e) {
      const int p = tp + e * NLD, c = p < NPF ? p / HW : 0, col = 2 * (p % HW);
      fo[e] = p < NPF ? LW::ix(p / HW, 2 * (p % HW)) : -1;
      pe_off[e] = col == NX ? c : -1;
      ux[e] = 0; uy[e] = 0;
#pragma unroll
      for (int m = 1; m <= 3; ++m) {               // modes with stance components: LF (0..5), RF (6..11), STANCE (all)
        const int c0s = m == 2 ? 6 : 0, nsf = m == 3 ? 12 : 6;
        const int s = c - c0s, ucol = (s >= 0 && s < nsf) ? BC + s : -1;
        ux[e] |= (col == ucol ? 1 : 0) << m;
        uy[e] |= (col + 1 == ucol ? 1 : 0) << m;
      }
    }
  }
  // the stage the pointers stand on (nt reduced inputs; contact mode code `mode`: 0..3, kModeEvent for an event node), then one stage down
  __device__ __forceinline__ void prefetch(int nt, int mode) {
    const int cend = 16 * ((BC + nt + 15) >> 4);       // the elimination kernel wrote the columns below this one
    const double2* zero2 = reinterpret_cast<const double2*>(zero_one + 2);
#pragma unroll
    for (int e = 0; e < SV; ++e) {                     // a pair beyond the written columns (or a lane without a pair) loads zeros: masked by the ADDRESS, nothing to do when it is staged
      const double2 v = *((vo[e] >= 0 && vc(e) < cend) ? gV + e * NLD : zero2);
      vx[e] = v.x; vy[e] = v.y;
    }
#pragma unroll
    for (int e = 0; e < SF; ++e) {
      fx[e] = *(pe_off[e] >= 0 ? gPe + pe_off[e] : zero_one + ((ux[e] >> mode) & 1));
      fy[e] = ((uy[e] >> mode) & 1) ? 1.0 : 0.0;
    }
    gV -= (NJ * WP) / 2; gPe -= NU;
  }
  __device__ __forceinline__ void stage(double* PWf) const {
#pragma unroll
    for (int e = 0; e < SV; ++e)
      if ((e + 1) * NLD <= NPV || vo[e] >= 0) { double2 v; v.x = vx[e]; v.y = vy[e]; *reinterpret_cast<double2*>(PWf + vo[e]) = v; }
#pragma unroll
    for (int e = 0; e < SF; ++e)
      if ((e + 1) * NLD <= NPF || fo[e] >= 0) { double2 v; v.x = fx[e]; v.y = fy[e]; *reinterpret_cast<double2*>(PWf + fo[e]) = v; }
  }
};

// Prefetch registers and staging of the loader threads of the sweeps (riccati_mfma.h, riccati_mfma8.h).  The projected model
// arrives in the packed layout of project_node.h (PackedLq: Wt = [At | bt | Bt], Qp = [Qt | qt], Mt = [Pt | rt | Rt], row strides
// of whole block columns), which is the column layout of the LDS operands: a 16-byte pair of HBM is a 16-byte pair of LDS, one
// ds_write_b128 each.  Loader thread t owns the pairs t, t + NLD, .. of every stream.  The writer leaves the block columns
// >= nbc and the rows >= nut of Mt untouched; they are neither loaded nor trusted here, zeros are staged instead.
//   NLD: loader threads; MR: rows of Mt that are kept (reduced inputs the sweep can hold); LDW / LDN: LDS leading dimensions.
//   AM: pairs that are not written by the projection kernel are masked by the ADDRESS of their load (a pair of zeros) instead of by selects at
//   staging time - the staging sits between the S update and S W of the sweep, the requests do not (riccati_mfma8.h, round 4).
//   JR: Wt holds no joint rows (k_project_fast<.., WJ = false>): they are [I | b | 0] + dt x (joint rows of [Px | Pe | Pu]) and are completed at
//   staging time from the pairs of Vt this thread holds anyway - the same pair index, the same LDS offset, in W instead of PW (round 4).
template <int NJ, int NLD, int MR, int LDW, int LDN, bool AM = false, bool JR = false, class LW = LdsRows<LDW>, class LN = LdsRows<LDN>>
struct PackedStageLoader {
  using PL = PackedLq<NJ>;
  static constexpr int NX = PL::NX, NU = PL::NU, WP = PL::WP, QP = PL::QP, BC = NX + 1, NXX = NX * NX;
  static constexpr int HW = WP / 2, HQ = QP / 2;                    // pairs per row in HBM
  static constexpr int HQU = (NX + 2) / 2;                          // pairs per row of Qp that carry anything ([Q~ | q~]: nx + 1 columns)
  static constexpr int NPW = (JR ? 12 : NX) * HW, NPQ = NX * HQU, NPM = MR * HW;
  static constexpr int SW = (NPW + NLD - 1) / NLD, SQ = (NPQ + NLD - 1) / NLD, SM = (NPM + NLD - 1) / NLD;
  static_assert(LDW % 2 == 0 && LDN % 2 == 0 && NX % 2 == 0 && WP <= LW::kCols && QP <= LN::kCols && MR <= NU, "pairs stay aligned and inside the rows");
  // (x / y halves in separate arrays of doubles: arrays of double2 that live across the stage loop end up in scratch memory)
  double wx[SW], wy[SW], qx[SQ], qy[SQ], mx[SM], my[SM];
  const double2 *gW, *gQ, *gM;
  PwVtLoader<NJ, NLD, LDW, LW> pw;                                      // [Px | Pe | Pu]
  // per slot, fixed for the whole sweep: LDS element offset of the pair, its column (and row, for Mt) for the masks; -1: no pair
  int wo[SW], wc[SW], qo[SQ], mo[SM], mc[SM], mr[SM];
  int tl;
  // JR: per pair of Vt: the identity entries of the joint row inside the pair, the row whose b the pair carries (the pair that opens at column nx; -1: none)
  static constexpr int SVJ = PwVtLoader<NJ, NLD, LDW, LW>::SV;
  double jix[SVJ], jiy[SVJ], jb[SVJ], dtk;
  int jbo[SVJ];
  const double *gB, *gDt;

  __device__ __forceinline__ void init(const RiccatiFastIO& io, int tl_, bool loader, size_t k) {
    tl = tl_;
    const int tp = loader ? tl : 0;
    pw.init(io, tl_, loader, k);
    gW = reinterpret_cast<const double2*>(io.Wt + k * PL::W_SIZE) + tp;
    gM = reinterpret_cast<const double2*>(io.Mt + k * PL::M_SIZE) + tp;
    // Qp: the thread's pairs are not t + e NLD of the HBM rows (the last pairs of a row carry nothing), so the pointer stands on
    // the node and the pair offset is part of the slot
    gQ = reinterpret_cast<const double2*>(io.Qp + k * PL::Q_SIZE);
    if constexpr (JR) {
      gB = io.lqb + k * NX; gDt = io.gdt + k; dtk = 0.0;
#pragma unroll
      for (int e = 0; e < SVJ; ++e) {
        const int p = tp + e * NLD, row = 12 + p / HW, col = 2 * (p % HW);
        const bool ok = p < NJ * HW;
        jix[e] = (ok && col == row) ? 1.0 : 0.0; jiy[e] = (ok && col + 1 == row) ? 1.0 : 0.0;
        jbo[e] = (ok && col == NX) ? row : -1; jb[e] = 0.0;
      }
    }
#pragma unroll
    for (int e = 0; e < SW; ++e) { const int p = tp + e * NLD; const bool ok = p < NPW; wo[e] = ok ? LW::ix(p / HW, 2 * (p % HW)) : -1; wc[e] = 2 * (p % HW); }
#pragma unroll
    for (int e = 0; e < SQ; ++e) { const int p = tp + e * NLD; const bool ok = p < NPQ; qo[e] = ok ? LN::ix(p / HQU, 2 * (p % HQU)) : -1; }
#pragma unroll
    for (int e = 0; e < SM; ++e) { const int p = tp + e * NLD; const bool ok = p < NPM; mo[e] = ok ? LW::ix(p / HW, 2 * (p % HW)) : -1; mc[e] = 2 * (p % HW); mr[e] = p / HW; }
  }
  // loads of the stage the pointers stand on (its reduced input dimension: nt), then one stage down
  __device__ __forceinline__ void prefetch(int nt, int mode) {
    const int cend = 16 * ((BC + nt + 15) >> 4);                     // first column that is not written / not needed
#pragma unroll
    for (int e = 0; e < SW; ++e) {
      if constexpr (AM) { if ((e + 1) * NLD <= NPW || wo[e] >= 0) { const double2 v = *(wc[e] < cend ? gW + e * NLD : reinterpret_cast<const double2*>(pw.zero_one + 2)); wx[e] = v.x; wy[e] = v.y; } }
      else if (((e + 1) * NLD <= NPW || wo[e] >= 0) && wc[e] < cend) { const double2 v = gW[e * NLD]; wx[e] = v.x; wy[e] = v.y; }
    }
#pragma unroll
    for (int e = 0; e < SQ; ++e) if ((e + 1) * NLD <= NPQ || qo[e] >= 0) { const int p = tl + e * NLD; const double2 v = gQ[(p / HQU) * HQ + p % HQU]; qx[e] = v.x; qy[e] = v.y; }
#pragma unroll
    for (int e = 0; e < SM; ++e) {
      if constexpr (AM) { if ((e + 1) * NLD <= NPM || mo[e] >= 0) { const double2 v = *((mr[e] < nt && mc[e] < cend) ? gM + e * NLD : reinterpret_cast<const double2*>(pw.zero_one + 2)); mx[e] = v.x; my[e] = v.y; } }
      else if (((e + 1) * NLD <= NPM || mo[e] >= 0) && mr[e] < nt && mc[e] < cend) { const double2 v = gM[e * NLD]; mx[e] = v.x; my[e] = v.y; }
    }
    pw.prefetch(nt, mode);
    if constexpr (JR) {
      dtk = *gDt;
#pragma unroll
      for (int e = 0; e < SVJ; ++e) jb[e] = *(jbo[e] >= 0 ? gB + jbo[e] : pw.zero_one);
      gB -= NX; gDt -= 1;
    }
    gW -= PL::W_SIZE / 2; gQ -= PL::Q_SIZE / 2; gM -= PL::M_SIZE / 2;
  }
  // registers -> LDS: W = [A~ | b~ | B~], Qq = [Q~ | q~], M = [P~ | r~ | R~] (MR rows), PW = [Px | Pe | Pu], r~ also to rvec
  __device__ __forceinline__ void stage(double* W, double* PW, double* Qq, double* M, double* rvec, int nt) const {      // (flat: element (r, c) at LW / LN::ix)
    stage_wm(W, M, rvec, nt);
    stage_pq(PW, Qq);
  }
  // the two halves of stage(): what the first products of a stage read (W, M) and what is read a phase later or by the outputs only (PW, Qq)
  __device__ __forceinline__ void stage_wm(double* Wf, double* Mf, double* rvec, int nt) const {
    const int cend = 16 * ((BC + nt + 15) >> 4);
#pragma unroll
    for (int e = 0; e < SW; ++e)
      if ((e + 1) * NLD <= NPW || wo[e] >= 0) {     // only the last slot of a stream is partial; (a select between two double2 goes through scratch memory: component-wise)
        const bool in = AM || wc[e] < cend; double2 v; v.x = in ? wx[e] : 0.0; v.y = in ? wy[e] : 0.0; *reinterpret_cast<double2*>(Wf + wo[e]) = v;
      }
#pragma unroll
    for (int e = 0; e < SM; ++e)
      if ((e + 1) * NLD <= NPM || mo[e] >= 0) {
        const bool in = AM || (mr[e] < nt && mc[e] < cend);
        double2 v; v.x = in ? mx[e] : 0.0; v.y = in ? my[e] : 0.0;
        *reinterpret_cast<double2*>(Mf + mo[e]) = v;
        if (mc[e] == NX) rvec[mr[e]] = v.x;                          // r~ (nx is even: the first element of its pair)
      }
    if constexpr (JR) {
#pragma unroll
      for (int e = 0; e < SVJ; ++e)
        if ((e + 1) * NLD <= NJ * HW || pw.vo[e] >= 0) {
          double2 v; v.x = __builtin_fma(dtk, pw.vx[e], jix[e] + jb[e]); v.y = __builtin_fma(dtk, pw.vy[e], jiy[e]);
          *reinterpret_cast<double2*>(Wf + pw.vo[e]) = v;
        }
    }
  }
  __device__ __forceinline__ void stage_pq(double* PW, double* Qf) const {
#pragma unroll
    for (int e = 0; e < SQ; ++e)
      if ((e + 1) * NLD <= NPQ || qo[e] >= 0) { double2 v; v.x = qx[e]; v.y = qy[e]; *reinterpret_cast<double2*>(Qf + qo[e]) = v; }
    pw.stage(PW);
  }
};

// Forward roll-out dx_{k+1} = Acl_k dx_k + bcl_k with the rows of three stages in flight (then du, Armijo metric and step
// norms: riccati_step_norms).  The recurrence is one short mat-vec per stage, so whatever global-memory latency sits inside a
// step dominates it:
//   * wave 0 computes, one row of [Acl | bcl] per lane in registers, three register sets loaded three stages ahead; the loop
//     body is straight-line code with unconditional loads, so the compiler can count how many younger loads may stay in flight
//     when a set is consumed;
//   * the state history stays in LDS (the workspace of the backward sweep is dead by now) and goes to HBM afterwards in one
//     coalesced pass: inside the loop wave 0 issues no store, so no load ever waits for a store (vmcnt retires in order).
// (Two lanes per row - half the loads, LDS reads and FMAs per lane, partial sums joined by a DPP quad permutation - was measured:
//  0.3412 -> 0.339 ms at batch 256, 4.48 -> 4.51 ms at batch 4096; the step is bound by the LDS round trip of the state, not by its
//  instruction count.  Not kept.)
// acc += (x of lane P of this lane's 16-lane DPP row) * m, the broadcast as an operand modifier (riccati_rollout_deep)
template <int P, bool NOP>
__device__ __forceinline__ void roll_fma(double& acc, double x, double m) {
  if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(P));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(P));
}
template <int NX, int OFF, int L, int H>      // acc += sum over l = L .. H - 1 of x[lane l] * r[OFF + l]
struct RollDot {
  static __device__ __forceinline__ void run(double& acc, double x, const double (&r)[NX]) { roll_fma<L, false>(acc, x, r[OFF + L]); RollDot<NX, OFF, L + 1, H>::run(acc, x, r); }
};
template <int NX, int OFF, int H>
struct RollDot<NX, OFF, H, H> { static __device__ __forceinline__ void run(double&, double, const double (&)[NX]) {} };

template <int NX, int HH, int L, int E>       // a += sum over l = L .. E - 1 of xlo[lane l] * r[l], c += the same of xhi and r[HH + l]: two independent chains, interleaved
struct RollDot2 {
  static __device__ __forceinline__ void run(double& a, double& c, double xlo, double xhi, const double (&r)[NX]) {
    roll_fma<L, false>(a, xlo, r[L]); roll_fma<L, false>(c, xhi, r[HH + L]); RollDot2<NX, HH, L + 1, E>::run(a, c, xlo, xhi, r);
  }
};
template <int NX, int HH, int E>
struct RollDot2<NX, HH, E, E> { static __device__ __forceinline__ void run(double&, double&, double, double, const double (&)[NX]) {} };

template <int NJ, int NT = kRiccatiThreads>
__device__ __forceinline__ void riccati_rollout_deep(double* hist /*LDS, (cap + 4) * nx doubles*/, int cap, int status, const RiccatiFastIO& io) {
  constexpr int NX = 12 + NJ, NXX = NX * NX;
  const int tid = threadIdx.x;
  const int N = io.base.N;
  if (tid < NX) hist[tid] = io.base.dx0[tid];
  if (tid >= kWave && tid < kWave + NX) io.base.dx[tid - kWave] = io.base.dx0[tid - kWave];
  __syncthreads();
  for (int k0 = 0; k0 < N; k0 += cap) {                 // one pass unless the horizon exceeds the LDS history
    const int nk = N - k0 < cap ? N - k0 : cap;
    if (tid < kWave) {
      // Round 6.  What a step costs is the memory latency of its matrix row divided by the number of stages in flight: with three register sets of
      // 23 doubles per lane a step took ~600 cycles whether the state went through LDS or stayed in registers, whether the requests were 8 or 16 bytes,
      // one cache line or 22 per request (all measured: experiments/LOG.md) - a fourth set took 2 % off the kernel, a fifth does not fit.  So the two
      // HALVES of the wave hold different stages in the same registers - lower half stage 2 m, upper half stage 2 m + 1 of a set - which doubles the stages in
      // flight at the same register count, and the state never leaves the registers:
      //   * every 16-lane DPP row holds the whole state in two registers - xlo: element p in lane position p, xhi: element H + p (H = nx / 2) - and a
      //     lane's dot product takes its operands by `v_fmac_f64_dpp .. row_newbcast` (the broadcast costs no instruction and no memory);
      //   * rows 0 / 2 of the wave compute the low half of dx+, rows 1 / 3 the high half; v_permlane16_swap_b32 (gfx950: swaps the odd rows of its
      //     first operand with the even rows of its second; both operands the new value: tools/probes/permlane_probe.hip) makes the new xlo / xhi of a half;
      //   * v_permlane32_swap_b32 hands the state of the half that owned the stage to the other half, which owns the next one.
      // The history still goes to LDS - for the coalesced output and the step norms behind the loop - but nothing waits for it.
      constexpr int H = NX / 2;
      static_assert(NX % 2 == 0 && H <= 16, "two halves of the state, one per DPP row");
      const int half = tid >> 5, q = (tid >> 4) & 1, p = tid & 15;
      const bool has = p < H;
      const int ri = has ? q * H + p : 0;               // lanes without an element shadow row 0 (keeps the loads unconditional)
      double xlo = hist[has ? p : 0], xhi = hist[has ? H + p : 0];
      double rA[NX], rB[NX], rC[NX], bA, bB, bC;      // three register sets, each holds two stages (one per half of the wave): six in flight (a fourth set: no gain)
      auto load = [&](double (&r)[NX], double& b, int m) {            // pair m of this pass: stages k0 + 2 m (lower half), k0 + 2 m + 1 (upper half)
        const int k = k0 + 2 * m + half;
        const int kc = k < N ? k : N - 1;               // beyond the end: a valid, unused stage
        const double2* p_ = reinterpret_cast<const double2*>(io.Acl + (size_t)kc * NXX + (size_t)ri * NX);      // (rows of nx doubles are 16-byte aligned: nx is even)
#pragma unroll
        for (int l = 0; l < NX / 2; ++l) { const double2 v = p_[l]; r[2 * l] = v.x; r[2 * l + 1] = v.y; }
        b = io.bcl[(size_t)kc * NX + ri];
      };
      auto dot = [&](const double (&r)[NX], double b) {                // b + (row of this lane) . (state); NOP: two wait states between the VALU write of
        double t = b;                                                  // xlo / xhi (the swaps) and their first DPP read, which the compiler cannot see
        roll_fma<0, true>(t, xlo, r[0]);
        RollDot<NX, 0, 1, H>::run(t, xlo, r);
        roll_fma<0, false>(t, xhi, r[H]);
        RollDot<NX, H, 1, H>::run(t, xhi, r);
        return t;
      };
      auto advance = [&](double t, int owner) {                        // the new state, computed by the half `owner`, in every row of the wave
        const auto wl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(t), (unsigned)__double2loint(t), false, false);
        const auto wh = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(t), (unsigned)__double2hiint(t), false, false);
        // [0]: the values of rows 0 / 2 (the low half of dx+) in rows 0, 1 / 2, 3; [1]: those of rows 1 / 3 (the high half)
        const auto ll = __builtin_amdgcn_permlane32_swap(wl[0], wl[0], false, false), lh = __builtin_amdgcn_permlane32_swap(wh[0], wh[0], false, false);
        const auto hl = __builtin_amdgcn_permlane32_swap(wl[1], wl[1], false, false), hh = __builtin_amdgcn_permlane32_swap(wh[1], wh[1], false, false);
        // [owner]: the lower (0) / upper (1) 32 lanes' values in both halves
        xlo = __hiloint2double((int)lh[owner], (int)ll[owner]);
        xhi = __hiloint2double((int)hh[owner], (int)hl[owner]);
      };
      auto step = [&](const double (&r)[NX], double b, int m) {       // pair m: stage j = 2 m on the lower half, j + 1 on the upper half
        const int j = 2 * m;
        const double t0 = dot(r, b);                                   // lower half: dx_(j+1); upper half: its row against a state it does not own yet
        if (has && half == 0 && j < nk) hist[(j + 1) * NX + ri] = t0;  // (stages past the end of the pass run on a valid, unused stage and leave nothing behind)
        advance(t0, 0);
        const double t1 = dot(r, b);                                   // upper half: dx_(j+2)
        if (has && half == 1 && j + 1 < nk) hist[(j + 2) * NX + ri] = t1;
        advance(t1, 1);
      };
      const int npairs = (nk + 1) / 2;
      load(rA, bA, 0); load(rB, bB, 1); load(rC, bC, 2);
      for (int m = 0; m < npairs; m += 3) {
        step(rA, bA, m);     load(rA, bA, m + 3);
        step(rB, bB, m + 1); load(rB, bB, m + 4);
        step(rC, bC, m + 2); load(rC, bC, m + 5);
      }
    } else if (k0 == 0 && io.with_ls && tid < 2 * kWave) {
      // wave 1 has nothing to do here: it prepares the line search of this problem (LDS: the step norms' tile area behind the history)
      linesearch_begin_wave<NJ>(hist + (size_t)(cap + 4) * NX, io.ls, tid - kWave);
    }
    __syncthreads();
    for (int idx = tid; idx < nk * NX; idx += NT) io.base.dx[(size_t)(k0 + 1) * NX + idx] = hist[NX + idx];
    __syncthreads();
    if (k0 + nk < N && tid < NX) hist[tid] = hist[nk * NX + tid];      // input of the next pass (a single pass leaves dx_0 .. dx_N for the step norms)
    __syncthreads();
  }
  // the history stays for the step norms when the horizon fitted one pass; their tiles of K follow it in LDS
  riccati_step_norms<NJ, NT>(status, io, N <= cap ? hist : nullptr, hist + (size_t)(cap + 4) * NX);
}

// Roll-out that reads 5.9 instead of 8.3 KB per stage.  The roll-out of riccati_mfma.h walks dx+ = Acl dx + bcl and computes du = K dx + kff
// for all stages afterwards: Acl AND K, 3.9 KB each.  But 13 of the nx rows of the discretised dynamics are structural - rows 0..2:
// dx+ = dx + b + (dt / m) (sum of the force inputs of that component), joint rows: dx+ = dx + b + dt du - so with du in hand only the rows
// 3..11 of Acl are needed.  Here K is read INSIDE the recurrence (one row per lane, lanes 0..nx-1), the nine dense rows of Acl by lanes
// 32..40, three stages in flight as before; dx and du stay in LDS for the norms and leave in coalesced passes.  Same mathematics
// (dx+ = A dx + B du + b), results equal to the other roll-out to rounding.  At batch 4096 the roll-out is a streaming kernel (5.2 TB/s).
template <int NJ, int NT = kRiccatiThreads>
__device__ __forceinline__ void riccati_rollout_sparse(double* lds /* (cap + 4) nx + cap nu + 256 doubles */, int cap, int status, const RiccatiFastIO& io) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ, NXX = NX * NX, NXU = NX * NU;
  static_assert(NX <= 32 && 9 <= 32, "lanes 0..nx-1: rows of K; lanes 32..40: rows 3..11 of Acl");
  const int tid = threadIdx.x, l = tid & 63;
  const int N = io.base.N;
  double* const hist = lds;                               // dx_0 .. dx_cap (+ slack), row stride nx
  double* const duh = lds + (size_t)(cap + 4) * NX;       // du of the pass, row stride nu
  double* const scratch = duh + (size_t)cap * NU;         // line-search opening (3 * 64 + 5 doubles), then the stance table (<= 512 bytes)
  // which force components are stance components, per stage: a component without contact has K = 0 (and kff = Pe) - its row is not
  // read, the lane loads a row of zeros instead (same instructions, another address: the loads stay unconditional)
  unsigned char* const stance_tab = reinterpret_cast<unsigned char*>(scratch + 200);
  for (int idx = tid; idx < N && idx < kMaxRiccatiStages; idx += NT) {
    const int m = io.base.nut[idx] > 0 ? (io.mode[idx] & 3) : 0;
    stance_tab[idx] = (unsigned char)m;                   // bit 0: components 0..5, bit 1: components 6..11
  }
  const double* const zero_row_ptr = io.zero_one + 4;
  const double dt_over_m_factor = 1.0 / io.model->robot_mass;
  if (tid < NX) hist[tid] = io.base.dx0[tid];
  if (tid >= kWave && tid < kWave + NX) io.base.dx[tid - kWave] = io.base.dx0[tid - kWave];
  __syncthreads();
  double acc_arm = 0.0, acc_x = 0.0, acc_u = 0.0;
  for (int k0 = 0; k0 < N; k0 += cap) {
    const int nk = N - k0 < cap ? N - k0 : cap;
    if (tid < kWave) {
      const bool is_k = l < NX, is_a = l >= 32 && l < 32 + 9;
      const int ri = is_k ? l : (is_a ? 3 + (l - 32) : 0);                     // row of K / of Acl
      double rA[NX], rB[NX], rC[NX], sA, sB, sC, bA, bB, bC, dA, dB, dC;
      int nA, nB, nC;
      auto load = [&](double (&r)[NX], double& sv, double& bv, double& dv, int& nv, int k) {
        const int kc = k < N ? k : N - 1;               // beyond the end: a valid, unused stage
        const int m = stance_tab[kc];
        const bool zero_row = is_k && l < 12 && !((m >> (l >= 6 ? 1 : 0)) & 1);      // force component without contact: K row = 0
        const double* p = zero_row ? zero_row_ptr : (is_a ? io.Acl + (size_t)kc * NXX : io.Kfull + (size_t)kc * NXU) + (size_t)ri * NX;
#pragma unroll
        for (int c = 0; c < NX; ++c) r[c] = p[c];
        sv = *(is_a ? io.bcl + (size_t)kc * NX + ri : io.kff + (size_t)kc * NU + ri);
        bv = io.lqb[(size_t)kc * NX + ri];
        dv = io.gdt[kc];
        nv = io.base.nut[kc];
      };
      auto step = [&](const double (&r)[NX], double sv, double bv, double dv, int nv, int j) {      // j: stage index inside this pass
        const double* cur = hist + j * NX;
        double t0 = sv, t1 = 0.0;
#pragma unroll
        for (int c = 0; c < NX; c += 2) { t0 += r[c] * cur[c]; t1 += r[c + 1] * cur[c + 1]; }
        const double t = t0 + t1;
        const double du = (is_k && nv > 0) ? t : 0.0;                          // event node: no input
        // force inputs of the component l % 3 (lanes 0..2): du_l + du_(l+3) + du_(l+6) + du_(l+9)
        const double f4 = du + __shfl(du, (l + 3) & 63) + __shfl(du, (l + 6) & 63) + __shfl(du, (l + 9) & 63);
        const double own = cur[is_k ? l : 0];
        double nxt;
        if (is_a) nxt = t;                                                      // rows 3..11: [Acl | bcl]
        else if (l < 3) nxt = own + bv + dv * dt_over_m_factor * f4;
        else nxt = own + bv + dv * du;                                          // joint rows (lanes 12..nx-1)
        if (is_a || l < 3 || (l >= 12 && l < NX)) hist[(j + 1) * NX + ri] = nxt;
        if (is_k) duh[j * NU + l] = du;
        lds_wave_sync();
      };
      load(rA, sA, bA, dA, nA, k0); load(rB, sB, bB, dB, nB, k0 + 1); load(rC, sC, bC, dC, nC, k0 + 2);
      for (int j = 0; j < nk; j += 3) {                 // steps past nk write history rows that are never read (cap + 4 rows; du: see below)
        step(rA, sA, bA, dA, nA, j);     load(rA, sA, bA, dA, nA, k0 + j + 3);
        if (j + 1 < nk) step(rB, sB, bB, dB, nB, j + 1);
        load(rB, sB, bB, dB, nB, k0 + j + 4);
        if (j + 2 < nk) step(rC, sC, bC, dC, nC, j + 2);
        load(rC, sC, bC, dC, nC, k0 + j + 5);
      }
    } else if (k0 == 0 && io.with_ls && tid < 2 * kWave) {
      linesearch_begin_wave<NJ>(scratch, io.ls, tid - kWave);
    }
    __syncthreads();
    // outputs of the pass (coalesced) and its share of the norms
    for (int idx = tid; idx < nk * NX; idx += NT) {
      const double d = hist[idx];                        // dx_(k0 + idx / nx): the Armijo metric and |dx| run over the stages 0..N-1 here, dx_N below
      const double dn = hist[NX + idx];
      io.base.dx[(size_t)(k0 + 1) * NX + idx] = dn;
      const double u = duh[idx];                         // nu == nx
      io.base.du[(size_t)k0 * NU + idx] = u;
      acc_u += u * u;
      acc_x += d * d;
      acc_arm += io.mvec[(size_t)k0 * NX + idx] * d;
      if (idx % NX == 0) acc_arm += io.mscal[k0 + idx / NX];
    }
    __syncthreads();
    if (tid < NX) {
      const double dl = hist[nk * NX + tid];             // the last state of the pass
      if (k0 + nk < N) hist[tid] = dl;                   // input of the next pass
      else acc_x += dl * dl;                             // dx_N
    }
    __syncthreads();
  }
  __shared__ double red3s[3][NT / kWave];
  for (int off = kWave / 2; off >= 1; off >>= 1) {
    acc_arm += __shfl_down(acc_arm, off);
    acc_x += __shfl_down(acc_x, off);
    acc_u += __shfl_down(acc_u, off);
  }
  if ((tid & (kWave - 1)) == 0) { red3s[0][tid / kWave] = acc_arm; red3s[1][tid / kWave] = acc_x; red3s[2][tid / kWave] = acc_u; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, x2 = 0.0, u2 = 0.0;
    for (int w = 0; w < NT / kWave; ++w) { a += red3s[0][w]; x2 += red3s[1][w]; u2 += red3s[2][w]; }
    io.base.summary[0] = a;
    io.base.summary[1] = x2;
    io.base.summary[2] = u2;
    io.base.summary[3] = (double)status;
  }
}


// Spin on an LDS word that another wave of the workgroup advances (the loads in flight of the spinning wave stay in flight: the fence orders LDS only)
__device__ __forceinline__ void lds_wait_ge(int* flag, int v) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Roll-out of the workgroup sweeps, round 6: du inside the recurrence (riccati_rollout_sparse: the joint rows of dx+ follow from du, so only the rows 0..11
// of [Acl | bcl] are read - 6.4 instead of 8.3 KB per stage - and no step-norm pass over K follows) fed through a RING in LDS.  With a CU per problem every
// workgroup reaches its roll-out at the same time: the phase was a chip-wide stream (Acl, then K: 2 x 104 MB at batch 256, ~4.5 TB/s) requested by one wave
// per CU, a row per lane - 22 cache lines per request.  Here wave 0 computes and touches no global memory; the waves 2 .. NW-1 stream chunks of C stages -
// K, the twelve rows of Acl, kff, bcl, b: each a contiguous piece, consecutive 16-byte units on consecutive lanes - into a double buffer, two chunks ahead
// in registers; an LDS word per loader wave counts the chunks it has stored, another the chunks consumed.  Wave 1 opens the line search as before.
template <int NJ, int NT, int C>
struct RolloutRing {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ, NXX = NX * NX, NXU = NX * NU;
  static constexpr int NW = NT / kWave, NL = NW - 2, NLT = NL * kWave;
  static constexpr int AR = 12 * NX;                                                   // rows 0..11 of Acl
  static constexpr int OK = 0, OA = OK + C * NXU, OF = OA + C * AR, OB = OF + C * NU, OL = OB + C * NX, SIZE = OL + C * NX;   // doubles of a buffer
  static constexpr int UT = SIZE / 2, UPL = (UT + NLT - 1) / NLT;                      // 16-byte units of a chunk, per loader lane
  static constexpr int kScratch = 256;                                                 // opening of the line search (197 doubles), the counters (200 ..), a slot nobody reads (240)
  static constexpr int kDt = kMaxRiccatiStages + 16;                                   // interval length per stage (negative: a stage without inputs)
  static constexpr int kTab = kDt + kMaxRiccatiStages / 8;                             // + a byte per stage: which of the two feet stand
  static constexpr int kFixed = kScratch + kTab + 2 * SIZE + (2 * C + 2) * NX;         // + the slack rows of the two histories (steps past the end of a pass store too)
  static_assert(C % 2 == 0 && C <= 14 && NX % 2 == 0 && NX == NU && NL >= 1, "16-byte units");
  static constexpr int cap(int lds_doubles) { return ((lds_doubles - kFixed) / (NX + NU)) & ~1; }
};
template <int NJ, int NT, int C>
__device__ __forceinline__ void riccati_rollout_ring(double* lds /* (2 cap + 2 C + 2) nx + RolloutRing::kFixed doubles */, int cap, int status, const RiccatiFastIO& io) {
  using RR = RolloutRing<NJ, NT, C>;
  constexpr int NX = RR::NX, NU = RR::NU, NXX = RR::NXX, NXU = RR::NXU, NL = RR::NL, NLT = RR::NLT, UPL = RR::UPL, UT = RR::UT;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int N = io.base.N;
  double* const hist = lds;                                      // dx_0 .. dx_cap (+ slack), row stride nx
  double* const duh = lds + (size_t)(cap + C + 2) * NX;          // du of the pass (+ slack), row stride nu
  double* const scratch = duh + (size_t)(cap + C) * NU;          // opening of the line search
  int* const flags = reinterpret_cast<int*>(scratch + 200);      // [1]: chunks consumed, [2 + i]: chunks loader wave i has stored its share of
  double* const dtab = scratch + RR::kScratch;
  unsigned char* const mtab = reinterpret_cast<unsigned char*>(dtab + RR::kDt);
  double* const ring = dtab + RR::kTab;
  for (int idx = tid; idx < N + C + 2 && idx < RR::kDt; idx += NT) dtab[idx] = (idx < N && io.base.nut[idx] > 0) ? io.gdt[idx] : -1.0;
  for (int idx = tid; idx < N && idx < kMaxRiccatiStages; idx += NT) mtab[idx] = (unsigned char)(io.base.nut[idx] > 0 ? (io.mode[idx] & 3) : 0);
  if (tid < NX) hist[tid] = io.base.dx0[tid];
  if (tid >= kWave && tid < kWave + NX) io.base.dx[tid - kWave] = io.base.dx0[tid - kWave];
  if (tid < 2 + NL) flags[tid] = 0;
  __syncthreads();
#ifdef BPMPC_RICCATI_PROFILE
  const long long tp1 = clock64();
#endif
  double acc_arm = 0.0, acc_x = 0.0, acc_u = 0.0;
  int cb = 0;                                             // chunks of the passes before this one
  for (int k0 = 0; k0 < N; k0 += cap) {
    const int nk = N - k0 < cap ? N - k0 : cap;
    const int nch = (nk + C - 1) / C;
    // what the Armijo terms behind the recurrence need from global memory: the first four elements per thread are requested before it
    double mv[4], ms[4];
    auto armijo_terms = [&](int i0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = i0 + e * NT + tid, ic = idx < nk * NX ? idx : 0;
        mv[e] = io.mvec[(size_t)k0 * NX + ic];
        ms[e] = io.mscal[k0 + ic / NX];
      }
    };
    armijo_terms(0);
    if (w == 0) {
      // A lone wave issues an instruction every 5 .. 8 cycles whatever it depends on: a step costs its instruction count.  The state never leaves the
      // registers (as in riccati_rollout_deep): every 16-lane DPP row holds all of it - xlo: element p in lane position p, xhi: element H + p - and a lane's
      // dot product takes its operands by `v_fmac_f64_dpp .. row_newbcast`.  Row 0 of the wave forms the elements of xlo, row 1 those of xhi, each lane in the
      // lane position of its element - from its row of [Acl | bcl] (elements 0..11) or, the joint rows, as dx + b + dt du with du from its row of
      // [K | kff] -, so one v_permlane32_swap + one v_permlane16_swap per half of the value broadcast the two rows to all four.  Row 2: the force inputs.
      constexpr int H = NX / 2;
      static_assert(H <= 12 && NX - H <= 16 && H >= 6, "elements 0..11 in row 0 and the head of row 1");
      const int q = l >> 4, p = l & 15;
      const int xe = q == 0 ? p : H + p;                                                         // element of dx+ a lane of the rows 0, 1 forms
      const bool x_lane = (q == 0 && p < H) || (q == 1 && xe < NX);
      const bool a_lane = x_lane && xe < 12;                                                      // from [Acl | bcl]
      const bool k_lane = (x_lane && !a_lane) || (q == 2 && p < 12);                              // forms an input
      const int ki = q == 2 ? p : xe;                                                            // its row of K
      const int row_off = a_lane ? RR::OA + xe * NX : (k_lane ? ki * NX : 0), row_str = a_lane ? RR::AR : NXU;
      const int sv_off = a_lane ? RR::OB + xe : RR::OF + (k_lane ? ki : 0);
      const int bv_off = RR::OL + (x_lane && !a_lane ? xe : 0);
      const double ownm = x_lane && !a_lane ? 1.0 : 0.0;
      // where a step stores: the element of dx+, the input; lanes that form none write a slot nobody reads (stride 0)
      int hp = x_lane ? NX + xe : (int)(scratch + 240 - hist), up = k_lane ? (int)(duh - hist) + ki : (int)(scratch + 242 - hist);
      const int hs = x_lane ? NX : 0, us = k_lane ? NU : 0;
      double xlo = hist[p < H ? p : 0], xhi = hist[H + p < NX ? H + p : 0];
      double rA[NX], rB[NX], sA, sB, bA, bB, dA, dB;
      // (LDS loads return in order: the scalars of a stage first, so that a step never waits for the row of the NEXT stage that is requested ahead of it)
      auto load = [&](double (&r)[NX], double& sv, double& bv, double& dv, const double* buf, int s, int j) {
        dv = dtab[k0 + j];
        sv = buf[sv_off + s * NX];
        bv = buf[bv_off + s * NX];
        const double2* pr = reinterpret_cast<const double2*>(buf + row_off + s * row_str);
#pragma unroll
        for (int c = 0; c < NX / 2; ++c) { const double2 v = pr[c]; r[2 * c] = v.x; r[2 * c + 1] = v.y; }
      };
      auto step = [&](const double (&r)[NX], double sv, double bv, double dv) {
        const bool keep = a_lane || dv >= 0.0;                                  // a stage without inputs (an event node): du = 0
        const double dvl = a_lane ? 1.0 : dv, bvl = a_lane ? 0.0 : bv;          // dx+ = t for a row of Acl, own element + b + dt du for a joint row
        const double base = __builtin_fma(xhi, ownm, bvl);
        constexpr int H2 = H / 2;
        double t = sv, t2 = 0.0, t3 = 0.0, t4 = 0.0;                            // four chains; the first DPP read of xlo / xhi two wait states behind their VALU write
        roll_fma<0, true>(t, xlo, r[0]);
        roll_fma<0, false>(t2, xhi, r[H]);
        roll_fma<H2, false>(t3, xlo, r[H2]);
        roll_fma<H2, false>(t4, xhi, r[H + H2]);
        RollDot2<NX, H, 1, H2>::run(t, t2, xlo, xhi, r);
        RollDot2<NX, H, H2 + 1, H>::run(t3, t4, xlo, xhi, r);
        t = (t + t2) + (t3 + t4);
        t = keep ? t : 0.0;
        const double nxt = __builtin_fma(dvl, t, base);
        hist[hp] = nxt; hp += hs;
        hist[up] = t; up += us;
        const auto l32 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(nxt), (unsigned)__double2loint(nxt), false, false);
        const auto h32 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(nxt), (unsigned)__double2hiint(nxt), false, false);
        // [0]: rows 0, 1 of the wave in both halves
        const auto l01 = __builtin_amdgcn_permlane16_swap(l32[0], l32[0], false, false), h01 = __builtin_amdgcn_permlane16_swap(h32[0], h32[0], false, false);
        // [0]: row 0 in all four rows, [1]: row 1
        xlo = __hiloint2double((int)h01[0], (int)l01[0]);
        xhi = __hiloint2double((int)h01[1], (int)l01[1]);
      };
      // a loader wave may be a chunk ahead of another: one word each.  The words of the NEXT chunk are read a step before its first row is requested
      // (beside the last step but one of this chunk), so that neither the poll nor that request sits between two steps.
      int* const f = flags + 2 + (l < NL ? l : 0);
      auto filled = [&]() { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
      auto await = [&](int fv, int g) {
        while (__builtin_amdgcn_ballot_w64(fv < g + 1) != 0) { __builtin_amdgcn_s_sleep(1); fv = filled(); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      };
      await(filled(), cb);
      load(rA, sA, bA, dA, ring + (cb & 1) * RR::SIZE, 0, 0);
      for (int c = 0; c < nch; ++c) {
        const double* buf = ring + ((cb + c) & 1) * RR::SIZE;
        const double* nbuf = ring + ((cb + c + 1) & 1) * RR::SIZE;
        const int j0 = c * C;
        const bool more = c + 1 < nch;
#pragma unroll
        for (int s = 0; s < C; s += 2) {
          load(rB, sB, bB, dB, buf, s + 1, j0 + s + 1);
          int fv = 0;
          if (s + 2 == C && more) fv = filled();
          step(rA, sA, bA, dA);
          if (s + 2 < C) load(rA, sA, bA, dA, buf, s + 2, j0 + s + 2);
          else if (more) { await(fv, cb + c + 1); load(rA, sA, bA, dA, nbuf, 0, j0 + C); }
          step(rB, sB, bB, dB);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (l == 0) __hip_atomic_store(&flags[1], cb + c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if (w >= 2) {
      const int lt = (w - 2) * kWave + l;
      // unit u of a chunk (its doubles 2 u, 2 u + 1 of the buffer): which array, which stage of the chunk, where in the stage
      // A force component of a foot in the air has no gain (K row = 0, written by the sweep as such): its units are requested from a page of zeros instead -
      // same instructions, another address, 6 of the 34 rows of a single-support stage that never leave the memory
      const double* arr[UPL]; int str[UPL], st[UPL], off[UPL], foot[UPL];
      const double* const zeros = io.zero_one + 4;
#pragma unroll
      for (int j = 0; j < UPL; ++j) {
        const int u = lt + j * NLT, d = 2 * u;
        foot[j] = 0;
        if (d < RR::OA)        { arr[j] = io.Kfull; str[j] = NXU; st[j] = d / NXU;               off[j] = d % NXU; foot[j] = off[j] < 6 * NX ? 1 : (off[j] < 12 * NX ? 2 : 0); }
        else if (d < RR::OF)   { arr[j] = io.Acl;   str[j] = NXX; st[j] = (d - RR::OA) / RR::AR; off[j] = (d - RR::OA) % RR::AR; }
        else if (d < RR::OB)   { arr[j] = io.kff;   str[j] = NU;  st[j] = (d - RR::OF) / NU;     off[j] = (d - RR::OF) % NU; }
        else if (d < RR::OL)   { arr[j] = io.bcl;   str[j] = NX;  st[j] = (d - RR::OB) / NX;     off[j] = (d - RR::OB) % NX; }
        else if (d < RR::SIZE) { arr[j] = io.lqb;   str[j] = NX;  st[j] = (d - RR::OL) / NX;     off[j] = (d - RR::OL) % NX; }
        else                   { arr[j] = io.lqb;   str[j] = NX;  st[j] = 0;                     off[j] = 0; }             // no unit: a valid address, nothing stored
      }
      double ax[UPL], ay[UPL], bx[UPL], by[UPL];
      auto issue = [&](double (&vx)[UPL], double (&vy)[UPL], int c) {
        const int ks = k0 + c * C;
#pragma unroll
        for (int j = 0; j < UPL; ++j) {
          int k = ks + st[j];
          k = k < N ? k : N - 1;                         // beyond the end: a valid, unused stage
          const bool air = (foot[j] & ~mtab[k]) != 0;
          const double2 v = *reinterpret_cast<const double2*>(air ? zeros : arr[j] + (size_t)k * str[j] + off[j]);
          vx[j] = v.x; vy[j] = v.y;
        }
      };
      auto put = [&](const double (&vx)[UPL], const double (&vy)[UPL], int c) {
        if (c >= nch) return;
        lds_wait_ge(&flags[1], cb + c - 1);              // the buffer's previous chunk has been consumed
        double* buf = ring + ((cb + c) & 1) * RR::SIZE;
#pragma unroll
        for (int j = 0; j < UPL; ++j)
          if ((j + 1) * NLT <= UT || lt + j * NLT < UT) { double2 v; v.x = vx[j]; v.y = vy[j]; *reinterpret_cast<double2*>(buf + 2 * (lt + j * NLT)) = v; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (l == 0) __hip_atomic_store(&flags[w], cb + c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // flags[2 + loader]
      };
      issue(ax, ay, 0); issue(bx, by, 1);
      for (int c = 0; c < nch; c += 2) {                // (the same requests in the same order on every path: the compiler can count what may stay in flight)
        put(ax, ay, c); issue(ax, ay, c + 2);
        put(bx, by, c + 1); issue(bx, by, c + 3);
      }
    } else if (k0 == 0 && io.with_ls) {
      linesearch_begin_wave<NJ>(scratch, io.ls, l);
    }
#ifdef BPMPC_RICCATI_PROFILE
    if (BPMPC_RICCATI_PROFILE == 1 && io.prof && tid == 0 && k0 == 0) io.prof[6] = (double)(clock64() - tp1);      // the recurrence of the first pass (wave 0)
#endif
    __syncthreads();
    cb += nch;
    // outputs of the pass (coalesced) and its share of the norms
    for (int i0 = 0; i0 < nk * NX; i0 += 4 * NT) {
      if (i0 > 0) armijo_terms(i0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = i0 + e * NT + tid;
        if (idx < nk * NX) {
          const double d = hist[idx];                    // dx_(k0 + idx / nx): the Armijo metric and |dx| run over the stages 0..N-1 here, dx_N below
          const double dn = hist[NX + idx];
          io.base.dx[(size_t)(k0 + 1) * NX + idx] = dn;
          const double u = duh[idx];                     // nu == nx
          io.base.du[(size_t)k0 * NU + idx] = u;
          acc_u += u * u;
          acc_x += d * d;
          acc_arm += mv[e] * d;
          if (idx % NX == 0) acc_arm += ms[e];
        }
      }
    }
    __syncthreads();
    if (tid < NX) {
      const double dl = hist[nk * NX + tid];             // the last state of the pass
      if (k0 + nk < N) hist[tid] = dl;                   // input of the next pass
      else acc_x += dl * dl;                             // dx_N
    }
    __syncthreads();
  }
  __shared__ double red3r[3][NT / kWave];
  for (int off = kWave / 2; off >= 1; off >>= 1) {
    acc_arm += __shfl_down(acc_arm, off);
    acc_x += __shfl_down(acc_x, off);
    acc_u += __shfl_down(acc_u, off);
  }
  if ((tid & (kWave - 1)) == 0) { red3r[0][tid / kWave] = acc_arm; red3r[1][tid / kWave] = acc_x; red3r[2][tid / kWave] = acc_u; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, x2 = 0.0, u2 = 0.0;
    for (int ww = 0; ww < NT / kWave; ++ww) { a += red3r[0][ww]; x2 += red3r[1][ww]; u2 += red3r[2][ww]; }
    io.base.summary[0] = a;
    io.base.summary[1] = x2;
    io.base.summary[2] = u2;
    io.base.summary[3] = (double)status;
  }
}

template <int NJ, bool DB, bool JW = true>     // JW: Wt holds its joint rows (off: the loaders complete them from Vt, PackedStageLoader JR)
__device__ __forceinline__ void riccati_mfma(RiccatiMfmaWorkspace<NJ, DB>& ws, const RiccatiFastIO& io) {
  using WS = RiccatiMfmaWorkspace<NJ, DB>;
  constexpr int NX = WS::NX, NU = WS::NU, NT = kRiccatiThreads, LDN = WS::LDN, LDW = WS::LDW, RB = WS::RB, ZR = WS::ZR, RCL = WS::RCL, RBM = WS::RBM;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  constexpr int KS = (NX + 3) / 4;          // k-steps over the state dimension
  constexpr int BC = NX + 1;                // first column of B~ / Pu / R~ in the packed layouts
  static_assert(NX == NU, "packed layouts assume nx == nu");
  static_assert(NX + 1 + NU <= kWave, "one lane per column of [H | G g]");
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int li = l & 15, lk = l >> 4;       // operand row/column index and k index of this lane
  const int N = io.base.N;

  const int k_top = (io.k_hi < N ? io.k_hi : N) - 1;
  const bool resumed = io.k_hi < N;
  {
    double* z = &ws.S[0][0];
    constexpr int total = (int)(sizeof(WS) / sizeof(double));
    for (int idx = tid; idx < total - 1; idx += NT) z[idx] = 0.0;   // every matrix and its padding (status is set below)
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
    ws.mode[idx] = (unsigned char)(n > 0 ? (io.mode[idx] & 3) : kModeEvent);      // (an event node has no inputs)
    too_wide |= (RBM < NU && n > RBM) ? 1 : 0;
  }
  if (RBM < NU && __syncthreads_or(too_wide)) {         // more reduced inputs than this variant holds: fail loudly (status 2 in bpmpc_stats)
    if (tid == 0) {
      if (io.k_lo > 0) io.carry[NXX + NX] = 1.0;
      else { io.base.summary[0] = 0.0; io.base.summary[1] = 0.0; io.base.summary[2] = 0.0; io.base.summary[3] = 1.0; }
    }
    // the roll-out, which normally opens the line search of this problem, is skipped: open it here, otherwise done / alpha / base keep the
    // previous iteration's values and k_ls_decide would skip the problem instead of reporting the failure (advisor r02)
    if (io.k_lo == 0 && io.with_ls && tid < kWave) linesearch_begin_wave<NJ>(&ws.S[0][0], io.ls, tid);
    return;
  }

  // Prefetch registers and staging of the loader waves (0..2): PackedStageLoader, pairs t, t + 192, ..
  constexpr int NLD = 3 * kWave;                       // loader threads
  const bool loader = w < 3;
  PackedStageLoader<NJ, NLD, (RBM < NU ? RBM : NU), LDW, LDN, false, !JW> ld;
  ld.init(io, tid, loader, (size_t)(k_top > 0 ? k_top : 0));
  if (loader && k_top >= io.k_lo) {     // the stage the loader's pointers stand on (the LDS copies of nut and mode may not be visible yet)
    const int kt = k_top > 0 ? k_top : 0, n0 = io.base.nut[kt];
    ld.prefetch(n0, n0 > 0 ? (io.mode[kt] & 3) : kModeEvent);
  }
  auto prefetch_next = [&](int k) { ld.prefetch(ws.nut[k - 1], ws.mode[k - 1]); };
  __syncthreads();
#ifdef BPMPC_RICCATI_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define RMPROF(slot) do { const long long tn_ = clock64(); tacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define RMPROF(slot) ((void)0)
#endif

  // Deferred stores.  vmcnt retires in order, so a wait for the prefetched registers also waits for every store issued
  // after the prefetch; stores issued at the end of a stage would put the HBM store latency on the critical path of the
  // next one.  The outputs of stage k are therefore held in registers and stored right after the staging barrier of
  // stage k-1, immediately before its prefetch: whatever the next wait covers is then a whole stage old.
  v4d held_acl = {0.0, 0.0, 0.0, 0.0}, held_kf = {0.0, 0.0, 0.0, 0.0};
  double held_m = 0.0;
  int held_k = -1;
  auto flush_held = [&]() {
    if (held_k < 0) return;                            // uniform
    const int r0 = 16 * (w >> 1), c0 = 16 * (w & 1);
    double* Acl = io.Acl + (size_t)held_k * NXX;
    double* Kf = io.Kfull + (size_t)held_k * NXU;
    const int col = c0 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = r0 + lk + 4 * r;
      if (rr < NX) {
        if (col < NX) { Acl[rr * NX + col] = held_acl[r]; Kf[rr * NX + col] = held_kf[r]; }
        else if (col == NX) { io.bcl[(size_t)held_k * NX + rr] = held_acl[r]; io.kff[(size_t)held_k * NU + rr] = held_kf[r]; }
      }
    }
    if (w == 3 && l < NX) io.mvec[(size_t)held_k * NX + l] = held_m;
    if (w == 3 && l == NX) io.mscal[held_k] = held_m;
  };

  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = ws.nut[k];        // max_nodes <= kMaxRiccatiStages is checked when the solver is created
    const int cur = DB ? (k & 1) : 0;
    double (*const W)[LDW] = ws.W[cur];
    double (*const PW)[LDW] = ws.PW[cur];
    double (*const Qq)[LDN] = ws.Qq[cur];
    double (*const M)[LDW] = ws.M[cur];
    double* const rvec = ws.r[cur];
    const int ksn = (nt + 3) >> 2;                   // k-steps over the reduced input
    const int nbc = (BC + nt + 15) >> 4;             // block columns of the packed width nx + 1 + nt
    const int ntb = (nt + 15) >> 4;                  // block rows of the reduced input
    // ---- P0: registers -> packed LDS layouts; what the projection kernel does not write (block columns >= nbc, rows >= nt) is staged as zero
    if (loader) ld.stage(&W[0][0], &PW[0][0], &Qq[0][0], &M[0][0], rvec, nt);
    lds_barrier();
    RMPROF(0);
    RMPROF(1);
    // ---- P1: SW = sym(S) W, s added to the b column
    for (int id = w; id < 2 * nbc; id += 4) {
      const int bi = id >= nbc ? 1 : 0;
      const int r0 = 16 * bi, c0 = 16 * (id - bi * nbc);
      const int row = r0 + li;
      // unconditional operand loads (a predicated load compiles to a branch plus a full LDS wait per k-step): rows >= nx
      // are switched off by the factor, k >= nx meets the zero rows of W
      const double half = row < NX ? 0.5 : 0.0;
      const int rowc = RCL >= 32 ? row : (row < RCL ? row : ZR);
      // all operands first, then the chain of dependent MFMAs back to back (interleaved, every k-step exposes an LDS round trip)
      double a[KS], b[KS], sv[4];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 4 * ks + lk;
        a[ks] = half * (lds1(ws.S[rowc][kk]) + lds1(ws.S[kk][rowc]));
        b[ks] = lds1(W[kk][c0 + li]);
      }
      const double smask = (c0 + li == NX) ? 1.0 : 0.0;              // s rides in the b column; rows >= nx of S are zero
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int rr = r0 + lk + 4 * r; sv[r] = smask * lds1(ws.S[RCL >= 32 ? rr : (rr < RCL ? rr : ZR)][NX]); }
      __builtin_amdgcn_sched_barrier(0);
      v4d acc = {sv[0], sv[1], sv[2], sv[3]};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
      blk_store<LDW, RB>(&ws.SW[0][0], r0, c0, l, acc);
    }
    lds_barrier();
    RMPROF(2);
    // ---- P2: [G | g | H] = [P | r | R] + B' SW
    for (int id = w; id < ntb * nbc; id += 4) {
      const int bi = id >= nbc ? 1 : 0;
      const int r0 = 16 * bi, c0 = 16 * (id - bi * nbc);
      v4d acc = blk_load<LDW, RCL, ZR>(&M[0][0], r0, c0, l);
      // compact rows: input rows >= nu do not exist, read the (zero) last padding column of W instead
      const int bcol = (RCL >= 32 || r0 + li < NU) ? BC + r0 + li : LDW - 1;
      double a[KS], b[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 4 * ks + lk;
        a[ks] = lds1(W[kk][bcol]);                                 // B'(i, kk); columns >= nt and rows >= nx of B~ are zero
        b[ks] = lds1(ws.SW[kk][c0 + li]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
      blk_store<LDW, RBM>(&M[0][0], r0, c0, l, acc);
      if (c0 < 32) blk_store<LDN, RB>(&ws.G0[0][0], r0, c0, l, acc);
    }
    if (w == 3) flush_held();        // (wave 3 runs the elimination in P3, the other waves store and prefetch there)
    lds_barrier();
    RMPROF(3);
    // ---- P3: Y = H^-1 [G g] by Gauss-Jordan in registers on wave 3 (and wave 2 when one wave cannot hold all right-hand sides);
    //          the other waves: [Sn | sn] = [Q | q] + A' SW(:, 0..nx)
    // row layout of the elimination (riccati_fast.h): 16 - nt right-hand sides per 16-lane row, 4 rows per wave
    const int rpr = 16 - nt;
    const int gj_waves = 0;
    const int sn_waves = gj_waves == 2 ? 2 : 3;
    if (w == 3 && gj_waves == 0) {
      const int col = l < nt ? BC + l : l - nt;
      const bool used = l < nt + NX + 1;
      bool ok;
#define BP_GJ_CASE(ROWS)                                                                      \
      {                                                                                       \
        double v[ROWS];                                                                       \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) v[i] = (used && i < nt) ? M[i][col] : 0.0; \
        ok = gauss_jordan_wave<ROWS>(v, nt);                                                  \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) if (used && i < nt && l >= nt) M[i][col] = v[i]; \
      }
      // the elimination is the longest dependent chain of a stage: instantiate it for the actual number of rows
      if (nt <= 8) BP_GJ_CASE(8)
      else if (nt == 9) BP_GJ_CASE(9)          // single support of this robot class: 14 rows of rank 13
      else if (nt <= 10) BP_GJ_CASE(10)
      else if (nt <= 12) BP_GJ_CASE(12)
      else BP_GJ_CASE(NU)
#undef BP_GJ_CASE
      if (l == 0 && !ok) ws.status = 1;
    } else if (w >= 4 - gj_waves && gj_waves > 0) {
      if (w != 3) {                    // wave 2 is a loader: its share of the global memory traffic first
        flush_held();
        if (loader && k > io.k_lo) prefetch_next(k);
      }
      const int c16 = l & 15;
      const int rhs = ((3 - w) * 4 + (l >> 4)) * rpr + (c16 - nt);      // right-hand side of this lane (lanes >= nt of the row)
      const bool is_h = c16 < nt, is_rhs = !is_h && rhs < NX + 1;
      const int col = is_h ? BC + c16 : (is_rhs ? rhs : 0);
      bool ok;
      // (no masks on the loads - as conditional expressions they are a branch and a full wait each -: a lane without a column eliminates column 0 and
      //  stores nothing, rows >= nt only ever change themselves)
#define BP_GJ_CASE(ROWS, EXACT)                                                               \
      {                                                                                       \
        double v[ROWS];                                                                       \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) v[i] = lds1(M[i][col]);              \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) asm volatile("" : "+v"(v[i]));       \
        lds_wave_sync();             /* every lane has its column before any right-hand side is overwritten (two waves: disjoint ones) */ \
        ok = gauss_jordan_rows<ROWS, EXACT>(v, nt);                                           \
        _Pragma("unroll") for (int i = 0; i < ROWS; ++i) if (is_rhs && i < nt) M[i][col] = v[i]; \
      }
      if (nt <= 8) BP_GJ_CASE(8, false)
      else if (nt == 9) BP_GJ_CASE(9, true)    // single support of this robot class: 14 rows of rank 13
      else if (nt <= 10) BP_GJ_CASE(10, true)
      else BP_GJ_CASE(12, false)
#undef BP_GJ_CASE
      if (l == 0 && !ok) ws.status = 1;
    } else {
      // global memory traffic of the stage, off the critical path (the elimination is the long pole of P3) and
      // as early as possible: outputs of the previous stage, then the operands of the next one
      flush_held();
      if (loader && k > io.k_lo) prefetch_next(k);     // never beyond the chunk: earlier stages may not be projected yet
      for (int id = w; id < 4; id += sn_waves) {
        const int r0 = 16 * (id >> 1), c0 = 16 * (id & 1);
        v4d acc = blk_load<LDN, RCL, ZR>(&Qq[0][0], r0, c0, l);
        const int acol = r0 + li < NX ? r0 + li : LDW - 1;             // the last padding column of W is always zero
        double a[KS], b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int kk = 4 * ks + lk;
          a[ks] = lds1(W[kk][acol]);                                // A'(i, kk)
          b[ks] = lds1(ws.SW[kk][c0 + li]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
        blk_store<LDN, RB>(&ws.Sn[0][0], r0, c0, l, acc);
      }
    }
#ifdef BPMPC_RICCATI_PROFILE
    { const long long tn_ = clock64(); if (w == 0) tacc[6] += tn_ - tprev; if (w == 3) tacc[7] += tn_ - tprev; }   // own work of P3, before the barrier
#endif
    lds_barrier();
    RMPROF(4);
    // ---- P4: wave w owns block w of each result: [S | s] first (the next stage waits for it), then [Acl | bcl], [K | kff]
    {
      const int r0 = 16 * (w >> 1), c0 = 16 * (w & 1);
      const int row = r0 + li;
      v4d acc = blk_load<LDN, RCL, ZR>(&ws.Sn[0][0], r0, c0, l);
      v4d acl = blk_load<LDW, RCL, ZR>(&W[0][0], r0, c0, l);
      v4d kf = blk_load<LDW, RCL, ZR>(&PW[0][0], r0, c0, l);
      const int gcol = row < NX ? row : LDN - 1;                       // the last padding column of G0 is always zero
      const int rowc = RCL >= 32 ? row : (row < RCL ? row : ZR);
      if (ksn <= 3) {
        // up to 12 reduced inputs (every reference configuration): three k-steps, operands first, then the MFMAs of the
        // three independent accumulators; the rows nt.. of Y and G are zero, so a surplus k-step adds nothing
        double ag[3], ab[3], ap[3], yb[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int kk = 4 * ks + lk;
          yb[ks] = lds1(M[kk][c0 + li]);
          ag[ks] = -lds1(ws.G0[kk][gcol]);                                  // -G'(i, kk)
          ab[ks] = -lds1(W[rowc][BC + kk]);                             // -B(i, kk); rows >= nx of W and PW are zero
          ap[ks] = -lds1(PW[rowc][BC + kk]);                             // -Pu(i, kk)
        }
        __builtin_amdgcn_sched_barrier(0);
        const int nks = ksn == 3 ? 3 : 2;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          if (ks < nks) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[ks], yb[ks], acc, 0, 0, 0);
            acl = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[ks], yb[ks], acl, 0, 0, 0);
            kf = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[ks], yb[ks], kf, 0, 0, 0);
          }
        }
      } else {
        for (int ks = 0; ks < ksn; ++ks) {
          const int kk = 4 * ks + lk;
          const double yv = M[kk][c0 + li];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-ws.G0[kk][gcol], yv, acc, 0, 0, 0);
          acl = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[rowc][BC + kk], yv, acl, 0, 0, 0);
          kf = __builtin_amdgcn_mfma_f64_16x16x4f64(-PW[rowc][BC + kk], yv, kf, 0, 0, 0);
        }
      }
      blk_store<LDN, RB>(&ws.S[0][0], r0, c0, l, acc);    // S was last read in P1, three barriers ago
      // m = q~ - Y' r~ (Kt = -Y), m0 = -r~' H^-1 g
      double mt = 0.0;
      if (w == 3 && l <= NX) {
        mt = l < NX ? Qq[l][NX] : 0.0;
        if (nt <= 12) {
#pragma unroll
          for (int i = 0; i < 12; ++i) mt -= M[i][l] * rvec[i];     // rows >= nt of Y are zero
        } else {
          for (int i = 0; i < nt; ++i) mt -= M[i][l] * rvec[i];
        }
      }
      // results of this stage stay in registers; they are stored after the staging barrier of the next stage (see there)
      held_acl = acl; held_kf = kf; held_m = mt; held_k = k;
    }
    RMPROF(5);
    // double buffered: no barrier here, the next stage stages into the other buffer set and its staging barrier also orders S
    if (!DB) lds_barrier();
  }
  flush_held();
#ifdef BPMPC_RICCATI_PROFILE
  if (io.prof && tid == 0)
    for (int i = 0; i < 7; ++i) io.prof[i] = (double)tacc[i];
  if (io.prof && tid == 3 * kWave) io.prof[7] = (double)tacc[7];
#endif
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
    constexpr int kHistCap = ((int)(sizeof(WS) / sizeof(double)) - kStepNormsScratch * kRiccatiThreads / kWave) / NX - 8;
    static_assert(kHistCap >= 64, "roll-out history");
    riccati_rollout_deep<NJ>(reinterpret_cast<double*>(&ws), kHistCap, st, io);
  }
}

}  // namespace bpmpc
