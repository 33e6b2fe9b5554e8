// Shared pieces of the fast Riccati sweep (HIP only; the sweep itself is riccati_mfma.h, the lane-emulated reference
// version is riccati.h - same mathematics): the view of the per-stage data, LDS-only barriers, the Gauss-Jordan
// elimination held in the registers of one wave, and the step norms after the roll-out (riccati_mfma.h).
//   * the backward sweep stores the closed-loop quantities the roll-out needs
//        Acl = A~ + B~ Kt, bcl = b~ + B~ kt, K = Px + Pu Kt, kff = Pe + Pu kt, m = q~ + Kt' r~, m0 = r~' kt
//     so the roll-out is one mat-vec per stage (dx+ = Acl dx + bcl); du = K dx + kff is done afterwards for all
//     stages in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "../device_model.h"
#include "riccati.h"
#include "project_node.h"
#include "linesearch.h"

namespace bpmpc {

struct RiccatiFastIO {
  RiccatiIO base;            // same views as the reference kernel (Kt/kt unused; At .. rt unused: the projected model comes packed)
  const double *Wt, *Qp, *Mt; // per node PackedLq<NJ>::W_SIZE, Q_SIZE, M_SIZE: [At | bt | Bt], [Qt | qt], [Pt | rt | Rt] (project_node.h)
  const double* Vt;          // per node NJ * PackedLq<NJ>::WP: the joint rows of [Px | Pe | Pu], packed (project_lu_s.h); base.Pe: its Pe column incl. the force rows
  // folded change of variables (riccati_fold8.h): what the lineariser left of the node's LQ model (per node nx*nx, nx*nu, nx, nx, nu, kQrdStride),
  // the interval lengths of the problem's grid, the model constants
  const double *lqA, *lqB, *lqb, *lqq, *lqr, *qrd, *gdt;
  const DeviceModel* model;
  const double* zero_one;    // {0.0, 1.0} in global memory (what a loader lane loads for the zeros and ones of the generated force rows)
  const int* mode;           // per node: contact mode of the stage (its low two bits), from which the force rows of [Px | Pe | Pu] are generated
  double *Acl, *bcl;         // per node NX*NX, NX
  double *kff;               // per node NU
  double *mvec, *mscal;      // per node NX, 1
  double* Kfull;             // per node NU*NX (always materialised: it is the feedback gain K)
  double* prof;              // optional [8] per problem: accumulated cycles per phase (debug)
  // The horizon can be swept in chunks (one launch each, latest stages first) so that the sweep of one chunk overlaps
  // with the linearisation / projection of the earlier stages: this launch covers the stages [k_lo, k_hi) and hands
  // S, s and the status to the next one through `carry` (NX*NX + NX + 1 doubles).  The roll-out runs when k_lo == 0.
  int k_lo, k_hi;
  double* carry;
  double reg;                // settings.reg_prim: the terminal value function starts at reg * I (every other stage gets it from the projection kernel)
  // linesearch_begin of the problem (PerformanceIndex of the linearised iterate, alpha = 1, done flag) needs nothing of the sweep:
  // a wave that idles during the serial roll-out does it, instead of a launch of its own behind this kernel
  bool with_ls;
  ProblemLS ls;
};

// Workgroup barrier that orders LDS traffic only: outstanding global loads (the prefetch) and stores stay in flight.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void lds_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// linesearch_begin (linesearch.h) on one wavefront of a larger workgroup: the same sums in the same order
template <int NJ>
__device__ __forceinline__ void linesearch_begin_wave(double* partial /*kWave*3 + 5 LDS*/, const ProblemLS& p, int lane) {
  constexpr int NX = 12 + NJ;
  {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = lane; k < p.n_nodes; k += kWave) { a += p.node_perf[3 * k]; b += p.node_perf[3 * k + 1]; c += p.node_perf[3 * k + 2]; }
    if (lane < NX) { const double d = p.x0[lane] - p.x[lane]; b += d * d; }
    partial[lane] = a; partial[kWave + lane] = b; partial[2 * kWave + lane] = c;
  }
  lds_wave_sync();
  if (lane < 3) {
    double s = 0.0;
    for (int i = 0; i < kWave; ++i) s += partial[lane * kWave + i];
    partial[3 * kWave + 2 + lane] = s;
  }
  lds_wave_sync();
  if (lane == 0) {
    p.base[0] = partial[3 * kWave + 2]; p.base[1] = partial[3 * kWave + 3]; p.base[2] = partial[3 * kWave + 4];
    p.alpha[0] = 1.0;
    const bool run = p.active[0] != 0;
    p.done[0] = run ? 0 : 1;
    if (run) atomicAdd(p.remaining, 1);
  }
}
// LDS operand loads of the workgroup sweeps.  The compiler pairs neighbouring 8-byte LDS loads into ds_read2_b64, which costs 8 LDS-array cycles per
// pair and is banked mod 32 in 16-lane groups (a lane-per-row access with an even row stride is 2-way conflicted there); ds_read_b64 costs 2 cycles
// each and is banked mod 64 in 32-lane groups (MI355X_MICROARCH.md, LDS table).  A volatile load through an LDS-address-space pointer cannot be
// paired (a volatile GENERIC pointer would become a flat load).
__device__ __forceinline__ double lds1(const double& r) {
  typedef const volatile __attribute__((address_space(3))) double* lds_cvp;
  return *(lds_cvp)(&r);
}
struct d2 { double x, y; };
__device__ __forceinline__ d2 lds_pair(const double* p) {  // 16-byte aligned pair
  const double2 v = *reinterpret_cast<const double2*>(p);
  return d2{v.x, v.y};
}

// 1 / x from v_rcp_f64 and two Newton steps (full double precision for normal x; the IEEE division sequence is several
// times longer and sits on the critical path of every elimination step)
__device__ __forceinline__ double fast_reciprocal(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// Gauss-Jordan on the columns held by the lanes of one wave: lane c owns column c of [H | G g] in v[0..ROWS), ROWS >= nt;
// afterwards the lanes >= nt hold H^-1 times their column.  Returns false on a non-positive pivot.
template <int ROWS>
__device__ __forceinline__ bool gauss_jordan_wave(double (&v)[ROWS], int nt) {
  bool ok = true;
#pragma unroll
  for (int p = 0; p < ROWS; ++p) {
    if (p < nt) {  // wave-uniform
      double f[ROWS];
#pragma unroll
      for (int i = 0; i < ROWS; ++i) f[i] = readlane_f64(v[i], p);
      ok = ok && (f[p] > 0.0);
      const double row = v[p] * fast_reciprocal(f[p]);
#pragma unroll
      for (int i = 0; i < ROWS; ++i) v[i] = (i == p) ? row : v[i] - f[i] * row;
    }
  }
  return ok;
}

// The same elimination with the pivot column broadcast by DPP instead of v_readlane.  gfx90a+ DP-ALU DPP knows one control,
// row_newbcast:P - lane P of every 16-lane row to all lanes of that row - and v_fmac_f64 takes it on src0.  Layout: every
// row of 16 lanes holds the nt <= 16 columns of H in its lanes 0..nt-1 (four identical copies in the wave) and 16 - nt columns
// of the right-hand side [G g] in the others, so lane p of the own row always holds the pivot column and one
//     v[i] += bcast_p(v[i]) * (-row)                      (one instruction, no SGPR round trip)
// replaces two v_readlane_b32 and an FMA: 4 (16 - nt) right-hand sides per wave, ~1/2 of the issue cycles per pivot.
// Same operations on the same values as gauss_jordan_wave: bit-identical results.
template <int P>
__device__ __forceinline__ double row_bcast(double v) {       // v_mov_b64_dpp; hazards are the compiler's
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + P, 0xf, 0xf, true);
}
// acc -= bcast_P(acc) * m.  "VALU writes a VGPR -> DPP reads it" needs two wait states and the hazard recogniser does not look
// into inline assembly: FIRST = the statement opens a pivot step (s_nop 1; as expensive as an FMA, tools/probes/dpp_f64_probe.hip);
// the other updates of a step read registers that were last written a whole step earlier.
template <int P, bool FIRST>
__device__ __forceinline__ void fnmac_row_bcast(double& acc, double m) {
  if constexpr (FIRST) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(P));
  else asm volatile("v_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(P));
}
// Round 5: one pivot step of the Gauss-Jordan elimination as ONE assembly statement (why: forward_eliminate_rows_step below).  %0: the next pivot row, %1 ..: the ROWS - 2 other rows; the reciprocal chain of the next pivot (broadcast, v_rcp_f64, two Newton
// steps, product - the operations of fast_reciprocal) stands between the updates.  Same operations on the same values: bit-identical results.
#define BP_GJ_U(i) "v_fmac_f64_dpp %" #i ", %" #i ", -%[row] row_newbcast:%[p] row_mask:0xf bank_mask:0xf\n\t"
#define BP_GJ_HEAD "s_nop 1\n\t" BP_GJ_U(0) BP_GJ_U(1) BP_GJ_U(2) "v_mov_b64_dpp %[piv], %0 row_newbcast:%[pn] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_rcp_f64 %[r], %[piv]\n\t" BP_GJ_U(3) "v_fma_f64 %[e], -%[piv], %[r], 1.0\n\t" BP_GJ_U(4) "v_fmac_f64 %[r], %[r], %[e]\n\t" BP_GJ_U(5) \
  "v_fma_f64 %[e], -%[piv], %[r], 1.0\n\t" BP_GJ_U(6) "v_fmac_f64 %[r], %[r], %[e]\n\t"
#define BP_GJ_MUL "v_mul_f64 %[nr], %0, %[r]\n\t"
#define BP_GJ_OUT [piv] "=&v"(piv), [r] "=&v"(r), [e] "=&v"(e), [nr] "=&v"(next_row)
#define BP_GJ_IN [row] "v"(row), [p] "n"(P), [pn] "n"(PN)
#define BP_GJ_O(j) "+v"(v[(j) < P ? (j) : ((j) + 2 < ROWS ? (j) + 2 : 0)])      /* j-th row that is neither P nor P + 1 */
template <int ROWS, int P, bool EXACT>
__device__ __forceinline__ void gauss_jordan_rows_step(double (&v)[ROWS], int nt, bool& ok, int& minhi, double& lastp, double row) {
  constexpr bool has_next = P + 1 < ROWS;
  constexpr int PN = has_next ? P + 1 : P;
  double piv = 1.0, next_row = 0.0;
  if constexpr (has_next) {
    double r, e;
    static_assert(ROWS == 8 || ROWS == 9 || ROWS == 10 || ROWS == 12, "gauss_jordan_rows: row counts with a statement");
    if constexpr (ROWS == 8)
      asm volatile(BP_GJ_HEAD BP_GJ_MUL : "+v"(v[PN]), BP_GJ_O(0), BP_GJ_O(1), BP_GJ_O(2), BP_GJ_O(3), BP_GJ_O(4), BP_GJ_O(5), BP_GJ_OUT : BP_GJ_IN);
    else if constexpr (ROWS == 9)
      asm volatile(BP_GJ_HEAD BP_GJ_U(7) BP_GJ_MUL : "+v"(v[PN]), BP_GJ_O(0), BP_GJ_O(1), BP_GJ_O(2), BP_GJ_O(3), BP_GJ_O(4), BP_GJ_O(5), BP_GJ_O(6), BP_GJ_OUT : BP_GJ_IN);
    else if constexpr (ROWS == 10)
      asm volatile(BP_GJ_HEAD BP_GJ_U(7) BP_GJ_MUL BP_GJ_U(8)
                   : "+v"(v[PN]), BP_GJ_O(0), BP_GJ_O(1), BP_GJ_O(2), BP_GJ_O(3), BP_GJ_O(4), BP_GJ_O(5), BP_GJ_O(6), BP_GJ_O(7), BP_GJ_OUT : BP_GJ_IN);
    else
      asm volatile(BP_GJ_HEAD BP_GJ_U(7) BP_GJ_MUL BP_GJ_U(8) BP_GJ_U(9) BP_GJ_U(10)
                   : "+v"(v[PN]), BP_GJ_O(0), BP_GJ_O(1), BP_GJ_O(2), BP_GJ_O(3), BP_GJ_O(4), BP_GJ_O(5), BP_GJ_O(6), BP_GJ_O(7), BP_GJ_O(8), BP_GJ_O(9), BP_GJ_OUT : BP_GJ_IN);
  } else {
    fnmac_row_bcast<P, true>(v[0], row);
#pragma unroll
    for (int i = 1; i < ROWS - 1; ++i) fnmac_row_bcast<P, false>(v[i], row);
  }
  v[P] = row;
  if constexpr (has_next) {
    if constexpr (EXACT) {
      if constexpr (P + 2 == ROWS) lastp = piv; else minhi = min(minhi, __double2hiint(piv));     // (see forward_eliminate_rows_step)
      gauss_jordan_rows_step<ROWS, P + 1, EXACT>(v, nt, ok, minhi, lastp, next_row);
    } else if (P + 1 < nt) {  // wave-uniform
      ok = ok && (piv > 0.0);
      gauss_jordan_rows_step<ROWS, P + 1, EXACT>(v, nt, ok, minhi, lastp, next_row);
    }
  }
}
#undef BP_GJ_U
#undef BP_GJ_HEAD
#undef BP_GJ_MUL
#undef BP_GJ_OUT
#undef BP_GJ_IN
#undef BP_GJ_O
template <int ROWS, bool EXACT = false>
__device__ __forceinline__ bool gauss_jordan_rows(double (&v)[ROWS], int nt) {
  if (nt <= 0) return true;
  const double piv = row_bcast<0>(v[0]);
  bool ok = piv > 0.0;
  int minhi = 0x7fffffff;
  double lastp = piv;
  gauss_jordan_rows_step<ROWS, 0, EXACT>(v, nt, ok, minhi, lastp, v[0] * fast_reciprocal(piv));
  if constexpr (EXACT) ok = ok && minhi > 0 && lastp > 0.0;
  return ok;
}

// Forward elimination only, same row layout and pivot pipeline as gauss_jordan_rows: row P is divided by its pivot and
// eliminated from the rows below it; emit(P, z, y) sees pivot row P before (z) and after (y) the division.  The rows end
// up unit upper triangular in the H lanes; back_substitute_rows finishes H^-1 [G g].
// The reciprocal of pivot P + 1 is the dependent chain of a step (update of row P + 1, broadcast, v_rcp_f64, Newton, product:
// the wave is alone on its SIMD and waits for every link), so it takes ONE Newton step here: v_rcp_f64 is good to 4.6e-8,
// one step to 2.2e-15 relative, two are correctly rounded (tools/probes/rcp_probe.hip) - a perturbation of the pivot by ten
// ulp, the size of the rounding errors of the elimination itself.
// Round 5: a lone wave issues one FP64 instruction in ~8 cycles whether it depends on the one before or not (tools/probes/dep_probe.hip: dependent
// v_fma_f64 8.0, two independent chains 6.2 each), so a pivot step costs its INSTRUCTIONS.  Written as one statement per instruction (rounds 3, 4:
// ~150 instructions for nine pivots) a third of them computed nothing: the compiler puts a wait state in front of every statement that reads a
// register an earlier assembly statement defined (it does not count the assembly statements in between), and every pivot had a compare, a
// scalar AND and a uniform branch.  Now a step is ONE assembly statement in a fixed order - the update of the next pivot row %0, two more
// updates (the two wait states a DPP read needs behind the VALU write of its register), broadcast, v_rcp_f64, an update (the wait state of a
// transcendental result), Newton step, product, the other updates (%1 ..: NO of them); s_nop only where the last steps have no updates left to
// fill with - and with EXACT (nt == ROWS: the instantiations for 9 and 10 reduced inputs) the positivity of the pivots is one integer minimum per
// step, no branch.  Same operations on the same values as before: bit-identical results (tools/probes/elim_probe.hip: 2058 -> 1725 cycles, nt = 9).
// (every statement opens with two wait states: the compiler may copy an operand into its register right in front of the statement, and a DPP read
//  needs them behind a VALU write - seen: a v_mov_b64 in front of the update of the last step, wrong results)
#define BP_PIV_U(i) "v_fmac_f64_dpp %" #i ", %" #i ", -%[row] row_newbcast:%[p] row_mask:0xf bank_mask:0xf\n\t"
#define BP_PIV_MOV "v_mov_b64_dpp %[piv], %0 row_newbcast:%[pn] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define BP_PIV_RCP "v_rcp_f64 %[r], %[piv]\n\t"
#define BP_PIV_FMA "v_fma_f64 %[e], -%[piv], %[r], 1.0\n\t"
#define BP_PIV_FMAC "v_fmac_f64 %[r], %[r], %[e]\n\t"
#define BP_PIV_MUL "v_mul_f64 %[nr], %0, %[r]\n\t"
#define BP_PIV_OUT [piv] "=&v"(piv), [r] "=&v"(r), [e] "=&v"(e), [nr] "=&v"(next_row)
#define BP_PIV_IN [row] "v"(row), [p] "n"(P), [pn] "n"(PN)
template <int ROWS, int P, int PN>
__device__ __forceinline__ void pivot_step_asm(double (&v)[ROWS], double row, double& piv, double& next_row) {
  constexpr int NO = ROWS - P - 2;
  double r, e;
  auto& a = v;
  constexpr int B = P + 2;                        // first of the other rows
  if constexpr (NO == 0)
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) "s_nop 1\n\t" BP_PIV_MOV BP_PIV_RCP "s_nop 0\n\t" BP_PIV_FMA BP_PIV_FMAC BP_PIV_MUL : "+v"(a[PN]), BP_PIV_OUT : BP_PIV_IN);
  else if constexpr (NO == 1)
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) BP_PIV_U(1) "s_nop 0\n\t" BP_PIV_MOV BP_PIV_RCP "s_nop 0\n\t" BP_PIV_FMA BP_PIV_FMAC BP_PIV_MUL : "+v"(a[PN]), "+v"(a[B < ROWS ? B : 0]), BP_PIV_OUT : BP_PIV_IN);
  else if constexpr (NO == 2)
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) BP_PIV_U(1) BP_PIV_U(2) BP_PIV_MOV BP_PIV_RCP "s_nop 0\n\t" BP_PIV_FMA BP_PIV_FMAC BP_PIV_MUL
                 : "+v"(a[PN]), "+v"(a[B < ROWS ? B : 0]), "+v"(a[B + 1 < ROWS ? B + 1 : 0]), BP_PIV_OUT : BP_PIV_IN);
  else if constexpr (NO == 3)
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) BP_PIV_U(1) BP_PIV_U(2) BP_PIV_MOV BP_PIV_RCP BP_PIV_U(3) BP_PIV_FMA BP_PIV_FMAC BP_PIV_MUL
                 : "+v"(a[PN]), "+v"(a[B < ROWS ? B : 0]), "+v"(a[B + 1 < ROWS ? B + 1 : 0]), "+v"(a[B + 2 < ROWS ? B + 2 : 0]), BP_PIV_OUT : BP_PIV_IN);
  else if constexpr (NO == 4)
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) BP_PIV_U(1) BP_PIV_U(2) BP_PIV_MOV BP_PIV_RCP BP_PIV_U(3) BP_PIV_FMA BP_PIV_U(4) BP_PIV_FMAC BP_PIV_MUL
                 : "+v"(a[PN]), "+v"(a[B < ROWS ? B : 0]), "+v"(a[B + 1 < ROWS ? B + 1 : 0]), "+v"(a[B + 2 < ROWS ? B + 2 : 0]), "+v"(a[B + 3 < ROWS ? B + 3 : 0]), BP_PIV_OUT : BP_PIV_IN);
  else if constexpr (NO == 5)
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) BP_PIV_U(1) BP_PIV_U(2) BP_PIV_MOV BP_PIV_RCP BP_PIV_U(3) BP_PIV_FMA BP_PIV_U(4) BP_PIV_FMAC BP_PIV_U(5) BP_PIV_MUL
                 : "+v"(a[PN]), "+v"(a[B < ROWS ? B : 0]), "+v"(a[B + 1 < ROWS ? B + 1 : 0]), "+v"(a[B + 2 < ROWS ? B + 2 : 0]), "+v"(a[B + 3 < ROWS ? B + 3 : 0]),
                   "+v"(a[B + 4 < ROWS ? B + 4 : 0]), BP_PIV_OUT : BP_PIV_IN);
  else if constexpr (NO == 6)
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) BP_PIV_U(1) BP_PIV_U(2) BP_PIV_MOV BP_PIV_RCP BP_PIV_U(3) BP_PIV_FMA BP_PIV_U(4) BP_PIV_FMAC BP_PIV_U(5) BP_PIV_MUL BP_PIV_U(6)
                 : "+v"(a[PN]), "+v"(a[B < ROWS ? B : 0]), "+v"(a[B + 1 < ROWS ? B + 1 : 0]), "+v"(a[B + 2 < ROWS ? B + 2 : 0]), "+v"(a[B + 3 < ROWS ? B + 3 : 0]),
                   "+v"(a[B + 4 < ROWS ? B + 4 : 0]), "+v"(a[B + 5 < ROWS ? B + 5 : 0]), BP_PIV_OUT : BP_PIV_IN);
  else if constexpr (NO == 7)
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) BP_PIV_U(1) BP_PIV_U(2) BP_PIV_MOV BP_PIV_RCP BP_PIV_U(3) BP_PIV_FMA BP_PIV_U(4) BP_PIV_FMAC BP_PIV_U(5) BP_PIV_MUL BP_PIV_U(6) BP_PIV_U(7)
                 : "+v"(a[PN]), "+v"(a[B < ROWS ? B : 0]), "+v"(a[B + 1 < ROWS ? B + 1 : 0]), "+v"(a[B + 2 < ROWS ? B + 2 : 0]), "+v"(a[B + 3 < ROWS ? B + 3 : 0]),
                   "+v"(a[B + 4 < ROWS ? B + 4 : 0]), "+v"(a[B + 5 < ROWS ? B + 5 : 0]), "+v"(a[B + 6 < ROWS ? B + 6 : 0]), BP_PIV_OUT : BP_PIV_IN);
  else {
    static_assert(NO == 8, "forward_eliminate_rows: at most ten rows");
    asm volatile("s_nop 1\n\t" BP_PIV_U(0) BP_PIV_U(1) BP_PIV_U(2) BP_PIV_MOV BP_PIV_RCP BP_PIV_U(3) BP_PIV_FMA BP_PIV_U(4) BP_PIV_FMAC BP_PIV_U(5) BP_PIV_MUL BP_PIV_U(6) BP_PIV_U(7) BP_PIV_U(8)
                 : "+v"(a[PN]), "+v"(a[B < ROWS ? B : 0]), "+v"(a[B + 1 < ROWS ? B + 1 : 0]), "+v"(a[B + 2 < ROWS ? B + 2 : 0]), "+v"(a[B + 3 < ROWS ? B + 3 : 0]),
                   "+v"(a[B + 4 < ROWS ? B + 4 : 0]), "+v"(a[B + 5 < ROWS ? B + 5 : 0]), "+v"(a[B + 6 < ROWS ? B + 6 : 0]), "+v"(a[B + 7 < ROWS ? B + 7 : 0]), BP_PIV_OUT : BP_PIV_IN);
  }
}
#undef BP_PIV_U
#undef BP_PIV_MOV
#undef BP_PIV_RCP
#undef BP_PIV_FMA
#undef BP_PIV_FMAC
#undef BP_PIV_MUL
#undef BP_PIV_OUT
#undef BP_PIV_IN
template <int ROWS, int P, bool EXACT, class Emit>
__device__ __forceinline__ void forward_eliminate_rows_step(double (&v)[ROWS], int nt, bool& ok, int& minhi, double& lastp, double row, Emit& emit) {
  constexpr bool has_next = P + 1 < ROWS;
  constexpr int PN = has_next ? P + 1 : P;
  double piv = 1.0, next_row = 0.0;
  if constexpr (has_next) pivot_step_asm<ROWS, P, PN>(v, row, piv, next_row);
  emit(P, v[P], row);
  v[P] = row;
  if constexpr (has_next) {
    if constexpr (EXACT) {
      // positive pivots, one integer instruction per step: the high word of a double > 0 is a positive integer (zero, denormals and negative numbers are
      // not; a NaN may be, but it makes every later pivot NaN and the last one is compared as a double)
      if constexpr (P + 2 == ROWS) lastp = piv; else minhi = min(minhi, __double2hiint(piv));
      forward_eliminate_rows_step<ROWS, P + 1, EXACT>(v, nt, ok, minhi, lastp, next_row, emit);
    } else if (P + 1 < nt) {  // wave-uniform
      ok = ok && (piv > 0.0);
      forward_eliminate_rows_step<ROWS, P + 1, EXACT>(v, nt, ok, minhi, lastp, next_row, emit);
    }
  }
}
template <int ROWS, bool EXACT = false, class Emit>
__device__ __forceinline__ bool forward_eliminate_rows(double (&v)[ROWS], int nt, Emit&& emit) {
  if (nt <= 0) return true;
  const double piv = row_bcast<0>(v[0]);
  bool ok = piv > 0.0;
  int minhi = 0x7fffffff;
  double lastp = piv;
  forward_eliminate_rows_step<ROWS, 0, EXACT>(v, nt, ok, minhi, lastp, v[0] * fast_reciprocal(piv), emit);
  if constexpr (EXACT) ok = ok && minhi > 0 && lastp > 0.0;
  return ok;
}
template <int ROWS, int P>
__device__ __forceinline__ void back_substitute_rows_step(double (&v)[ROWS], int nt) {
  if constexpr (P >= 1) {
    if (P < nt) {  // wave-uniform
      fnmac_row_bcast<P, true>(v[0], v[P]);
#pragma unroll
      for (int i = 1; i < P; ++i) fnmac_row_bcast<P, false>(v[i], v[P]);
    }
    back_substitute_rows_step<ROWS, P - 1>(v, nt);
  }
}
template <int ROWS>
__device__ __forceinline__ void back_substitute_rows(double (&v)[ROWS], int nt) {
  back_substitute_rows_step<ROWS, ROWS - 1>(v, nt);
}

__device__ __forceinline__ double quad_swap_pairs(double v) {       // value of the lane l ^ 1
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xf, 0xf, true);    // quad_perm:[1,0,3,2]
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// du_k = K_k dx_k + kff_k for every stage in parallel, Armijo metric and step norms (dx is in HBM, workgroup-visible).
// One row of K per thread straight from global memory made every load instruction of a wave touch 64 rows x 176 B = all 88
// cache lines of its 11 KB, eleven times over, with eight waves thrashing the 16 KB L1: 23 us of a 340 us kernel.  Now a wave
// copies 32 consecutive rows (one contiguous 5.6 KB piece of K, each byte requested once, the next piece already in flight) into
// its own LDS tile and works from there with two lanes per row; the partial sums meet through a DPP quad permutation.
constexpr int kMaxRiccatiStages = 512;            // longest horizon of a solver handle (node tables of the sweeps, bpmpc_solver_create)
constexpr int kStepNormsScratch = 32 * 26;     // doubles of LDS per wave (StepNormsTile)
template <int NJ>
struct StepNormsTile {
  static constexpr int NX = 12 + NJ, NCH = NX / 2, CP = (NCH + 1) / 2;      // 16-byte chunks per row, chunks per lane
  static constexpr int ROWS = 32, LDR = 26;                                   // rows per wave and pass; row stride (13 chunks: odd, conflict free)
  static constexpr int CPI = ROWS * NCH, LPL = (CPI + kWave - 1) / kWave;     // chunks per pass, loads per lane
  static_assert(NX % 2 == 0 && 2 * CP <= LDR / 2 + 1 && NX <= LDR - 2 && ROWS * LDR == kStepNormsScratch, "tile");
};
template <int NJ, int NT = kRiccatiThreads>
__device__ __forceinline__ void riccati_step_norms(int status, const RiccatiFastIO& io, const double* hist /* LDS: dx_0 .. dx_N, row stride nx, or null */,
                                                   double* scratch /* LDS, (NT / 64) * 32 * 26 doubles */) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  (void)NXX;
  const int tid = threadIdx.x;
  const int N = io.base.N;
  double acc_arm = 0.0, acc_x = 0.0, acc_u = 0.0;
  using T = StepNormsTile<NJ>;
  constexpr int NCH = T::NCH, CP = T::CP, ROWS = T::ROWS, LDR = T::LDR, CPI = T::CPI, LPL = T::LPL, NW = NT / kWave;
  const int w = tid >> 6, l = tid & 63;
  double* tile = scratch + w * ROWS * LDR;
  for (int idx = l; idx < ROWS * LDR; idx += kWave) tile[idx] = 0.0;      // the padding columns stay zero
  const int total_rows = N * NU;
  const long total_chunks = (long)total_rows * NCH;
  const int n_pass = (total_rows + ROWS - 1) / ROWS;
  const double2* Kc = reinterpret_cast<const double2*>(io.Kfull);
  int trow[LPL], tch[LPL];
#pragma unroll
  for (int j = 0; j < LPL; ++j) { const int c = l + kWave * j; trow[j] = c / NCH; tch[j] = c % NCH; }
  // everything a pass reads from global memory is requested one pass ahead: the piece of K and, per output, kff, m, m0, nut
  double bx[LPL], by[LPL];     // (an array of double2 carried around the loop stays in scratch memory)
  double nkf = 0.0, nmv = 0.0, nms = 0.0;
  int nnut = 0;
  const int r = l >> 1, part = l & 1;
  auto fetch = [&](int pass) {
#pragma unroll
    for (int j = 0; j < LPL; ++j) {
      const long c = (long)pass * CPI + l + kWave * j;
      const bool ok = l + kWave * j < CPI && c < total_chunks;
      const double2 v = Kc[ok ? c : 0];
      bx[j] = v.x; by[j] = v.y;
    }
    const int g = pass * ROWS + r;
    const size_t gc = g < total_rows ? g : 0;
    const int k = (int)(gc / NU);
    nkf = io.kff[gc]; nmv = io.mvec[gc]; nms = io.mscal[k]; nnut = io.base.nut[k];     // mvec: k nx + i = g (nu == nx)
  };
  lds_wave_sync();
  if (w < n_pass) fetch(w);
  for (int pass = w; pass < n_pass; pass += NW) {
#pragma unroll
    for (int j = 0; j < LPL; ++j)
      if (l + kWave * j < CPI) { double2 v; v.x = bx[j]; v.y = by[j]; *reinterpret_cast<double2*>(tile + trow[j] * LDR + 2 * tch[j]) = v; }
    const double kf = nkf, mv = nmv, ms = nms;
    const int nutk = nnut;
    lds_wave_sync();
    if (pass + NW < n_pass) fetch(pass + NW);            // the next piece travels while this one is used
    const int g = pass * ROWS + r;                       // output index: stage k, input i
    const bool valid = g < total_rows;
    const int k = valid ? g / NU : 0, i = valid ? g % NU : 0;
    // dx from the history of the roll-out in LDS (from HBM when the horizon did not fit it); chunk 11 of a 22-wide row is the head
    // of the next dx, it meets the zero padding of the tile
    double2 xv[CP];
    double d;
    if (hist) {
      const double2* dxk = reinterpret_cast<const double2*>(hist + (size_t)k * NX);
#pragma unroll
      for (int jj = 0; jj < CP; ++jj) xv[jj] = dxk[2 * jj + part];
      d = hist[(size_t)k * NX + i];
    } else {
      const double2* dxk = reinterpret_cast<const double2*>(io.base.dx + (size_t)k * NX);
#pragma unroll
      for (int jj = 0; jj < CP; ++jj) xv[jj] = dxk[2 * jj + part];
      d = io.base.dx[(size_t)k * NX + i];
    }
    const double2* row = reinterpret_cast<const double2*>(tile + r * LDR);
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) {
      const double2 kv = row[2 * jj + part];
      t0 += kv.x * xv[jj].x; t1 += kv.y * xv[jj].y;
    }
    double t = t0 + t1;
    t += quad_swap_pairs(t);
    if (valid && part == 0) {
      t += kf;
      if (nutk == 0) t = 0.0;                            // event node: no input
      io.base.du[g] = t;
      acc_u += t * t;
      acc_x += d * d;                                    // NU == NX: the same index walks the state vector
      acc_arm += mv * d;
      if (i == 0) acc_arm += ms;
    }
    lds_wave_sync();                                     // the tile is rewritten by the next pass
  }
  if (tid < NX) { const double d = io.base.dx[(size_t)N * NX + tid]; acc_x += d * d; }
  __shared__ double red3[3][NT / kWave];
  for (int off = kWave / 2; off >= 1; off >>= 1) {
    acc_arm += __shfl_down(acc_arm, off);
    acc_x += __shfl_down(acc_x, off);
    acc_u += __shfl_down(acc_u, off);
  }
  if ((tid & (kWave - 1)) == 0) { red3[0][tid / kWave] = acc_arm; red3[1][tid / kWave] = acc_x; red3[2][tid / kWave] = acc_u; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, x2 = 0.0, u2 = 0.0;
    for (int w = 0; w < NT / kWave; ++w) { a += red3[0][w]; x2 += red3[1][w]; u2 += red3[2][w]; }
    io.base.summary[0] = a;
    io.base.summary[1] = x2;
    io.base.summary[2] = u2;
    io.base.summary[3] = (double)status;
  }
}

}  // namespace bpmpc
