// Constraint elimination, part one (HIP only): LU with complete pivoting of D and the solves that give Px, Pu, Pe.
// Reference version with identical semantics: project_node.h (Eigen::FullPivLU semantics, see its header).
//
// Four nodes per wavefront, 16 lanes per node.  Lane j owns the columns j and 16 + j of D (the latter for j < NU - 16)
// and the columns j and 16 + j of the right-hand sides [C | e] in registers, all 16 rows of each.  Every lane applies the
// same row operations, so the unit-lower solve of the right-hand sides is folded into the elimination.
//   * rows are swapped physically (register row index == current position), the step index k is a compile-time constant
//     (fully unrolled), so the trailing block shrinks statically and the first-in-column-major tie break of
//     Eigen::FullPivLU::compute falls out of a forward scan; columns are permuted logically (position per slot);
//   * reductions over a node are 16-lane DPP rotations (maximum of |a|, then minimum of the packed (column position,
//     row, slot, lane) key among the entries that hold the maximum);
//   * the pivot column is broadcast with ds_bpermute (no LDS storage), pivoted columns are simply never searched again:
//     their stale entries below the diagonal are harmless;
//   * the back substitution U11 y = [c | U12] runs in place on the same registers; only U11 (16 x 16 per node), the
//     reciprocal diagonal and the column permutation go through LDS;
//   * Px, Pu, Pe are written once, every element by exactly one lane.
// A node whose k-th step finds a zero block stops there (nonzero pivots = k), as FullPivLU does.
#pragma once
#include <hip/hip_runtime.h>

#include "project_node.h"
#include "riccati_fast.h"   // lds_wave_sync

namespace bpmpc {

constexpr int kLuLanes = 16;                 // lanes per node
constexpr int kLuNodes = kWave / kLuLanes;   // nodes per wavefront

template <int NJ>
struct ProjectLuLds {                        // per node
  union {
    alignas(16) double U[kMaxEqRows][kMaxEqRows];   // upper factor, rows and columns by position
    double tile[(12 + NJ) / 2][12 + NJ + 2];   // output staging, half of the rows of [Px | Pe] resp. Pu at a time (9.5 KB of LDS per wave at H1)
  };
  double idiag[kMaxEqRows];
  int colat[32];                             // physical column of D at position p
};

__device__ __forceinline__ double row16_allreduce_max(double x) {
#define BP_ROR_MAX(n)                                                                                           \
  {                                                                                                             \
    const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x120 + n, 0xf, 0xf, false);              \
    const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x120 + n, 0xf, 0xf, false);              \
    x = fmax(x, __hiloint2double(hi_, lo_));                                                                    \
  }
  BP_ROR_MAX(1) BP_ROR_MAX(2) BP_ROR_MAX(4) BP_ROR_MAX(8)
#undef BP_ROR_MAX
  return x;
}
__device__ __forceinline__ int row16_allreduce_min(int x) {
#define BP_ROR_MIN(n)                                                                   \
  {                                                                                     \
    const int o_ = __builtin_amdgcn_update_dpp(0, x, 0x120 + n, 0xf, 0xf, false);       \
    x = o_ < x ? o_ : x;                                                                \
  }
  BP_ROR_MIN(1) BP_ROR_MIN(2) BP_ROR_MIN(4) BP_ROR_MIN(8)
#undef BP_ROR_MIN
  return x;
}

// One elimination step with compile-time position K (see project_lu4 below).
struct LuLane {
  double vd0[kMaxEqRows], vd1[kMaxEqRows], vr0[kMaxEqRows], vr1[kMaxEqRows];
  double maxpiv;
  int cpos0, cpos1, size, nonzero;
  bool done0, done1, alive;
};
// RM: rows RM.. are zero for every node of the launch (RM >= the largest row count), so the row loops stop there.
template <int K, int RM>
__device__ __forceinline__ void lu_step(LuLane& s, int j) {
  constexpr int R = kMaxEqRows, k = K;
  double (&vd0)[R] = s.vd0, (&vd1)[R] = s.vd1, (&vr0)[R] = s.vr0, (&vr1)[R] = s.vr1;
  double& maxpiv = s.maxpiv;
  int &cpos0 = s.cpos0, &cpos1 = s.cpos1, &nonzero = s.nonzero;
  const int size = s.size;
  bool &done0 = s.done0, &done1 = s.done1, &alive = s.alive;
    // ---- largest |a| of the trailing block
    double m0 = fabs(vd0[k]), m1 = fabs(vd1[k]);
#pragma unroll
    for (int r = k + 1; r < RM; ++r) { m0 = fmax(m0, fabs(vd0[r])); m1 = fmax(m1, fabs(vd1[r])); }
    const bool stepping = alive && k < size;
    double cand = fmax(done0 ? -1.0 : m0, done1 ? -1.0 : m1);
    if (!stepping) cand = -1.0;
    const double pivabs = row16_allreduce_max(cand);
    const bool act = stepping && pivabs > 0.0;
    if (stepping && !act) { nonzero = k; alive = false; }
    // ---- first occurrence in column-major order: smallest column position, then smallest row
    int row0 = 31, row1 = 31;
#pragma unroll
    for (int r = RM - 1; r >= k; --r) {
      row0 = (fabs(vd0[r]) == pivabs) ? r : row0;
      row1 = (fabs(vd1[r]) == pivabs) ? r : row1;
    }
    const int key0 = (!done0 && row0 < 31) ? ((cpos0 << 10) | (row0 << 5) | j) : 0x7fffffff;
    const int key1 = (!done1 && row1 < 31) ? ((cpos1 << 10) | (row1 << 5) | 16 | j) : 0x7fffffff;
    const int key = row16_allreduce_min(key0 < key1 ? key0 : key1);
    const int pc = key >> 10, ps = (key >> 4) & 1, pl = key & 15;
    const int pr = act ? ((key >> 5) & 31) : k;
    if (act) maxpiv = fmax(maxpiv, pivabs);
    // ---- physical row swap k <-> pr
    {
      double n0 = vd0[k], n1 = vd1[k], n2 = vr0[k], n3 = vr1[k];
#pragma unroll
      for (int r = k + 1; r < RM; ++r) {
        const bool hit = pr == r;
        // opaque copies: without them the compiler folds the select chain into a dynamically indexed access, which
        // forces the column arrays into scratch memory
        double a0 = vd0[r], a1 = vd1[r], a2 = vr0[r], a3 = vr1[r];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        n0 = hit ? a0 : n0; n1 = hit ? a1 : n1; n2 = hit ? a2 : n2; n3 = hit ? a3 : n3;
        vd0[r] = hit ? vd0[k] : a0; vd1[r] = hit ? vd1[k] : a1; vr0[r] = hit ? vr0[k] : a2; vr1[r] = hit ? vr1[k] : a3;
      }
      vd0[k] = n0; vd1[k] = n1; vr0[k] = n2; vr1[k] = n3;
    }
    // ---- logical column swap: position k <-> pc; the pivot column is never searched again
    if (act) {
      const bool me0 = ps == 0 && pl == j, me1 = ps == 1 && pl == j;
      if (cpos0 == k) cpos0 = pc; else if (me0) cpos0 = k;
      if (cpos1 == k) cpos1 = pc; else if (me1) cpos1 = k;
      done0 = done0 || me0;
      done1 = done1 || me1;
    }
    // ---- elimination with the multipliers of the pivot column (broadcast from its lane)
    const double pivot = __shfl(ps ? vd1[k] : vd0[k], pl, kLuLanes);
    const double inv = act ? fast_reciprocal(pivot) : 0.0;
#pragma unroll
    for (int r = k + 1; r < RM; ++r) {
      const double colv = __shfl(ps ? vd1[r] : vd0[r], pl, kLuLanes);
      const double f = act ? colv * inv : 0.0;
      vd0[r] -= f * vd0[k]; vd1[r] -= f * vd1[k]; vr0[r] -= f * vr0[k]; vr1[r] -= f * vr1[k];
    }
  __builtin_amdgcn_sched_barrier(0);           // steps are strictly sequential: interleaving them only costs registers
}

// `valid` false: the lanes run along with an empty problem (nc = 0) and write nothing.
// RM: upper bound of the row counts of the launch (12 double stance, 14 single support, 16 flight), known on the host.
template <int NJ, int RM>
__device__ __forceinline__ void project_lu4(ProjectLuLds<NJ>& nl, bool valid, int nc, const double* D, const double* C, const double* e, double* Px,
                                            double* Pu, double* Pe, int* nut_out, int sub, int j) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ, R = kMaxEqRows;
  static_assert(NU > 16 && NU <= 32 && NX + 1 <= 32 && R == 16, "lane layout");
  const bool has_d1 = j < NU - 16, has_c1 = j < NX - 16, is_e = j == NX - 16;
  double vd0[R], vd1[R], vr0[R], vr1[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {                // rows >= nc do not exist (the fused solve mode does not even write their zero padding)
    const bool rv = valid && r < nc;
    vd0[r] = rv ? D[r * NU + j] : 0.0;
    vd1[r] = (rv && has_d1) ? D[r * NU + 16 + j] : 0.0;
    vr0[r] = rv ? C[r * NX + j] : 0.0;
    vr1[r] = (rv && has_c1) ? C[r * NX + 16 + j] : ((rv && is_e) ? e[r] : 0.0);
  }
  const int size = valid ? nc : 0;             // min(rows, cols), rows <= 16 < cols
  int nonzero = size;
  bool alive = size > 0;
  double maxpiv = 0.0;
  int cpos0 = j, cpos1 = has_d1 ? 16 + j : 64;
  bool done0 = false, done1 = !has_d1;
  int smax = __builtin_amdgcn_readlane(size, 0);
  {
    const int s1 = __builtin_amdgcn_readlane(size, 16), s2 = __builtin_amdgcn_readlane(size, 32), s3 = __builtin_amdgcn_readlane(size, 48);
    smax = smax > s1 ? smax : s1;
    smax = smax > s2 ? smax : s2;
    smax = smax > s3 ? smax : s3;
  }
  LuLane st;
#pragma unroll
  for (int r = 0; r < R; ++r) { st.vd0[r] = vd0[r]; st.vd1[r] = vd1[r]; st.vr0[r] = vr0[r]; st.vr1[r] = vr1[r]; }
  st.maxpiv = 0.0; st.cpos0 = cpos0; st.cpos1 = cpos1; st.size = size; st.nonzero = nonzero; st.done0 = done0; st.done1 = done1; st.alive = alive;
#define BP_LU_STEP(K) if (K < RM && K < smax) lu_step<K < RM ? K : 0, RM>(st, j);
  BP_LU_STEP(0) BP_LU_STEP(1) BP_LU_STEP(2) BP_LU_STEP(3) BP_LU_STEP(4) BP_LU_STEP(5) BP_LU_STEP(6) BP_LU_STEP(7)
  BP_LU_STEP(8) BP_LU_STEP(9) BP_LU_STEP(10) BP_LU_STEP(11) BP_LU_STEP(12) BP_LU_STEP(13) BP_LU_STEP(14) BP_LU_STEP(15)
#undef BP_LU_STEP
#pragma unroll
  for (int r = 0; r < R; ++r) { vd0[r] = st.vd0[r]; vd1[r] = st.vd1[r]; vr0[r] = st.vr0[r]; vr1[r] = st.vr1[r]; }
  maxpiv = st.maxpiv; cpos0 = st.cpos0; cpos1 = st.cpos1; nonzero = st.nonzero;
  // ---- U11 by position, column permutation, rank (threshold of Eigen::FullPivLU::rank), reciprocal diagonal
#pragma unroll
  for (int i = 0; i < R; ++i) {
    if (cpos0 < R) nl.U[i][cpos0] = vd0[i];
    if (cpos1 < R) nl.U[i][cpos1] = vd1[i];
  }
  nl.colat[cpos0] = j;
  if (has_d1) nl.colat[cpos1] = 16 + j;
  lds_wave_sync();
  const double ujj = nl.U[j][j];
  const double thr = maxpiv * (2.220446049250313e-16 * size);
  const unsigned long long big = __ballot(j < nonzero && fabs(ujj) > thr);
  const int rank = __popc((unsigned)(big >> (kLuLanes * sub)) & 0xffffu);
  nl.idiag[j] = j < rank ? 1.0 / ujj : 0.0;
  lds_wave_sync();
  // ---- back substitution in place: the four columns of this lane are right-hand sides ([c | e], and U12 where the
  //      D column is a free one); rows >= rank come out as zero through the zero reciprocal diagonal
#pragma unroll
  for (int i = R - 1; i >= 0; --i) {
    double t0 = vd0[i], t1 = vd1[i], t2 = vr0[i], t3 = vr1[i];
#pragma unroll
    for (int l = i + 1; l < R; ++l) {
      const double u = nl.U[i][l];
      t0 -= u * vd0[l]; t1 -= u * vd1[l]; t2 -= u * vr0[l]; t3 -= u * vr1[l];
    }
    const double id = nl.idiag[i];
    vd0[i] = t0 * id; vd1[i] = t1 * id; vr0[i] = t2 * id; vr1[i] = t3 * id;
    // pin the results here: otherwise the solves of the second-slot columns are sunk into the predicated output blocks
    // below, which keeps every U entry alive in registers until then
    asm volatile("" : "+v"(vd0[i]), "+v"(vd1[i]), "+v"(vr0[i]), "+v"(vr1[i]));
    __builtin_amdgcn_sched_barrier(0);         // keep the LDS reads of later rows from being hoisted (register pressure)
  }
  // ---- Px = -Q [y; 0], Pe likewise, Pu = Q [-U11^-1 U12; I]; the pivot columns of D fill the zero columns nut.. of Pu
  const int nut = NU - rank;
  const bool free0 = cpos0 >= rank, free1 = cpos1 >= rank;
  const int kc0 = free0 ? cpos0 - rank : nut + cpos0;
  const int kc1 = free1 ? cpos1 - rank : nut + cpos1;
  // The results are permuted through an LDS tile (it overlays U, which is dead now) and leave in row order with
  // immediate-offset stores; writing them straight from the solve needs a computed 64-bit address per element, which
  // tripled the register count of the kernel.  The tile holds half of the rows, so [Px | Pe] and Pu each go out in two passes
  // of whole rows (splitting by columns instead wrote the cache sectors under the split twice: +38 % HBM write traffic).
  static_assert(NU % 2 == 0, "row passes");
  constexpr int HR = NU / 2;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r0 = pass * HR;
    lds_wave_sync();
#pragma unroll
    for (int p = 0; p < NU; ++p) {
      const int row = nl.colat[p] - r0;
      if (row >= 0 && row < HR) nl.tile[row][j] = (p < R && p < rank) ? -vr0[p < R ? p : 0] : 0.0;
    }
    if (has_c1 || is_e) {                      // columns 16.. of Px and Pe (kept in column NX of the tile)
#pragma unroll
      for (int p = 0; p < NU; ++p) {
        const int row = nl.colat[p] - r0;
        if (row >= 0 && row < HR) nl.tile[row][16 + j] = (p < R && p < rank) ? -vr1[p < R ? p : 0] : 0.0;
      }
    }
    lds_wave_sync();
    if (valid) {
#pragma unroll
      for (int row = 0; row < HR; ++row) Px[(r0 + row) * NX + j] = nl.tile[row][j];
      if (has_c1) {
#pragma unroll
        for (int row = 0; row < HR; ++row) Px[(r0 + row) * NX + 16 + j] = nl.tile[row][16 + j];
      }
      if (j < HR) Pe[r0 + j] = nl.tile[j][NX];
    }
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r0 = pass * HR;
    lds_wave_sync();
#pragma unroll
    for (int p = 0; p < NU; ++p) {
      const int row = nl.colat[p] - r0;
      const double y = p < R ? vd0[p < R ? p : 0] : 0.0;
      if (row >= 0 && row < HR) nl.tile[row][kc0] = free0 ? (p < rank ? -y : (p == cpos0 ? 1.0 : 0.0)) : 0.0;
    }
    if (has_d1) {
#pragma unroll
      for (int p = 0; p < NU; ++p) {
        const int row = nl.colat[p] - r0;
        const double y = p < R ? vd1[p < R ? p : 0] : 0.0;
        if (row >= 0 && row < HR) nl.tile[row][kc1] = free1 ? (p < rank ? -y : (p == cpos1 ? 1.0 : 0.0)) : 0.0;
      }
    }
    lds_wave_sync();
    if (valid) {
#pragma unroll
      for (int row = 0; row < HR; ++row) Pu[(r0 + row) * NU + j] = nl.tile[row][j];
      if (has_d1) {
#pragma unroll
        for (int row = 0; row < HR; ++row) Pu[(r0 + row) * NU + 16 + j] = nl.tile[row][16 + j];
      }
    }
  }
  if (valid && j == 0) nut_out[0] = nut;
}

}  // namespace bpmpc
