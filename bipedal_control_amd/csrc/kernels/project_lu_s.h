// Constraint elimination, part one, STRUCTURED (HIP only; the generic version that follows Eigen::FullPivLU on the whole D operation
// for operation is project_lu4.h, still used by the reference path and mirrored by the oracle).
//
// The equality rows of a node come in two kinds:
//   zero-force rows (swing contact i, ZeroForceConstraint.cpp:58-72):   F_i = 0            -> a unit row of D in a force column, C = 0
//   velocity rows (zero velocity of a stance contact, normal velocity of a swing contact):  -> D is nonzero in the JOINT-VELOCITY columns only
// so D = [[E_swing, 0], [0, D_v]] after a row permutation, and the projection  du = Px dx + Pu du~ + Pe  of
// [OCS2-upstream] LinearAlgebra::luConstraintProjection splits into a trivial part (swing forces: du_F = -F, stance forces: free
// inputs of their own) and an LU with complete pivoting of D_v alone: at most 12 rows x NJ columns instead of 16 x (12 + NJ), one D
// column per lane instead of two, steps = rank(D_v) <= NJ.  The projector differs from FullPivLU's (another particular solution,
// another null-space basis, the reduced inputs in another order), the SQP step (dx, du, K) does not: it is independent of the basis
// (tests/test_oracle_math.py, and every GPU parity test compares exactly those).  Rank decision as FullPivLU makes it, on D_v:
// pivots above eps * min(rows, cols) * |largest pivot|.
//
// Layout of the outputs (unchanged, so the change of variables and the Riccati sweep do not care): Px (nu x nx) rows 0..11 zero,
// Pe force entries -F for swing contacts, Pu = [unit columns of the stance force components | [0 ; Z_v]] in its first
// nut = 3 n_stance_points + NJ - rank columns, zero beyond.
// Four nodes per wavefront, 16 lanes per node; lane j < NJ owns column j of D_v, every lane the columns j and 16 + j of [C_v | e_v].
#pragma once
#include <hip/hip_runtime.h>

#include "project_lu4.h"

namespace bpmpc {

// velocity rows of a node in registration order zeroForce_i, zeroVelocity_i, normalVelocity_i per contact (BipedalRobotInterface.cpp:187-191)
// by mode (FLY, LF, RF, STANCE): original row of the k-th velocity row, 255 = none
__device__ __constant__ unsigned char kVelRow[4][12] = {{3, 7, 11, 15, 255, 255, 255, 255, 255, 255, 255, 255},
                                                         {0, 1, 2, 3, 4, 5, 9, 13, 255, 255, 255, 255},
                                                         {3, 7, 8, 9, 10, 11, 12, 13, 255, 255, 255, 255},
                                                         {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}};
// zero-force row of force component c (255: the contact is in stance)
__device__ __constant__ unsigned char kForceRow[4][12] = {{0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14},
                                                           {255, 255, 255, 255, 255, 255, 6, 7, 8, 10, 11, 12},
                                                           {0, 1, 2, 4, 5, 6, 255, 255, 255, 255, 255, 255},
                                                           {255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255}};

// The same two tables packed into registers, five bits per entry (31 = none): a table in constant memory costs the node a memory round trip
// between knowing its mode and issuing its loads
__device__ __forceinline__ constexpr unsigned long long lus_pack(const unsigned char (&t)[12]) {
  unsigned long long v = 0;
  for (int i = 11; i >= 0; --i) v = (v << 5) | (unsigned long long)(t[i] == 255 ? 31 : t[i]);
  return v;
}
constexpr unsigned char kVelRowH[4][12] = {{3, 7, 11, 15, 255, 255, 255, 255, 255, 255, 255, 255},
                                           {0, 1, 2, 3, 4, 5, 9, 13, 255, 255, 255, 255},
                                           {3, 7, 8, 9, 10, 11, 12, 13, 255, 255, 255, 255},
                                           {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}};
constexpr unsigned char kForceRowH[4][12] = {{0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14},
                                             {255, 255, 255, 255, 255, 255, 6, 7, 8, 10, 11, 12},
                                             {0, 1, 2, 4, 5, 6, 255, 255, 255, 255, 255, 255},
                                             {255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255}};
__device__ __forceinline__ unsigned long long lus_vel_rows(int mode) {
  constexpr unsigned long long p0 = lus_pack(kVelRowH[0]), p1 = lus_pack(kVelRowH[1]), p2 = lus_pack(kVelRowH[2]), p3 = lus_pack(kVelRowH[3]);
  return mode == 0 ? p0 : (mode == 1 ? p1 : (mode == 2 ? p2 : p3));
}
__device__ __forceinline__ unsigned long long lus_force_rows(int mode) {
  constexpr unsigned long long p0 = lus_pack(kForceRowH[0]), p1 = lus_pack(kForceRowH[1]), p2 = lus_pack(kForceRowH[2]), p3 = lus_pack(kForceRowH[3]);
  return mode == 0 ? p0 : (mode == 1 ? p1 : (mode == 2 ? p2 : p3));
}

template <int NJ, bool PK = false>
struct ProjectLuSLds {                       // per node
  static constexpr bool kCompact = PK;
  static constexpr int UC = kCompact ? NJ : kMaxJoints;
  static constexpr int TR = kCompact ? NJ : (12 + NJ) / 2, TC = kCompact ? 16 : 12 + NJ + 2;
  union {
    alignas(16) double U[12][UC];            // upper factor by position
    double tile[TR][TC];                     // output staging: half of the rows at a time (as project_lu4.h); packed outputs: a block column at a time
  };
  double idiag[kMaxJoints];
  int colat[kMaxJoints];                     // joint column of D_v at position p
};

struct LuSLane {
  double vd[12], vr0[12], vr1[12];
  double maxpiv;
  int cpos, size, nonzero;
  bool done, alive;
};

template <int K, int RM>
__device__ __forceinline__ void lu_s_step(LuSLane& s, int j) {
  constexpr int k = K;
  double (&vd)[12] = s.vd, (&vr0)[12] = s.vr0, (&vr1)[12] = s.vr1;
  // ---- largest |a| of the trailing block
  double m0 = fabs(vd[k]);
#pragma unroll
  for (int r = k + 1; r < RM; ++r) m0 = fmax(m0, fabs(vd[r]));
  const bool stepping = s.alive && k < s.size;
  double cand = s.done ? -1.0 : m0;
  if (!stepping) cand = -1.0;
  const double pivabs = row16_allreduce_max(cand);
  const bool act = stepping && pivabs > 0.0;
  if (stepping && !act) { s.nonzero = k; s.alive = false; }
  // ---- first occurrence in column-major order
  int row0 = 31;
#pragma unroll
  for (int r = RM - 1; r >= k; --r) row0 = (fabs(vd[r]) == pivabs) ? r : row0;
  const int key0 = (!s.done && row0 < 31) ? ((s.cpos << 10) | (row0 << 5) | j) : 0x7fffffff;
  const int key = row16_allreduce_min(key0);
  const int pc = key >> 10, pl = key & 15;
  const int pr = act ? ((key >> 5) & 31) : k;
  if (act) s.maxpiv = fmax(s.maxpiv, pivabs);
  // ---- physical row swap k <-> pr
  {
    double n0 = vd[k], n2 = vr0[k], n3 = vr1[k];
#pragma unroll
    for (int r = k + 1; r < RM; ++r) {
      const bool hit = pr == r;
      double a0 = vd[r], a2 = vr0[r], a3 = vr1[r];
      asm volatile("" : "+v"(a0), "+v"(a2), "+v"(a3));      // keeps the select chain out of scratch memory (see project_lu4.h)
      n0 = hit ? a0 : n0; n2 = hit ? a2 : n2; n3 = hit ? a3 : n3;
      vd[r] = hit ? vd[k] : a0; vr0[r] = hit ? vr0[k] : a2; vr1[r] = hit ? vr1[k] : a3;
    }
    vd[k] = n0; vr0[k] = n2; vr1[k] = n3;
  }
  // ---- logical column swap: position k <-> pc
  if (act) {
    const bool me = pl == j;
    if (s.cpos == k) s.cpos = pc; else if (me) s.cpos = k;
    s.done = s.done || me;
  }
  // ---- elimination with the multipliers of the pivot column
  const double pivot = __shfl(vd[k], pl, kLuLanes);
  const double inv = act ? fast_reciprocal(pivot) : 0.0;
#pragma unroll
  for (int r = k + 1; r < RM; ++r) {
    const double colv = __shfl(vd[r], pl, kLuLanes);
    const double f = act ? colv * inv : 0.0;
    vd[r] -= f * vd[k]; vr0[r] -= f * vr0[k]; vr1[r] -= f * vr1[k];
  }
  __builtin_amdgcn_sched_barrier(0);
}

// `valid` false: the lanes run along with an empty problem and write nothing.  RM: upper bound of the velocity-row counts of the
// launch (12 double stance, 8 single support / flight).
// PK: the joint rows of the packed operand [Px | Pe | Pu] go to Vt (row stride PackedLq::WP, columns below the first unwritten block
// column 16 ceil((nx + 1 + nut) / 16)) and Px / Pu are not written at all: their force rows are zeros and single ones that every
// reader generates from the contact mode (project_struct.h, riccati_mfma.h), 2.6 KB per node instead of 7.9.
template <int NJ, int RM, bool PK = false>
__device__ __forceinline__ void project_lu_s(ProjectLuSLds<NJ, PK>& nl, bool valid, int mode, const double* D, const double* C, const double* e, double* Px,
                                             double* Pu, double* Pe, int* nut_out, int sub, int j, double* Vt = nullptr, double* prof = nullptr) {
#ifdef BPMPC_LUS_PROFILE
  long long tprev = clock64();
  int pslot = 0;
#define LUSPROF() do { const long long tn_ = clock64(); if (prof) prof[pslot] = (double)(tn_ - tprev); ++pslot; tprev = tn_; } while (0)
#else
#define LUSPROF() ((void)0)
#endif
  constexpr int NX = 12 + NJ, NU = 12 + NJ, R = 12;
  static_assert(NJ <= 16 && NX + 1 <= 32, "lane layout");
  const bool has_d = j < NJ, has_c1 = j < NX - 16, is_e = j == NX - 16;
  const int nvr = !valid ? 0 : (mode == 3 ? 12 : (mode == 0 ? 4 : 8));
  LuSLane st;
  const unsigned long long vel_rows = lus_vel_rows(mode);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int orow = (r < RM && r < nvr) ? (int)((vel_rows >> (5 * r)) & 31) : 31;
    const bool rv = orow != 31;
    const int ro = rv ? orow : 0;
    st.vd[r] = (rv && has_d) ? D[ro * NU + 12 + j] : 0.0;
    st.vr0[r] = rv ? C[ro * NX + j] : 0.0;
    st.vr1[r] = (rv && has_c1) ? C[ro * NX + 16 + j] : ((rv && is_e) ? e[ro] : 0.0);
  }
#ifdef BPMPC_LUS_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  LUSPROF();     // 0: loads
  // swing forces: du_F = -F; read before anything is written
  double pe_force = 0.0;
  if (valid && j < 12) { const int fr = (int)((lus_force_rows(mode) >> (5 * j)) & 31); if (fr != 31) pe_force = -e[fr]; }
  const int size = nvr < NJ ? nvr : NJ;              // min(rows, cols)
  st.maxpiv = 0.0; st.cpos = has_d ? j : 64; st.size = size; st.nonzero = size; st.done = !has_d; st.alive = size > 0;
  int smax = __builtin_amdgcn_readlane(size, 0);
  {
    const int s1 = __builtin_amdgcn_readlane(size, 16), s2 = __builtin_amdgcn_readlane(size, 32), s3 = __builtin_amdgcn_readlane(size, 48);
    smax = smax > s1 ? smax : s1;
    smax = smax > s2 ? smax : s2;
    smax = smax > s3 ? smax : s3;
  }
#define BP_LUS_STEP(K) if (K < RM && K < NJ && K < smax) lu_s_step<(K < RM && K < NJ) ? K : 0, RM>(st, j);
  BP_LUS_STEP(0) BP_LUS_STEP(1) BP_LUS_STEP(2) BP_LUS_STEP(3) BP_LUS_STEP(4) BP_LUS_STEP(5)
  BP_LUS_STEP(6) BP_LUS_STEP(7) BP_LUS_STEP(8) BP_LUS_STEP(9) BP_LUS_STEP(10) BP_LUS_STEP(11)
#undef BP_LUS_STEP
  LUSPROF();     // 1: elimination steps
  double (&vd)[12] = st.vd, (&vr0)[12] = st.vr0, (&vr1)[12] = st.vr1;
  const int cpos = st.cpos;
  // ---- U11 by position, column permutation, rank, reciprocal diagonal
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (cpos < NJ) nl.U[i][cpos] = vd[i];
  if (has_d) nl.colat[cpos] = j;
  lds_wave_sync();
  const double ujj = j < NJ && j < R ? nl.U[j][j] : 0.0;
  const double thr = st.maxpiv * (2.220446049250313e-16 * size);
  const unsigned long long big = __ballot(j < st.nonzero && fabs(ujj) > thr);
  const int rank = __popc((unsigned)(big >> (kLuLanes * sub)) & 0xffffu);
  if (j < NJ) nl.idiag[j] = j < rank ? 1.0 / ujj : 0.0;
  lds_wave_sync();
  LUSPROF();     // 2: U, rank
  // ---- back substitution in place: the columns of this lane are right-hand sides ([c | e], and U12 where the D column is free)
#pragma unroll
  for (int i = R - 1; i >= 0; --i) {
    if (i < NJ) {
      double t0 = vd[i], t2 = vr0[i], t3 = vr1[i];
#pragma unroll
      for (int l = i + 1; l < R; ++l) {
        if (l < NJ) {
          const double u = nl.U[i][l];
          t0 -= u * vd[l]; t2 -= u * vr0[l]; t3 -= u * vr1[l];
        }
      }
      const double id = nl.idiag[i];
      vd[i] = t0 * id; vr0[i] = t2 * id; vr1[i] = t3 * id;
      asm volatile("" : "+v"(vd[i]), "+v"(vr0[i]), "+v"(vr1[i]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  LUSPROF();     // 3: back substitution
  // ---- outputs through the LDS tile, half of the rows at a time (row = input index: 0..11 forces, 12.. joints)
  const int nsf = mode == 3 ? 12 : (mode == 0 ? 0 : 6);      // free stance-force components = the first reduced inputs
  const int nut = nsf + NJ - rank;
  const bool free_col = has_d && cpos >= rank;
  const int kc = free_col ? nsf + cpos - rank : 0;             // reduced-input column of this lane's free D column
  static_assert(NU % 2 == 0, "row passes");
  constexpr int HR = NU / 2;
  if constexpr (PK) {
    // packed joint rows: two passes of 24 columns through the tile (rows = joints)
    constexpr int WP = PackedLq<NJ>::WP, BC = NX + 1;
    constexpr bool CP = ProjectLuSLds<NJ, PK>::kCompact;
    constexpr int PW = CP ? 16 : 24, NPASS = 48 / PW;
    static_assert(NJ <= ProjectLuSLds<NJ, PK>::TR && PW <= ProjectLuSLds<NJ, PK>::TC && NPASS * PW <= WP, "the tile holds nj rows of PW columns");
    // columns below the first unwritten block column (whole 128-byte lines; zeros between the reduced inputs and the block boundary); the
    // readers mask the rest where it costs nothing: the sweeps' loaders by the address they load from, a stage ahead (PwVtLoader)
    const int cend_ = 16 * ((BC + nut + 15) >> 4);
    const int cend = cend_ < NPASS * PW ? cend_ : NPASS * PW;      // the change of variables keeps three block columns (project_mfma.h NBC_MAX)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int c0 = pass * PW;
      lds_wave_sync();
#pragma unroll
      for (int row = 0; row < NJ; ++row) { nl.tile[row][j] = 0.0; if (!CP && j < PW - 16) nl.tile[row][16 + j] = 0.0; }
      lds_wave_sync();
#pragma unroll
      for (int p = 0; p < R; ++p) {
        if (p < NJ) {
          const int row = nl.colat[p];
          if (p < rank) {
            if (j >= c0 && j < c0 + PW) nl.tile[row][j - c0] = -vr0[p];                                         // Px column j
            if ((has_c1 || is_e) && 16 + j >= c0 && 16 + j < c0 + PW) nl.tile[row][16 + j - c0] = -vr1[p];     // Px column 16 + j, Pe
          }
          if (free_col) {
            const int col = BC + kc;
            if (col >= c0 && col < c0 + PW) nl.tile[row][col - c0] = p < rank ? -vd[p] : (p == cpos ? 1.0 : 0.0);
          }
        }
      }
      lds_wave_sync();
      if (valid) {
        if (c0 + j < cend) {
#pragma unroll
          for (int row = 0; row < NJ; ++row) Vt[row * WP + c0 + j] = nl.tile[row][j];
        }
        if (!CP && j < PW - 16 && c0 + 16 + j < cend) {
#pragma unroll
          for (int row = 0; row < NJ; ++row) Vt[row * WP + c0 + 16 + j] = nl.tile[row][16 + j];
        }
        if (NX >= c0 && NX < c0 + PW && j < NJ) Pe[12 + j] = nl.tile[j][NX - c0];
      }
    }
    if (valid && j < 12) Pe[j] = pe_force;
  } else {
  // [Px | Pe]: force rows are zero (Pe: -F for swing components), joint row colat[p] = -y_p for p < rank
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r0 = pass * HR;
    lds_wave_sync();
#pragma unroll
    for (int row = 0; row < HR; ++row) { nl.tile[row][j] = 0.0; if (16 + j < NU + 2) nl.tile[row][16 + j] = 0.0; }
    lds_wave_sync();
#pragma unroll
    for (int p = 0; p < R; ++p) {
      if (p < NJ) {
        const int row = 12 + nl.colat[p] - r0;
        if (row >= 0 && row < HR && p < rank) {
          nl.tile[row][j] = -vr0[p];
          if (has_c1 || is_e) nl.tile[row][16 + j] = -vr1[p];
        }
      }
    }
    if (j < 12 && j >= r0 && j < r0 + HR) nl.tile[j - r0][NX] = pe_force;
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
  // Pu: unit columns of the free stance-force components, then [0 ; Z_v]; zero beyond nut
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r0 = pass * HR;
    lds_wave_sync();
#pragma unroll
    for (int row = 0; row < HR; ++row) { nl.tile[row][j] = 0.0; if (16 + j < NU + 2) nl.tile[row][16 + j] = 0.0; }
    lds_wave_sync();
    if (j < 12) {                                            // force component j: free iff its contact is in stance
      const bool stance = kForceRow[mode][j] == 255;
      const int col = mode == 3 ? j : (mode == 1 ? j : j - 6);   // LF: components 0..5 are the stance ones, RF: 6..11
      if (valid && stance && j >= r0 && j < r0 + HR) nl.tile[j - r0][col] = 1.0;
    }
    if (free_col) {
#pragma unroll
      for (int p = 0; p < R; ++p) {
        if (p < NJ) {
          const int row = 12 + nl.colat[p] - r0;
          if (row >= 0 && row < HR) nl.tile[row][kc] = p < rank ? -vd[p] : (p == cpos ? 1.0 : 0.0);
        }
      }
    }
    lds_wave_sync();
    if (valid) {
#pragma unroll
      for (int row = 0; row < HR; ++row) Pu[(r0 + row) * NU + j] = nl.tile[row][j];
      if (j < NU - 16) {
#pragma unroll
        for (int row = 0; row < HR; ++row) Pu[(r0 + row) * NU + 16 + j] = nl.tile[row][16 + j];
      }
    }
  }
  }
#ifdef BPMPC_LUS_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  LUSPROF();     // 4: outputs
  if (valid && j == 0) nut_out[0] = nut;
}

}  // namespace bpmpc
