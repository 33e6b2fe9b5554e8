// Fast elimination of the state-input equality constraints (HIP only; reference version with identical semantics:
// project_node.h).  One wavefront per node.
//
//   * every HBM read of the node (D, C, e, A, Q, R, dense rows of B, b, q, r) is issued up front into registers, so the
//     wave pays one memory latency, not one per phase;
//   * LU with complete pivoting held in registers: lane c < NU owns column c of D, lanes NU .. 2NU own the columns of
//     [C | e]; all lanes apply the same row operations, so the unit-lower solve of the right-hand sides is free.
//     Rows and columns are permuted logically (position arrays); the pivot order, including the first-in-column-major
//     tie break of Eigen::FullPivLU::compute, is reproduced exactly from the tracked positions.  The wave maximum is a
//     DPP reduction; the pivot column is broadcast with v_readlane;
//   * the ordered factors go to LDS once; the triangular solves (U11 y = [c | U12]) run one lane per right-hand side with
//     fully unrolled loops;
//   * the change of variables uses 2x2 register tiles on 16-byte LDS reads, and the structure of the discretised
//     centroidal dynamics: only rows 3..11 of B are dense (rows 0..2 are (dt/m) on the three force components, rows 12..
//     are dt on the joint velocity), so B Px, B Pu, B Pe cost 9 dense rows instead of NX.
// The cost cross term P of the LQ model is structurally zero for this problem (tracking cost and soft cones have no
// state-input coupling) and is not read here; project_node.h handles a general P.
#pragma once
#include <hip/hip_runtime.h>

#include "project_node.h"
#include "riccati_fast.h"   // lds_wave_sync, readlane_f64, lds_pair

namespace bpmpc {

template <int NJ>
struct ProjectFastWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ, LD = NX + 2, NRHS = NX + 1 + NU;
  struct Factors {
    alignas(16) double U[kMaxEqRows][NU + 2];      // ordered upper factor (position space)
    alignas(16) double Y[kMaxEqRows][NRHS + 1];    // ordered right-hand sides [c | U12], overwritten by the solutions
  };
  union {                                          // the factors are dead once Px, Pu, Pe exist; R Pu reuses their space
    Factors f;
    alignas(16) double RPu[NU][LD];
  };
  alignas(16) double Px[NU][LD], Pu[NU][LD], R[NU][LD], RPx[NU][LD];
  alignas(16) double Bd[9][LD];                    // dense rows 3..11 of B
  alignas(16) double Pe[NU], rr[NU], idiag[kMaxEqRows];
  double bscale[NX];                               // the single entry of the sparse rows of B (rows 0..2, 12..)
  int lane_at_pos[NU];
};

__device__ __forceinline__ double select16(const double (&v)[kMaxEqRows], int r) {
  // The asm barrier keeps every element an opaque register value: without it the compiler folds the select chain into
  // a dynamically indexed load, which forces the whole column array into scratch memory.
  double out = v[0];
#pragma unroll
  for (int i = 1; i < kMaxEqRows; ++i) {
    double vi = v[i];
    asm volatile("" : "+v"(vi));
    out = (r == i) ? vi : out;
  }
  return out;
}

// maximum over the wavefront (DPP reduction ladder, result broadcast from lane 63)
__device__ __forceinline__ double wave_max_f64(double x) {
#define BP_DPP_MAX(ctrl, rmask)                                                                              \
  {                                                                                                          \
    const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), ctrl, rmask, 0xf, false); \
    const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), ctrl, rmask, 0xf, false); \
    x = fmax(x, __hiloint2double(hi_, lo_));                                                                 \
  }
  BP_DPP_MAX(0x111, 0xf)  // row_shr:1
  BP_DPP_MAX(0x112, 0xf)  // row_shr:2
  BP_DPP_MAX(0x114, 0xf)  // row_shr:4
  BP_DPP_MAX(0x118, 0xf)  // row_shr:8   -> lane 15 of every row holds the row maximum
  BP_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  BP_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
#undef BP_DPP_MAX
  return readlane_f64(x, 63);
}

template <int NJ>
__device__ __forceinline__ void project_fast(ProjectFastWorkspace<NJ>& ws, const ProjectIn& in, const ProjectOut& out, double* prof = nullptr) {
  using WS = ProjectFastWorkspace<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, LD = WS::LD, H2 = NX / 2;
  constexpr int IT_R = (NU * NU + kWave - 1) / kWave, IT_B = (9 * NU + kWave - 1) / kWave, IT_A = (NX * H2 + kWave - 1) / kWave,
                IT_Q = (H2 * H2 + kWave - 1) / kWave;
  static_assert(2 * NU + 1 <= kWave && NX == NU && (NX % 2) == 0, "lane layout");
  const int lane = threadIdx.x;
  (void)prof;

  if (in.kind == 1) {  // event node: pass-through (same as the reference kernel)
    for (int idx = lane; idx < NX * NX; idx += kWave) { out.At[idx] = in.A[idx]; out.Qt[idx] = in.Q[idx]; }
    for (int idx = lane; idx < NX * NU; idx += kWave) { out.Bt[idx] = 0.0; out.Pt[idx] = 0.0; out.Px[idx] = 0.0; }
    for (int idx = lane; idx < NU * NU; idx += kWave) { out.Rt[idx] = 0.0; out.Pu[idx] = 0.0; }
    if (lane < NX) { out.bt[lane] = in.b[lane]; out.qt[lane] = in.q[lane]; }
    if (lane < NU) { out.rt[lane] = 0.0; out.Pe[lane] = 0.0; }
    if (lane == 0) out.nut[0] = 0;
    return;
  }
#ifdef BPMPC_PROJECT_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define PPROF(slot) do { const long long tn_ = clock64(); tacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define PPROF(slot) ((void)0)
#endif
#ifdef BPMPC_PROJECT_PROFILE_LU
  long long lacc[8] = {0,0,0,0,0,0,0,0};
#define LPROF(slot) do { const long long tn_ = clock64(); lacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define LPROF(slot) ((void)0)
#endif

  const int rows = in.nc;                     // <= 16, wave uniform
  const bool is_d = lane < NU;                // column of D
  const bool is_rhs = lane >= NU && lane < NU + NX + 1;   // column of [C | e]
  // ---- all HBM reads of this node, back to back
  double v[kMaxEqRows];
  {
    const double* src = is_d ? in.D + lane : (lane < NU + NX ? in.C + (lane - NU) : in.e);
    const int stride = is_d ? NU : (lane < NU + NX ? NX : 1);
    const bool valid = lane < NU + NX + 1;
#pragma unroll
    for (int r = 0; r < kMaxEqRows; ++r) v[r] = valid ? src[r * stride] : 0.0;   // rows >= nc are zero in HBM
  }
  double pR[IT_R], pB[IT_B], pA[IT_A][2], pQ[IT_Q][4], pvec[3];
#pragma unroll
  for (int it = 0; it < IT_R; ++it) { const int idx = lane + it * kWave; pR[it] = idx < NU * NU ? in.R[idx] : 0.0; }
#pragma unroll
  for (int it = 0; it < IT_B; ++it) { const int idx = lane + it * kWave; pB[it] = idx < 9 * NU ? in.B[3 * NU + idx] : 0.0; }
#pragma unroll
  for (int it = 0; it < IT_A; ++it) {
    const int idx = lane + it * kWave;
    const bool ok = idx < NX * H2;
    const int i = idx / H2, tj = idx % H2;
    pA[it][0] = ok ? in.A[i * NX + 2 * tj] : 0.0;
    pA[it][1] = ok ? in.A[i * NX + 2 * tj + 1] : 0.0;
  }
#pragma unroll
  for (int it = 0; it < IT_Q; ++it) {
    const int w = lane + it * kWave;
    const bool ok = w < H2 * H2;
    const double* Qi = in.Q + (2 * (w / H2)) * NX + 2 * (w % H2);
    pQ[it][0] = ok ? Qi[0] : 0.0; pQ[it][1] = ok ? Qi[1] : 0.0; pQ[it][2] = ok ? Qi[NX] : 0.0; pQ[it][3] = ok ? Qi[NX + 1] : 0.0;
  }
  pvec[0] = lane < NX ? in.b[lane] : 0.0;
  pvec[1] = lane < NX ? in.q[lane] : 0.0;
  pvec[2] = lane < NU ? in.r[lane] : 0.0;
  const double bsc = lane < NX ? ((lane < 3 || lane >= 12) ? in.B[lane * NU + lane] : 0.0) : 0.0;
  for (int idx = lane; idx < NU * LD; idx += kWave) { (&ws.Px[0][0])[idx] = 0.0; (&ws.Pu[0][0])[idx] = 0.0; }
  if (lane < NU) ws.Pe[lane] = 0.0;
  PPROF(0);

  // ---- LU with complete pivoting.  rowpos[r]: current position of physical row r; colpos: position of this lane's column.
  int rowpos[kMaxEqRows];
#pragma unroll
  for (int r = 0; r < kMaxEqRows; ++r) rowpos[r] = r;
  int colpos = lane;                          // meaningful for D lanes
  const int size = rows < NU ? rows : NU;
  int nonzero = size;
  double maxpivot = 0.0;
#pragma nounroll
  for (int k = 0; k < size; ++k) {
    // candidate of this lane: first maximum over the not yet pivoted rows (positions >= k) of its column
    double best = -1.0;
    int brow = 0, bpos = 0x7fff;
#pragma unroll
    for (int r = 0; r < kMaxEqRows; ++r) {
      const int rp = rowpos[r];
      const double a = (r < rows && rp >= k) ? fabs(v[r]) : -2.0;   // pivoted rows never win
      const bool take = (a > best) || (a == best && rp < bpos);
      best = take ? a : best;
      brow = take ? r : brow;
      bpos = take ? rp : bpos;
    }
    if (!is_d || colpos < k) best = -1.0;
    LPROF(0);
    const double pivabs = wave_max_f64(best);
    LPROF(1);
    if (pivabs == 0.0) { nonzero = k; break; }
    // winner: among the lanes that hold the maximum, smallest (column position, row position)
    unsigned long long tied = __ballot(best == pivabs);
    int plane = __ffsll((long long)tied) - 1;
    if (__popcll(tied) > 1) {
      int wcol = 0x7fff, wpos = 0x7fff;
      while (tied) {
        const int l = __ffsll((long long)tied) - 1;
        tied &= tied - 1;
        const int c = __builtin_amdgcn_readlane(colpos, l), p = __builtin_amdgcn_readlane(bpos, l);
        if (c < wcol || (c == wcol && p < wpos)) { wcol = c; wpos = p; plane = l; }
      }
    }
    const int pr = __builtin_amdgcn_readlane(brow, plane), ppos = __builtin_amdgcn_readlane(bpos, plane);
    const int pcpos = __builtin_amdgcn_readlane(colpos, plane);
    if (pivabs > maxpivot) maxpivot = pivabs;
    LPROF(2);
    // logical swaps: row at position k <-> pivot row, column at position k <-> pivot column
#pragma unroll
    for (int r = 0; r < kMaxEqRows; ++r) { if (rowpos[r] == k) rowpos[r] = ppos; else if (r == pr) rowpos[r] = k; }
    if (is_d) { if (colpos == k) colpos = pcpos; else if (lane == plane) colpos = k; }
    // eliminate: factors = pivot column (broadcast from its lane), pivot row element of this lane
    LPROF(3);
    const double vp = select16(v, pr);
    const double piv = readlane_f64(vp, plane);
    const double scale = vp * (1.0 / piv);
    const bool update = (is_d && colpos > k) || is_rhs;
#pragma unroll
    for (int r = 0; r < kMaxEqRows; ++r) {
      if (r < rows && rowpos[r] > k) {        // uniform
        const double f = readlane_f64(v[r], plane);
        if (update) v[r] -= f * scale;
      }
    }
    LPROF(4);
  }
  PPROF(1);
  // ---- ordered factors to LDS, staged inputs to LDS, rank (threshold of Eigen::FullPivLU::rank)
  if (is_d) ws.lane_at_pos[colpos] = lane;
#pragma unroll
  for (int r = 0; r < kMaxEqRows; ++r) {
    if (r < rows) {
      const int p = rowpos[r];
      if (is_d) ws.f.U[p][colpos] = v[r];
      else if (is_rhs) ws.f.Y[p][lane - NU] = v[r];
    }
  }
#pragma unroll
  for (int it = 0; it < IT_R; ++it) { const int idx = lane + it * kWave; if (idx < NU * NU) ws.R[idx / NU][idx % NU] = pR[it]; }
#pragma unroll
  for (int it = 0; it < IT_B; ++it) { const int idx = lane + it * kWave; if (idx < 9 * NU) ws.Bd[idx / NU][idx % NU] = pB[it]; }
  if (lane < NX) ws.bscale[lane] = bsc;
  lds_wave_sync();
  const double thr = fabs(maxpivot) * (2.220446049250313e-16 * size);
  int rank = 0;
  for (int i = 0; i < nonzero; ++i) rank += (fabs(ws.f.U[i][i]) > thr) ? 1 : 0;
  const int nut = NU - rank;
  // kernel right-hand sides U12 and reciprocal diagonal
  for (int idx = lane; idx < rank * nut; idx += kWave) ws.f.Y[idx / nut][NX + 1 + idx % nut] = ws.f.U[idx / nut][rank + idx % nut];
  if (lane < rank) ws.idiag[lane] = 1.0 / ws.f.U[lane][lane];
  lds_wave_sync();
  PPROF(2);
  // ---- back substitution, one lane per right-hand side
  if (lane < NX + 1 + nut) {
    double y[kMaxEqRows];
#pragma unroll
    for (int i = kMaxEqRows - 1; i >= 0; --i) {
      if (i < rank) {                         // uniform
        double t = ws.f.Y[i][lane];
#pragma unroll
        for (int l = i + 1; l < kMaxEqRows; ++l)
          if (l < rank) t -= ws.f.U[i][l] * y[l];
        y[i] = t * ws.idiag[i];
      }
    }
    // scatter through the column permutation: Px = -Q [y; 0], Pe likewise, Pu = Q [-U11^-1 U12; I]
#pragma unroll
    for (int i = 0; i < kMaxEqRows; ++i) {
      if (i < rank) {
        const int row = ws.lane_at_pos[i];
        if (lane < NX) ws.Px[row][lane] = -y[i];
        else if (lane == NX) ws.Pe[row] = -y[i];
        else ws.Pu[row][lane - NX - 1] = -y[i];
      }
    }
    if (lane > NX) ws.Pu[ws.lane_at_pos[rank + (lane - NX - 1)]][lane - NX - 1] = 1.0;
  }
  lds_wave_sync();
  PPROF(3);
  if (lane == 0) out.nut[0] = nut;
  const int nt2 = (nut + 1) / 2;
  // ---- products.  P1: RPx = R Px (H2 x H2 tiles), RPu = R Pu (H2 x nt2 tiles), rr = r + R Pe; projection to HBM
  for (int w = lane; w < H2 * H2 + H2 * nt2; w += kWave) {
    const bool second = w >= H2 * H2;
    const int t = second ? w - H2 * H2 : w;
    const int ti = second ? t / nt2 : t / H2, tj = second ? t % nt2 : t % H2;
    const double* X = second ? &ws.Pu[0][2 * tj] : &ws.Px[0][2 * tj];
    const double* Rl = &ws.R[0][2 * ti];      // R symmetric: row pair read as column pair
    double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
#pragma unroll
    for (int l = 0; l < NU; ++l) {
      const d2 rv = lds_pair(Rl + l * LD), xv = lds_pair(X + l * LD);
      c00 += rv.x * xv.x; c01 += rv.x * xv.y; c10 += rv.y * xv.x; c11 += rv.y * xv.y;
    }
    double* O = second ? &ws.RPu[2 * ti][2 * tj] : &ws.RPx[2 * ti][2 * tj];
    O[0] = c00; O[1] = c01; O[LD] = c10; O[LD + 1] = c11;
  }
  if (lane < NU) {
    double t = pvec[2];
#pragma unroll
    for (int l = 0; l < NU; ++l) t += ws.R[l][lane] * ws.Pe[l];
    ws.rr[lane] = t;
    out.Pe[lane] = ws.Pe[lane];
  }
  for (int idx = lane; idx < NU * NX; idx += kWave) { out.Px[idx] = ws.Px[idx / NX][idx % NX]; out.Pu[idx] = ws.Pu[idx / NU][idx % NU]; }
  lds_wave_sync();
  PPROF(4);
  // ---- P2: everything that goes to HBM.  At = A + B Px, Bt = B Pu (column pairs per lane; sparse rows of B are one entry)
#pragma unroll
  for (int it = 0; it < IT_A; ++it) {
    const int idx = lane + it * kWave;
    if (idx < NX * H2) {
      const int i = idx / H2, tj = idx % H2;
      double a0 = pA[it][0], a1 = pA[it][1], b0 = 0.0, b1 = 0.0;
      if (i >= 3 && i < 12) {
        const double* Bl = ws.Bd[i - 3];
#pragma unroll
        for (int l = 0; l < NU; ++l) {
          const double bv = Bl[l];
          const d2 px = lds_pair(&ws.Px[l][2 * tj]), pu = lds_pair(&ws.Pu[l][2 * tj]);
          a0 += bv * px.x; a1 += bv * px.y; b0 += bv * pu.x; b1 += bv * pu.y;
        }
      } else {
        const double sc = ws.bscale[i];
        if (i < 3) {
          for (int c = 0; c < kNumContacts; ++c) {
            const d2 px = lds_pair(&ws.Px[3 * c + i][2 * tj]), pu = lds_pair(&ws.Pu[3 * c + i][2 * tj]);
            a0 += sc * px.x; a1 += sc * px.y; b0 += sc * pu.x; b1 += sc * pu.y;
          }
        } else {
          const d2 px = lds_pair(&ws.Px[i][2 * tj]), pu = lds_pair(&ws.Pu[i][2 * tj]);
          a0 += sc * px.x; a1 += sc * px.y; b0 += sc * pu.x; b1 += sc * pu.y;
        }
      }
      out.At[i * NX + 2 * tj] = a0; out.At[i * NX + 2 * tj + 1] = a1;
      out.Bt[i * NU + 2 * tj] = b0; out.Bt[i * NU + 2 * tj + 1] = b1;
    }
  }
  if (lane < NX) {
    const int i = lane;
    double t = pvec[0];
    if (i >= 3 && i < 12) {
#pragma unroll
      for (int l = 0; l < NU; ++l) t += ws.Bd[i - 3][l] * ws.Pe[l];
    } else if (i < 3) {
      for (int c = 0; c < kNumContacts; ++c) t += ws.bscale[i] * ws.Pe[3 * c + i];
    } else {
      t += ws.bscale[i] * ws.Pe[i];
    }
    out.bt[i] = t;
    double s = pvec[1];                       // qt = q + Px' (r + R Pe)
#pragma unroll
    for (int l = 0; l < NU; ++l) s += ws.Px[l][i] * ws.rr[l];
    out.qt[i] = s;
    double u = 0.0;
    if (i < nut)
#pragma unroll
      for (int l = 0; l < NU; ++l) u += ws.Pu[l][i] * ws.rr[l];
    out.rt[i] = u;
  }
  PPROF(5);
  // Qt = Q + Px' R Px (H2 x H2 tiles, Q prefetched), Pt = Pu' R Px, Rt = Pu' R Pu
#pragma unroll
  for (int it = 0; it < IT_Q; ++it) {
    const int w = lane + it * kWave;
    if (w < H2 * H2) {
      const int ti = w / H2, tj = w % H2;
      const double* L = &ws.Px[0][2 * ti];
      const double* X = &ws.RPx[0][2 * tj];
      double c00 = pQ[it][0], c01 = pQ[it][1], c10 = pQ[it][2], c11 = pQ[it][3];
#pragma unroll
      for (int l = 0; l < NU; ++l) {
        const d2 lv = lds_pair(L + l * LD), xv = lds_pair(X + l * LD);
        c00 += lv.x * xv.x; c01 += lv.x * xv.y; c10 += lv.y * xv.x; c11 += lv.y * xv.y;
      }
      double* O = out.Qt + (2 * ti) * NX + 2 * tj;
      O[0] = c00; O[1] = c01; O[NX] = c10; O[NX + 1] = c11;
    }
  }
  for (int w = lane; w < 2 * H2 * H2; w += kWave) {
    const int which = 1 + w / (H2 * H2), t = w % (H2 * H2);
    const int ti = t / H2, tj = t % H2;
    double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
    if (ti < nt2 && (which == 1 || tj < nt2)) {
      const double* L = &ws.Pu[0][2 * ti];
      const double* X = which == 2 ? &ws.RPu[0][2 * tj] : &ws.RPx[0][2 * tj];
#pragma unroll
      for (int l = 0; l < NU; ++l) {
        const d2 lv = lds_pair(L + l * LD), xv = lds_pair(X + l * LD);
        c00 += lv.x * xv.x; c01 += lv.x * xv.y; c10 += lv.y * xv.x; c11 += lv.y * xv.y;
      }
    }
    const int i0 = 2 * ti, j0 = 2 * tj;
    const bool r0 = i0 < nut, r1 = i0 + 1 < nut;
    if (which == 1) {
      double* O = out.Pt + i0 * NX + j0;
      O[0] = r0 ? c00 : 0.0; O[1] = r0 ? c01 : 0.0; O[NX] = r1 ? c10 : 0.0; O[NX + 1] = r1 ? c11 : 0.0;
    } else {
      double* O = out.Rt + i0 * NU + j0;
      const bool q0 = j0 < nut, q1 = j0 + 1 < nut;
      O[0] = (r0 && q0) ? c00 : 0.0; O[1] = (r0 && q1) ? c01 : 0.0; O[NU] = (r1 && q0) ? c10 : 0.0; O[NU + 1] = (r1 && q1) ? c11 : 0.0;
    }
  }
  PPROF(6);
#ifdef BPMPC_PROJECT_PROFILE
  if (prof && lane == 0)
    for (int i = 0; i < 8; ++i) prof[i] = (double)tacc[i];
#endif
#ifdef BPMPC_PROJECT_PROFILE_LU
  if (prof && lane == 0)
    for (int i = 0; i < 8; ++i) prof[i] = (double)lacc[i];
#endif
}

}  // namespace bpmpc
