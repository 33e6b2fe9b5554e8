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
  static constexpr int NX = 12 + NJ, NU = 12 + NJ, LD = NX + 2;
  alignas(16) double Px[NU][LD], Pu[NU][LD], R[NU][LD], RPx[NU][LD], RPu[NU][LD];
  alignas(16) double Bd[9][LD];                    // dense rows 3..11 of B
  alignas(16) double Pe[NU], rr[NU];
  double bscale[NX];                               // the single entry of the sparse rows of B (rows 0..2, 12..)
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

  // ---- all HBM reads of this node, back to back (Px, Pu, Pe, nut come from the LU kernel, project_lu4.h)
  const int nut = out.nut[0];
  double pR[IT_R], pB[IT_B], pA[IT_A][2], pQ[IT_Q][4], pvec[3], pPx[IT_R], pPu[IT_R];
#pragma unroll
  for (int it = 0; it < IT_R; ++it) {
    const int idx = lane + it * kWave;
    const bool ok = idx < NU * NU;
    pR[it] = ok ? in.R[idx] : 0.0;
    pPx[it] = ok ? out.Px[idx] : 0.0;
    pPu[it] = ok ? out.Pu[idx] : 0.0;
  }
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
  const double pe = lane < NU ? out.Pe[lane] : 0.0;
  PPROF(0);
#pragma unroll
  for (int it = 0; it < IT_R; ++it) {
    const int idx = lane + it * kWave;
    if (idx < NU * NU) { ws.R[idx / NU][idx % NU] = pR[it]; ws.Px[idx / NU][idx % NU] = pPx[it]; ws.Pu[idx / NU][idx % NU] = pPu[it]; }
  }
#pragma unroll
  for (int it = 0; it < IT_B; ++it) { const int idx = lane + it * kWave; if (idx < 9 * NU) ws.Bd[idx / NU][idx % NU] = pB[it]; }
  if (lane < NX) ws.bscale[lane] = bsc;
  if (lane < NU) ws.Pe[lane] = pe;
  if (lane < 2 * NU) {                          // the padding columns are read by the 2x2 tiles
    double* rowp = lane < NU ? ws.Px[lane] : ws.Pu[lane - NU];
    rowp[NU] = 0.0; rowp[NU + 1] = 0.0;
  }
  lds_wave_sync();
  PPROF(3);
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
  }
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
