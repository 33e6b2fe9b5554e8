// Fast Riccati sweep (HIP only; the lane-emulated reference version is riccati.h - same mathematics).
//
// One 256-thread workgroup per problem.  What changes against riccati.h:
//   * the next stage's projected LQ model is prefetched from HBM into registers while the current stage is computed
//     (stage barriers order LDS only, so the loads stay in flight across them);
//   * every product is computed in 2x2 register tiles fed by 16-byte LDS reads (two reads per four FMAs, four
//     independent accumulation chains per thread); S is symmetric, so row pairs of S are read as column pairs;
//   * (H | G g) is reduced by Gauss-Jordan elimination held in the registers of wave 0 (one lane per column, pivot
//     column broadcast with v_readlane), while waves 1-3 accumulate Q + A'SA, which does not need the gains;
//   * the backward sweep stores the closed-loop quantities the roll-out needs
//        Acl = A~ + B~ Kt, bcl = b~ + B~ kt, K = Px + Pu Kt, kff = Pe + Pu kt, m = q~ + Kt' r~, m0 = r~' kt
//     so the roll-out is one mat-vec per stage (dx+ = Acl dx + bcl); du = K dx + kff is done afterwards for all
//     stages in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "../device_model.h"
#include "riccati.h"

namespace bpmpc {

template <int NJ>
struct RiccatiFastWorkspace {
  static constexpr int NX = 12 + NJ, NU = 12 + NJ;
  static constexpr int LD = NX + 4;            // even (16-byte rows); 2*LD*k mod 64 spreads row pairs over the banks
  static constexpr int HC = NU + (NU & 1);     // columns reserved for H in the augmented matrix (even)
  static constexpr int LDM = HC + NX + 2;      // [H | G g pad]
  static constexpr int LDG = LD + 2;
  alignas(16) double S[NX][LD];
  alignas(16) double A[NX][LD];
  alignas(16) double B[NX][LD];
  alignas(16) double Px[NU][LD];
  alignas(16) double Pu[NU][LD];
  alignas(16) double Q[NX][LD];
  alignas(16) double SA[NX][LD];
  alignas(16) double SB[NX][LD];
  alignas(16) double Sn[NX][LD];
  alignas(16) double M[NU][LDM];               // [R | P r] -> [H | G g] -> Y = H^-1 [G g] in the G part
  alignas(16) double G0[NU][LDG];              // [G g] before the elimination
  alignas(16) double s[NX], b[NX], q[NX], r[NU], Pe[NU], Sb[NX], sn[NX];
  alignas(16) double dx[2][NX];
  int status;
};

struct RiccatiFastIO {
  RiccatiIO base;            // same views as the reference kernel (Kt/kt unused)
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

// Forward roll-out dx_{k+1} = Acl_k dx_k + bcl_k (wave 0, one row per lane, next row prefetched), then
// du_k = K_k dx_k + kff_k for every stage in parallel, Armijo metric and step norms.
template <int NJ>
__device__ __forceinline__ void riccati_rollout(double (&wsdx)[2][12 + NJ], int status, const RiccatiFastIO& io) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ, NT = kRiccatiThreads;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  const int tid = threadIdx.x;
  const int N = io.base.N;
  // ---- forward roll-out: dx_{k+1} = Acl_k dx_k + bcl_k (wave 0, one row per lane, next row prefetched)
  if (tid < NX) { const double v = io.base.dx0[tid]; wsdx[0][tid] = v; io.base.dx[tid] = v; }
  __syncthreads();
  if (tid < kWave) {
    double row[NX], nrow[NX], bc = 0.0, nbc = 0.0;
    if (tid < NX && N > 0) {
#pragma unroll
      for (int l = 0; l < NX; ++l) row[l] = io.Acl[(size_t)tid * NX + l];
      bc = io.bcl[tid];
    }
    for (int k = 0; k < N; ++k) {
      if (tid < NX && k + 1 < N) {
        const double* nr = io.Acl + (size_t)(k + 1) * NXX + (size_t)tid * NX;
#pragma unroll
        for (int l = 0; l < NX; ++l) nrow[l] = nr[l];
        nbc = io.bcl[(size_t)(k + 1) * NX + tid];
      }
      if (tid < NX) {
        const double* cur = wsdx[k & 1];
        double t = bc;
#pragma unroll
        for (int l = 0; l < NX; ++l) t += row[l] * cur[l];
        wsdx[(k + 1) & 1][tid] = t;
        io.base.dx[(size_t)(k + 1) * NX + tid] = t;
#pragma unroll
        for (int l = 0; l < NX; ++l) row[l] = nrow[l];
        bc = nbc;
      }
      lds_wave_sync();
    }
  }
  __syncthreads();
  // ---- du_k = K_k dx_k + kff_k for every stage in parallel, Armijo metric and step norms
  double acc_arm = 0.0, acc_x = 0.0, acc_u = 0.0;
  for (int idx = tid; idx < N * NU; idx += NT) {
    const int k = idx / NU, i = idx % NU;
    const double* dxk = io.base.dx + (size_t)k * NX;
    const double* Kr = io.Kfull + (size_t)k * NXU + (size_t)i * NX;
    double t = io.kff[(size_t)k * NU + i];
#pragma unroll
    for (int l = 0; l < NX; ++l) t += Kr[l] * dxk[l];
    if (io.base.nut[k] == 0) t = 0.0;   // event node: no input
    io.base.du[idx] = t;
    acc_u += t * t;
    const double d = dxk[i];            // NU == NX: the same index walks the state vector
    acc_x += d * d;
    acc_arm += io.mvec[(size_t)k * NX + i] * d;
    if (i == 0) acc_arm += io.mscal[k];
  }
  if (tid < NX) { const double d = io.base.dx[(size_t)N * NX + tid]; acc_x += d * d; }
  __shared__ double red3[3][kRiccatiThreads / kWave];
  for (int off = kWave / 2; off >= 1; off >>= 1) {
    acc_arm += __shfl_down(acc_arm, off);
    acc_x += __shfl_down(acc_x, off);
    acc_u += __shfl_down(acc_u, off);
  }
  if ((tid & (kWave - 1)) == 0) { red3[0][tid / kWave] = acc_arm; red3[1][tid / kWave] = acc_x; red3[2][tid / kWave] = acc_u; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, x2 = 0.0, u2 = 0.0;
    for (int w = 0; w < kRiccatiThreads / kWave; ++w) { a += red3[0][w]; x2 += red3[1][w]; u2 += red3[2][w]; }
    io.base.summary[0] = a;
    io.base.summary[1] = x2;
    io.base.summary[2] = u2;
    io.base.summary[3] = (double)status;
  }
}

template <int NJ>
__device__ __forceinline__ void riccati_fast(RiccatiFastWorkspace<NJ>& ws, const RiccatiFastIO& io) {
  using WS = RiccatiFastWorkspace<NJ>;
  constexpr int NX = WS::NX, NU = WS::NU, NT = kRiccatiThreads, LD = WS::LD, HC = WS::HC, LDM = WS::LDM, LDG = WS::LDG;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  constexpr int E2 = (NXX + NT - 1) / NT;  // elements per thread of an NX*NX block
  constexpr int H2 = NX / 2;               // 2x2 tiles per dimension
  static_assert(NX == NU && (NX % 2) == 0, "tiling relies on nx == nu, even");
  constexpr int TRI = H2 * (H2 + 1) / 2;   // upper-triangular tiles of S
  static_assert(H2 * H2 + TRI + NX + 1 <= NT, "one P4 work item per thread");
  static_assert(HC + NX + 1 <= kWave, "one lane per column of the augmented matrix");
  const int tid = threadIdx.x;
  const int N = io.base.N;

  const int k_top = (io.k_hi < N ? io.k_hi : N) - 1;   // first stage of this launch
  const bool resumed = io.k_hi < N;                    // a later chunk has already run: take over its value function
  for (int idx = tid; idx < NX * LD; idx += NT) { (&ws.S[0][0])[idx] = 0.0; (&ws.B[0][0])[idx] = 0.0; (&ws.Pu[0][0])[idx] = 0.0; (&ws.SB[0][0])[idx] = 0.0; }
  for (int idx = tid; idx < NU * LDM; idx += NT) (&ws.M[0][0])[idx] = 0.0;
  for (int idx = tid; idx < NU * LDG; idx += NT) (&ws.G0[0][0])[idx] = 0.0;
  if (tid < NX) ws.s[tid] = resumed ? io.carry[NXX + tid] : 0.0;
  if (tid == 0) ws.status = resumed ? (int)io.carry[NXX + NX] : 0;
  if (resumed) {
    __syncthreads();
    for (int idx = tid; idx < NXX; idx += NT) ws.S[idx / NX][idx % NX] = io.carry[idx];
  }

  // registers holding the prefetched stage
  double pA[E2], pB[E2], pQ[E2], pP[E2], pR[E2], pPx[E2], pPu[E2], pv[4];
  auto prefetch = [&](int k) {
    const size_t o2 = (size_t)k * NXX;
#pragma unroll
    for (int e = 0; e < E2; ++e) {
      const int idx = tid + e * NT;
      const bool in = idx < NXX;
      pA[e] = in ? io.base.At[o2 + idx] : 0.0;
      pB[e] = in ? io.base.Bt[o2 + idx] : 0.0;
      pQ[e] = in ? io.base.Qt[o2 + idx] : 0.0;
      pP[e] = in ? io.base.Pt[o2 + idx] : 0.0;
      pR[e] = in ? io.base.Rt[o2 + idx] : 0.0;
      pPx[e] = in ? io.base.Px[o2 + idx] : 0.0;
      pPu[e] = in ? io.base.Pu[o2 + idx] : 0.0;
    }
    if (tid < NX) {
      pv[0] = io.base.bt[(size_t)k * NX + tid];
      pv[1] = io.base.qt[(size_t)k * NX + tid];
      pv[2] = io.base.rt[(size_t)k * NU + tid];
      pv[3] = io.base.Pe[(size_t)k * NU + tid];
    }
  };
  if (k_top >= io.k_lo) prefetch(k_top);
  __syncthreads();
#ifdef BPMPC_RICCATI_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define RPROF(slot) do { const long long tn_ = clock64(); tacc[slot] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define RPROF(slot) ((void)0)
#endif

  for (int k = k_top; k >= io.k_lo; --k) {
    const int nt = io.base.nut[k];
    const int nt2 = (nt + 1) / 2;
    // ---- P0: registers -> LDS; M = [R | P r].  The projection kernel writes zeros beyond nt in B~, Pu, R~, P~, r~,
    //      so rows/columns >= nt of the staged blocks are zero and odd nt can be processed in pairs.
#pragma unroll
    for (int e = 0; e < E2; ++e) {
      const int idx = tid + e * NT;
      if (idx < NXX) {
        const int i = idx / NX, j = idx % NX;
        ws.A[i][j] = pA[e];
        ws.B[i][j] = pB[e];
        ws.Q[i][j] = pQ[e];
        ws.Px[i][j] = pPx[e];
        ws.Pu[i][j] = pPu[e];
        ws.M[i][HC + j] = pP[e];
        ws.M[i][j] = pR[e];
      }
    }
    if (tid < NX) { ws.b[tid] = pv[0]; ws.q[tid] = pv[1]; ws.r[tid] = pv[2]; ws.Pe[tid] = pv[3]; ws.M[tid][HC + NX] = pv[2]; }
    lds_barrier();
    RPROF(0);
    if (k > io.k_lo) prefetch(k - 1);   // never beyond the chunk: earlier stages may not be projected yet
    RPROF(1);
    // ---- P1: SA = S A (H2 x H2 tiles), SB = S B (H2 x nt2 tiles), Sb = S b + s (H2 row pairs)
    for (int w = tid; w < 2 * H2 * H2 + H2; w += NT) {
    if (w < 2 * H2 * H2) {
      const bool second = w >= H2 * H2;
      const int t = second ? w - H2 * H2 : w;
      const int ti = t / H2, tj = t % H2;
      if (!second || tj < nt2) {
        const double* X = second ? &ws.B[0][2 * tj] : &ws.A[0][2 * tj];
        const double* Sr = &ws.S[0][2 * ti];          // S symmetric: rows (2ti, 2ti+1) read as a column pair
        double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          const d2 sv = lds_pair(Sr + l * LD), xv = lds_pair(X + l * LD);
          c00 += sv.x * xv.x; c01 += sv.x * xv.y; c10 += sv.y * xv.x; c11 += sv.y * xv.y;
        }
        double* O = second ? &ws.SB[2 * ti][2 * tj] : &ws.SA[2 * ti][2 * tj];
        O[0] = c00; O[1] = c01; O[LD] = c10; O[LD + 1] = c11;
      }
    } else {
      const int ti = w - 2 * H2 * H2;
      const double* Sr = &ws.S[0][2 * ti];
      double c0 = ws.s[2 * ti], c1 = ws.s[2 * ti + 1];
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        const d2 sv = lds_pair(Sr + l * LD);
        const double bv = ws.b[l];
        c0 += sv.x * bv; c1 += sv.y * bv;
      }
      ws.Sb[2 * ti] = c0; ws.Sb[2 * ti + 1] = c1;
    }
    }
    lds_barrier();
    RPROF(2);
    // ---- P2: [H | G g] = [R | P r] + B' [SB | SA Sb]: nt2 row pairs x (nt2 + H2 + 1) column pairs, the last pair is (g, pad)
    {
      const int wt = nt2 + H2 + 1;
      const int tj = tid % 32;                       // up to 32 column pairs, 8 row pairs per pass
      if (tj < wt)
        for (int rp = tid / 32; rp < nt2; rp += NT / 32) {
          const double* Bl = &ws.B[0][2 * rp];
          double c00, c01, c10, c11;
          double* O0;
          if (tj < nt2 + H2) {
            const bool hpart = tj < nt2;
            const double* X = hpart ? &ws.SB[0][2 * tj] : &ws.SA[0][2 * (tj - nt2)];
            O0 = hpart ? &ws.M[2 * rp][2 * tj] : &ws.M[2 * rp][HC + 2 * (tj - nt2)];
            c00 = O0[0]; c01 = O0[1]; c10 = O0[LDM]; c11 = O0[LDM + 1];
#pragma unroll
            for (int l = 0; l < NX; ++l) {
              const d2 bv = lds_pair(Bl + l * LD), xv = lds_pair(X + l * LD);
              c00 += bv.x * xv.x; c01 += bv.x * xv.y; c10 += bv.y * xv.x; c11 += bv.y * xv.y;
            }
          } else {
            O0 = &ws.M[2 * rp][HC + NX];
            c00 = O0[0]; c10 = O0[LDM]; c01 = 0.0; c11 = 0.0;
#pragma unroll
            for (int l = 0; l < NX; ++l) {
              const d2 bv = lds_pair(Bl + l * LD);
              const double xv = ws.Sb[l];
              c00 += bv.x * xv; c10 += bv.y * xv;
            }
          }
          O0[0] = c00; O0[1] = c01; O0[LDM] = c10; O0[LDM + 1] = c11;
          if (tj >= nt2) {
            double* Gc = &ws.G0[2 * rp][2 * (tj - nt2)];
            Gc[0] = c00; Gc[1] = c01; Gc[LDG] = c10; Gc[LDG + 1] = c11;
          }
        }
    }
    lds_barrier();
    RPROF(3);
    // ---- P3: wave 0: Gauss-Jordan in registers; waves 1..3: Sn = Q + A' SA, sn = q + A' Sb
    if (tid < kWave) {
      const int lane = tid;
      const int col = lane < nt ? lane : HC + (lane - nt);      // lanes nt .. nt+NX hold G | g
      const bool used = lane < nt + NX + 1;
      bool ok;
      if (nt <= 12) {
        double v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = (used && i < nt) ? ws.M[i][col] : 0.0;
        ok = gauss_jordan_wave<12>(v, nt);
#pragma unroll
        for (int i = 0; i < 12; ++i) if (used && i < nt && lane >= nt) ws.M[i][col] = v[i];
      } else {
        double v[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) v[i] = (used && i < nt) ? ws.M[i][col] : 0.0;
        ok = gauss_jordan_wave<NU>(v, nt);
#pragma unroll
        for (int i = 0; i < NU; ++i) if (used && i < nt && lane >= nt) ws.M[i][col] = v[i];
      }
      if (lane == 0 && !ok) ws.status = 1;
    } else {
      for (int t3 = tid - kWave; t3 < H2 * H2 + H2; t3 += NT - kWave) {
      if (t3 < H2 * H2) {
        const int ti = t3 / H2, tj = t3 % H2;
        const double* Al = &ws.A[0][2 * ti];
        const double* X = &ws.SA[0][2 * tj];
        double c00 = ws.Q[2 * ti][2 * tj], c01 = ws.Q[2 * ti][2 * tj + 1], c10 = ws.Q[2 * ti + 1][2 * tj], c11 = ws.Q[2 * ti + 1][2 * tj + 1];
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          const d2 av = lds_pair(Al + l * LD), xv = lds_pair(X + l * LD);
          c00 += av.x * xv.x; c01 += av.x * xv.y; c10 += av.y * xv.x; c11 += av.y * xv.y;
        }
        double* O = &ws.Sn[2 * ti][2 * tj];
        O[0] = c00; O[1] = c01; O[LD] = c10; O[LD + 1] = c11;
      } else {
        const int ti = t3 - H2 * H2;
        const double* Al = &ws.A[0][2 * ti];
        double c0 = ws.q[2 * ti], c1 = ws.q[2 * ti + 1];
#pragma unroll
        for (int l = 0; l < NX; ++l) {
          const d2 av = lds_pair(Al + l * LD);
          const double xv = ws.Sb[l];
          c0 += av.x * xv; c1 += av.y * xv;
        }
        ws.sn[2 * ti] = c0; ws.sn[2 * ti + 1] = c1;
      }
      }
    }
    lds_barrier();
    RPROF(4);
    // now Y = H^-1 [G g] sits in M[0..nt)[HC ..];  Kt = -Y.  Rows nt..2*nt2 of the Y part are zero (never written).
    // ---- P4: Acl = A - B Y, K = Px - Pu Y (tiles); S <- sym(Sn - G0' Y) (tiles ti <= tj, both triangles); vectors
    double* Acl = io.Acl + (size_t)k * NXX;
    double* Kf = io.Kfull + (size_t)k * NXU;
    if (tid < H2 * H2) {
      const int ti = tid / H2, tj = tid % H2;
      double a00 = ws.A[2 * ti][2 * tj], a01 = ws.A[2 * ti][2 * tj + 1], a10 = ws.A[2 * ti + 1][2 * tj], a11 = ws.A[2 * ti + 1][2 * tj + 1];
      double k00 = ws.Px[2 * ti][2 * tj], k01 = ws.Px[2 * ti][2 * tj + 1], k10 = ws.Px[2 * ti + 1][2 * tj], k11 = ws.Px[2 * ti + 1][2 * tj + 1];
      for (int l2 = 0; l2 < nt2; ++l2) {
        const d2 b0 = lds_pair(&ws.B[2 * ti][2 * l2]), b1 = lds_pair(&ws.B[2 * ti + 1][2 * l2]);
        const d2 u0 = lds_pair(&ws.Pu[2 * ti][2 * l2]), u1 = lds_pair(&ws.Pu[2 * ti + 1][2 * l2]);
        const d2 y0 = lds_pair(&ws.M[2 * l2][HC + 2 * tj]), y1 = lds_pair(&ws.M[2 * l2 + 1][HC + 2 * tj]);
        a00 -= b0.x * y0.x + b0.y * y1.x; a01 -= b0.x * y0.y + b0.y * y1.y;
        a10 -= b1.x * y0.x + b1.y * y1.x; a11 -= b1.x * y0.y + b1.y * y1.y;
        k00 -= u0.x * y0.x + u0.y * y1.x; k01 -= u0.x * y0.y + u0.y * y1.y;
        k10 -= u1.x * y0.x + u1.y * y1.x; k11 -= u1.x * y0.y + u1.y * y1.y;
      }
      double* Ao = Acl + (2 * ti) * NX + 2 * tj;
      Ao[0] = a00; Ao[1] = a01; Ao[NX] = a10; Ao[NX + 1] = a11;
      double* Ko = Kf + (2 * ti) * NX + 2 * tj;
      Ko[0] = k00; Ko[1] = k01; Ko[NX] = k10; Ko[NX + 1] = k11;
    } else if (tid < H2 * H2 + TRI) {
      int t = tid - H2 * H2, ti = 0;
      while (t >= H2 - ti) { t -= H2 - ti; ++ti; }
      const int tj = ti + t;
      {
        // p(i,j) = Sn(i,j) - sum_l G0(l,i) Y(l,j)   and the transposed block   r(j,i) = Sn(j,i) - sum_l G0(l,j) Y(l,i)
        double p00 = ws.Sn[2 * ti][2 * tj], p01 = ws.Sn[2 * ti][2 * tj + 1], p10 = ws.Sn[2 * ti + 1][2 * tj], p11 = ws.Sn[2 * ti + 1][2 * tj + 1];
        double r00 = ws.Sn[2 * tj][2 * ti], r01 = ws.Sn[2 * tj][2 * ti + 1], r10 = ws.Sn[2 * tj + 1][2 * ti], r11 = ws.Sn[2 * tj + 1][2 * ti + 1];
        for (int l = 0; l < nt; ++l) {
          const d2 gi = lds_pair(&ws.G0[l][2 * ti]), gj = lds_pair(&ws.G0[l][2 * tj]);
          const d2 yi = lds_pair(&ws.M[l][HC + 2 * ti]), yj = lds_pair(&ws.M[l][HC + 2 * tj]);
          p00 -= gi.x * yj.x; p01 -= gi.x * yj.y; p10 -= gi.y * yj.x; p11 -= gi.y * yj.y;
          r00 -= gj.x * yi.x; r01 -= gj.x * yi.y; r10 -= gj.y * yi.x; r11 -= gj.y * yi.y;
        }
        const double s00 = 0.5 * (p00 + r00), s01 = 0.5 * (p01 + r10), s10 = 0.5 * (p10 + r01), s11 = 0.5 * (p11 + r11);
        ws.S[2 * ti][2 * tj] = s00; ws.S[2 * ti][2 * tj + 1] = s01; ws.S[2 * ti + 1][2 * tj] = s10; ws.S[2 * ti + 1][2 * tj + 1] = s11;
        ws.S[2 * tj][2 * ti] = s00; ws.S[2 * tj + 1][2 * ti] = s01; ws.S[2 * tj][2 * ti + 1] = s10; ws.S[2 * tj + 1][2 * ti + 1] = s11;
      }
    } else if (tid < H2 * H2 + TRI + NX) {
      const int i = tid - H2 * H2 - TRI;
      double t = ws.sn[i];
      double mv = ws.q[i], bc = ws.b[i], kf = ws.Pe[i];
      for (int l = 0; l < nt; ++l) {
        const double hg = ws.M[l][HC + NX];
        t -= ws.G0[l][i] * hg;
        mv -= ws.M[l][HC + i] * ws.r[l];     // Kt' r~ with Kt = -Y
        bc -= ws.B[i][l] * hg;
        kf -= ws.Pu[i][l] * hg;
      }
      ws.s[i] = t;
      io.mvec[(size_t)k * NX + i] = mv;
      io.bcl[(size_t)k * NX + i] = bc;
      io.kff[(size_t)k * NU + i] = kf;
    } else if (tid == NT - 1) {
      double m0 = 0.0;
      for (int l = 0; l < nt; ++l) m0 -= ws.r[l] * ws.M[l][HC + NX];
      io.mscal[k] = m0;
    }
    RPROF(5);
    lds_barrier();
    // clean-up for odd nt: the pair partner row nt of Y must read as zero in the next use (it holds P~ = 0 already,
    // but the H part of row nt and G0 may carry stale pairs) - zero the rows [nt, 2*nt2) of M and G0.
    if ((nt & 1) && tid < LDM) { ws.M[nt][tid] = 0.0; if (tid < LDG) ws.G0[nt][tid] = 0.0; }
    RPROF(6);
  }
#ifdef BPMPC_RICCATI_PROFILE
  if (io.prof && tid == 0)
    for (int i = 0; i < 8; ++i) io.prof[i] = (double)tacc[i];
#endif
  __syncthreads();
  if (io.k_lo > 0) {                                   // hand over to the launch that sweeps the earlier stages
    for (int idx = tid; idx < NXX; idx += NT) io.carry[idx] = ws.S[idx / NX][idx % NX];
    if (tid < NX) io.carry[NXX + tid] = ws.s[tid];
    if (tid == 0) io.carry[NXX + NX] = (double)ws.status;
    return;
  }

  riccati_rollout<NJ>(ws.dx, ws.status, io);
}

}  // namespace bpmpc
