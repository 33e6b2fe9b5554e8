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

namespace bpmpc {

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
  double reg;                // settings.reg_prim: the terminal value function starts at reg * I (every other stage gets it from the projection kernel)
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

// du_k = K_k dx_k + kff_k for every stage in parallel, Armijo metric and step norms (dx is in HBM, workgroup-visible).
template <int NJ>
__device__ __forceinline__ void riccati_step_norms(int status, const RiccatiFastIO& io) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ, NT = kRiccatiThreads;
  constexpr int NXX = NX * NX, NXU = NX * NU;
  (void)NXX;
  const int tid = threadIdx.x;
  const int N = io.base.N;
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

}  // namespace bpmpc
