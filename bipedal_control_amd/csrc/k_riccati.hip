// Workgroup-per-problem Riccati sweeps: reference body (kernels/riccati.h), four waves (kernels/riccati_mfma.h), eight waves with fixed roles (kernels/riccati_mfma8.h).
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernel_launchers.h"
#include "launch.h"
#include "kernels/riccati.h"
#include "kernels/riccati_mfma.h"
#include "kernels/riccati_mfma8.h"

namespace bpmpc {

template <int NJ>
__global__ __launch_bounds__(kRiccatiThreads) void k_riccati(Launch L) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  __shared__ RiccatiWorkspace<NJ> ws;
  const int b = blockIdx.x;
  if (!L.buf.active[b]) return;
  const size_t s0 = (size_t)b * L.N;
  // dx0 = x_measured - x_0
  double* dx0 = L.buf.dx0 + (size_t)b * NX;
  if (threadIdx.x < NX) dx0[threadIdx.x] = L.buf.p_x0[(size_t)b * NX + threadIdx.x] - L.buf.x[(size_t)b * (L.N + 1) * NX + threadIdx.x];
  __syncthreads();
  RiccatiIO io;
  io.N = L.buf.g_nodes[L.buf.p_grid[b]];
  io.nut = L.buf.nut + s0;
  io.At = L.buf.At + s0 * NX * NX; io.Bt = L.buf.Bt + s0 * NX * NU; io.bt = L.buf.bt + s0 * NX;
  io.Qt = L.buf.Qt + s0 * NX * NX; io.Rt = L.buf.Rt + s0 * NU * NU; io.Pt = L.buf.Pt + s0 * NU * NX; io.qt = L.buf.qt + s0 * NX;
  io.rt = L.buf.rt + s0 * NU;
  io.Px = L.buf.Px + s0 * NU * NX; io.Pu = L.buf.Pu + s0 * NU * NU; io.Pe = L.buf.Pe + s0 * NU;
  io.dx0 = dx0;
  io.Kt = L.buf.Kt + s0 * NU * NX; io.kt = L.buf.kt + s0 * NU;
  io.dx = L.buf.dx + (size_t)b * (L.N + 1) * NX; io.du = L.buf.du + s0 * NU;
  io.K = L.buf.K ? L.buf.K + s0 * NU * NX : nullptr;
  io.summary = L.buf.summary + (size_t)b * 4;
  riccati_problem<NJ>(ws, io);
}

// The single-buffered variant is meant to run two workgroups per CU: cap its registers at 256 (VGPR + AGPR).
template <int NJ, bool DB, bool JW>
__global__ __launch_bounds__(kRiccatiThreads) __attribute__((amdgpu_waves_per_eu(DB ? 1 : 2, DB ? 8 : 2))) void k_riccati_fast(Launch L) {
  __shared__ RiccatiMfmaWorkspace<NJ, DB> ws;
  RiccatiFastIO io;
  if (!riccati_fast_io<NJ>(L, io, reinterpret_cast<double*>(&ws))) return;
  riccati_mfma<NJ, DB, JW>(ws, io);
}

// Eight waves per problem with fixed roles (riccati_mfma8.h): one workgroup per CU.
template <int NJ, bool JW>
__global__ __launch_bounds__(kRiccati8Threads) void k_riccati_fast8(Launch L) {
  __shared__ RiccatiMfma8Workspace<NJ> ws;
  RiccatiFastIO io;
  if (!riccati_fast_io<NJ>(L, io, reinterpret_cast<double*>(&ws))) return;
  riccati_mfma8<NJ, JW>(ws, io);
}

#define KL_NJ(nj, ...)                                                          \
  do {                                                                          \
    if ((nj) == 10) { constexpr int NJ = 10; __VA_ARGS__; }                     \
    else if ((nj) == 12) { constexpr int NJ = 12; __VA_ARGS__; }                \
    else throw std::runtime_error("unsupported joint count");                   \
  } while (0)

namespace kl {

void riccati_reference(int nj, int batch, hipStream_t st, const Launch& L) { KL_NJ(nj, hipLaunchKernelGGL(k_riccati<NJ>, dim3(batch), dim3(kRiccatiThreads), 0, st, L)); }
void riccati_fast(int nj, bool double_buffered, bool joint_rows, int batch, hipStream_t st, const Launch& L) {
  KL_NJ(nj, {
    if (double_buffered) {
      if (!joint_rows) throw std::runtime_error("riccati_fast: the double-buffered four-wave kernel reads the joint rows of Wt");
      hipLaunchKernelGGL((k_riccati_fast<NJ, true, true>), dim3(batch), dim3(kRiccatiThreads), 0, st, L);
    } else if (joint_rows) hipLaunchKernelGGL((k_riccati_fast<NJ, false, true>), dim3(batch), dim3(kRiccatiThreads), 0, st, L);
    else hipLaunchKernelGGL((k_riccati_fast<NJ, false, false>), dim3(batch), dim3(kRiccatiThreads), 0, st, L);
  });
}
void riccati_fast8(int nj, bool joint_rows, int batch, hipStream_t st, const Launch& L) {
  KL_NJ(nj, {
    if (joint_rows) hipLaunchKernelGGL((k_riccati_fast8<NJ, true>), dim3(batch), dim3(kRiccati8Threads), 0, st, L);
    else hipLaunchKernelGGL((k_riccati_fast8<NJ, false>), dim3(batch), dim3(kRiccati8Threads), 0, st, L);
  });
}

}  // namespace kl
}  // namespace bpmpc
