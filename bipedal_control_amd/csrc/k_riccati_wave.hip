// Wave-per-problem Riccati sweeps (kernels/riccati_wave.h: one wave per SIMD; kernels/riccati_wave2.h: two) and the roll-out behind them.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernel_launchers.h"
#include "launch.h"
#include "kernels/riccati_mfma.h"
#include "kernels/riccati_wave.h"
#include "kernels/riccati_wave2.h"

namespace bpmpc {

// Batches larger than the chip (riccati_wave.h): one wavefront per problem, alone on its SIMD with the whole register file; the roll-out is
// a launch of its own.
template <int NJ, bool JW>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_riccati_wave(Launch L) {
  __shared__ RiccatiWaveWorkspace<NJ> ws;
  RiccatiFastIO io;
  if (!riccati_fast_io<NJ>(L, io, reinterpret_cast<double*>(&ws))) return;
  riccati_wave<NJ, JW>(ws, io);
}
// the same sweep arranged for two waves per SIMD (riccati_wave2.h)
template <int NJ, bool JW>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_riccati_wave2(Launch L) {
  __shared__ RiccatiWave2Workspace<NJ> ws;
  RiccatiFastIO io;
  if (!riccati_fast_io<NJ>(L, io, ws.T)) return;
  riccati_wave2<NJ, JW>(ws, io);
}

template <int NJ>
__global__ __launch_bounds__(kRolloutPairThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_riccati_rollout_pair(Launch L) {
  __shared__ RiccatiRolloutPairWorkspace<NJ> ws;
  RiccatiFastIO io;
  if (!riccati_fast_io<NJ>(L, io, ws.hist)) return;
  riccati_rollout_pair<NJ>(ws, io);
}

#define KL_NJ(nj, ...)                                                          \
  do {                                                                          \
    if ((nj) == 10) { constexpr int NJ = 10; __VA_ARGS__; }                     \
    else if ((nj) == 12) { constexpr int NJ = 12; __VA_ARGS__; }                \
    else throw std::runtime_error("unsupported joint count");                   \
  } while (0)

namespace kl {

// joint_rows false: Wt holds the rows 0..11 only, the sweep completes the joint rows from Vt (k_project_fast<.., false>)
void riccati_wave(int nj, bool two_per_simd, bool joint_rows, int batch, hipStream_t st, const Launch& L) {
  KL_NJ(nj, {
    if (two_per_simd && !joint_rows) hipLaunchKernelGGL((k_riccati_wave2<NJ, false>), dim3(batch), dim3(kWave), 0, st, L);
    else if (two_per_simd) hipLaunchKernelGGL((k_riccati_wave2<NJ, true>), dim3(batch), dim3(kWave), 0, st, L);
    else if (!joint_rows) hipLaunchKernelGGL((k_riccati_wave<NJ, false>), dim3(batch), dim3(kWave), 0, st, L);
    else hipLaunchKernelGGL((k_riccati_wave<NJ, true>), dim3(batch), dim3(kWave), 0, st, L);
  });
}
void riccati_rollout(int nj, int batch, hipStream_t st, const Launch& L) {
  KL_NJ(nj, hipLaunchKernelGGL(k_riccati_rollout_pair<NJ>, dim3(batch), dim3(kRolloutPairThreads), 0, st, L));
}

}  // namespace kl
}  // namespace bpmpc
