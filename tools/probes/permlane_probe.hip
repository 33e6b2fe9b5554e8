// v_permlane16_swap_b32 (gfx950): what lands where.  hipcc --offload-arch=gfx950 -O3 tools/probes/permlane_probe.hip -o /tmp/pp && /tmp/pp
// Used by the roll-out of the workgroup sweeps (riccati_mfma.h): new state elements computed in DPP rows 0 (low half) and 1 (high half) become two
// registers that hold the low / the high half in BOTH rows.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
  auto w = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + threadIdx.x] = w[0]; out[192 + threadIdx.x] = w[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) { std::printf("%s result %d:", r < 2 ? "permlane16_swap" : "permlane32_swap", r & 1); for (int i = 0; i < 64; i += 4) std::printf(" %u", h[64 * r + i]); std::printf("\n"); }
  return 0;
}
