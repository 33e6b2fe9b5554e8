// Probe: latency of dependent FP64 instruction chains on one wave alone on its SIMD (not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/dep_probe.bin tools/probes/dep_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x * 1e-3, y = 1.0000001, z = 0.5;
  constexpr int N = 256;
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    if (MODE == 2) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
    if (MODE == 3) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(z));
    if (MODE == 4) { double t; asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(x)); asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x) : "v"(t), "v"(y)); }
    if (MODE == 5) { asm volatile("v_rcp_f64 %0, %0" : "+v"(x)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y)); }
    if (MODE == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(z));
    if (MODE == 7) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x));
    if (MODE == 8) { int e; asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(e) : "v"(x)); asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(e)); }
    if (MODE == 9) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(y) : "v"(y), "v"(z)); }   // two independent chains
  }
  long long t1 = clock64();
  out[threadIdx.x] = x + y;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  const char* names[] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rcp_f64 dependent", "s_nop 1 + v_fmac_f64_dpp dependent", "s_nop 1 + v_mov_b64_dpp + v_mul_f64", "v_rcp_f64 + v_mul_f64",
                         "v_add_f64 dependent", "v_ldexp_f64 dependent", "v_frexp_exp + v_ldexp", "two independent v_fma_f64 chains (per pair)"};
  for (int m = 0; m < 10; ++m) {
    for (int r = 0; r < 2; ++r) {
      switch (m) {
        case 0: k<0><<<1, 64>>>(out, cyc, 1.0); break; case 1: k<1><<<1, 64>>>(out, cyc, 1.0); break; case 2: k<2><<<1, 64>>>(out, cyc, 1.0); break;
        case 3: k<3><<<1, 64>>>(out, cyc, 1.0); break; case 4: k<4><<<1, 64>>>(out, cyc, 1.0); break; case 5: k<5><<<1, 64>>>(out, cyc, 1.0); break;
        case 6: k<6><<<1, 64>>>(out, cyc, 1.0); break; case 7: k<7><<<1, 64>>>(out, cyc, 1.0); break; case 8: k<8><<<1, 64>>>(out, cyc, 1.0); break;
        case 9: k<9><<<1, 64>>>(out, cyc, 1.0); break;
      }
      hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-48s %6.1f cycles per link (s_memtime units)\n", names[m], (double)c / 256);
  }
  return 0;
}
