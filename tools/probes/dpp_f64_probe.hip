// Probe: issue cost of the DP-ALU DPP forms (v_fmac_f64_dpp / v_mov_b64_dpp row_newbcast) against plain v_fmac_f64 and the
// v_readlane_b32 pair they could replace, gfx950, one wave (not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 -o dpp_f64_probe dpp_f64_probe.hip ; run: ./dpp_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void k(double* out, long long* cyc, int reps) {
  const int l = threadIdx.x;
  double f[8];
  for (int c = 0; c < 8; ++c) f[c] = 1.0 + 1e-9 * (c + l);
  double m = 1e-12 * l;
  long long t0 = clock64();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %0, %1" : "+v"(f[c]) : "v"(m));
      if (MODE == 1) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(f[c]) : "v"(m));
      if (MODE == 2) asm volatile("v_fmac_f64_dpp %0, %0, -%1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(f[c]) : "v"(m));   // hazard-free here: 8 independent registers
      if (MODE == 3) { double t; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_e32 %2, %0, %3" : "=&v"(t), "+v"(f[c]) : "v"(f[c]), "v"(m)); }
      if (MODE == 4) { int lo, hi; asm volatile("v_readlane_b32 %0, %2, 3\n\tv_readlane_b32 %1, %3, 3" : "=s"(lo), "=s"(hi) : "v"(__double2loint(f[c])), "v"(__double2hiint(f[c])));
                       const double s = __hiloint2double(hi, lo); asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[c]) : "v"(m), "v"(s)); }
      if (MODE == 5) { double s = __shfl(f[c], 3, 16); asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[c]) : "v"(m), "v"(s)); }
    }
  }
  long long t1 = clock64();
  double s = 0; for (int c = 0; c < 8; ++c) s += f[c];
  out[l] = s;
  if (l == 0) cyc[0] = t1 - t0;
}

int main() {
  double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  const int reps = 2000;
  const char* names[] = {"v_fmac_f64", "s_nop 1 + v_fmac_f64_dpp", "v_fmac_f64_dpp", "v_mov_b64_dpp + v_fmac_f64", "2 v_readlane + v_fmac_f64 (sgpr operand)", "__shfl width 16 + v_fmac_f64"};
  for (int mode = 0; mode < 6; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (mode) {
        case 0: k<0><<<1, 64>>>(out, cyc, reps); break; case 1: k<1><<<1, 64>>>(out, cyc, reps); break; case 2: k<2><<<1, 64>>>(out, cyc, reps); break;
        case 3: k<3><<<1, 64>>>(out, cyc, reps); break; case 4: k<4><<<1, 64>>>(out, cyc, reps); break; case 5: k<5><<<1, 64>>>(out, cyc, reps); break;
      }
      hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-45s %.2f clock64 ticks per element (8 independent accumulators)\n", names[mode], (double)c / (reps * 8.0));
  }
  return 0;
}
