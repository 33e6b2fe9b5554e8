// Probe: lane layout and issue/latency cost of v_mfma_f64_16x16x4_f64 on gfx950 (not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip ; run: ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A /*16x4 row major*/, const double* B /*4x16 row major*/, double* D /*64 lanes x 4*/) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];
  const double b = B[(l / 16) * 16 + l % 16];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

__global__ void k_timing(double* out, long long* cyc, int reps) {
  const int l = threadIdx.x;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = clock64();
  for (int i = 0; i < reps; ++i) {      // dependent chain
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  }
  long long t1 = clock64();
  for (int i = 0; i < reps; ++i) {      // four independent chains
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  long long t2 = clock64();
  double f0 = a, f1 = b, f2 = a, f3 = b;
  for (int i = 0; i < reps; ++i) {      // dependent v_fma_f64 chain for reference
    f0 = f0 * a + b;
  }
  long long t3 = clock64();
  for (int i = 0; i < reps; ++i) { f0 = f0 * a + b; f1 = f1 * a + b; f2 = f2 * a + b; f3 = f3 * a + b; }
  long long t4 = clock64();
  out[l] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3;
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}


template <int CH>
__global__ void k_fma_chains(double* out, long long* cyc, int reps) {
  const int l = threadIdx.x;
  const double a = 1.0 + l * 1e-9, b = 1e-3;
  double f[CH];
  for (int c = 0; c < CH; ++c) f[c] = 1.0 + c;
  long long t0 = clock64();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) f[c] = f[c] * a + b;
  }
  long long t1 = clock64();
  double s = 0; for (int c = 0; c < CH; ++c) s += f[c];
  out[l] = s;
  if (l == 0) cyc[0] = t1 - t0;
}
template <int CH> void run_chains(double* dD, long long* dc) {
  long long c; const int reps = 500;
  k_fma_chains<CH><<<1, 64>>>(dD, dc, reps);
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("v_fma_f64, %2d independent chains, 1 wave: %.2f cycles per FMA\n", CH, (double)c / (reps * 4.0 * CH));
}

int main() {
  std::vector<double> A(64), B(64), D(256);
  for (int i = 0; i < 64; ++i) { A[i] = (rand() % 17) - 8; B[i] = (rand() % 13) - 6; }
  double *dA, *dB, *dD; long long* dc;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048); hipMalloc(&dc, 64);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  // candidate: lane l, reg r -> D[4*(l/16)+r][l%16]
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int i1 = 4 * (l / 16) + r, j1 = l % 16;
    int i2 = (l / 16) + 4 * r, j2 = l % 16;
    double e1 = 0, e2 = 0;
    for (int k = 0; k < 4; ++k) { e1 += A[i1 * 4 + k] * B[k * 16 + j1]; e2 += A[i2 * 4 + k] * B[k * 16 + j2]; }
    if (e1 != D[l * 4 + r]) ok1 = 0;
    if (e2 != D[l * 4 + r]) ok2 = 0;
  }
  printf("layout D[4*(l/16)+r][l%%16]: %d   layout D[(l/16)+4r][l%%16]: %d\n", ok1, ok2);
  long long c[4];
  const int reps = 1000;
  k_timing<<<1, 64>>>(dD, dc, reps);
  hipMemcpy(c, dc, 32, hipMemcpyDeviceToHost);
  printf("clock64 ticks per MFMA: dependent %.1f, 4 independent %.1f ; per v_fma_f64: dependent %.1f, 4 independent %.1f\n", (double)c[0] / reps,
         (double)c[1] / (4 * reps), (double)c[2] / reps, (double)c[3] / (4 * reps));
  // 4 waves on 4 SIMDs
  k_timing<<<1, 256>>>(dD, dc, reps);
  hipMemcpy(c, dc, 32, hipMemcpyDeviceToHost);
  printf("4 waves/CU: per MFMA dependent %.1f, independent %.1f ; fma %.1f %.1f\n", (double)c[0] / reps, (double)c[1] / (4 * reps), (double)c[2] / reps,
         (double)c[3] / (4 * reps));
  run_chains<1>(dD, dc); run_chains<2>(dD, dc); run_chains<4>(dD, dc); run_chains<8>(dD, dc); run_chains<16>(dD, dc); run_chains<32>(dD, dc);
  return 0;
}
