// Probe: cycles of the elimination variants of the Riccati sweep on one wave that has its SIMD to itself (not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bipedal_control_amd/csrc -o elim_probe.bin tools/probes/elim_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels/riccati_mfma8.h"
using namespace bpmpc;

template <int MODE, int ROWS>
__global__ void k(const double* Hin, double* out, long long* cyc, int nt, int reps) {
  __shared__ double M[16][50];
  __shared__ double Zt[16][34], Yn[16][34];
  __shared__ double ZY[16][2][34];
  const int l = threadIdx.x;
  for (int i = l; i < 16 * 50; i += 64) (&M[0][0])[i] = Hin[i];
  __syncthreads();
  constexpr int NX = 22, BC = 23;
  const bool rows_layout = MODE >= 2;
  const int rpr = 16 - nt, c16 = l & 15;
  const int rid = rows_layout ? (l >> 4) * rpr + (c16 - nt) : l - nt;
  const bool is_h = rows_layout ? c16 < nt : l < nt;
  const bool rhs = !is_h && rid < NX + 1;
  const bool used = is_h || rhs;
  const int col = is_h ? BC + (rows_layout ? c16 : l) : (rhs ? rid : 0);
  const int ecol = rhs ? col : NX + 2 + (l & 3);
  double acc = 0.0;
  long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    double v[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) { const double t = M[i][col]; v[i] = (used && i < nt) ? t : 0.0; }
    auto emit = [&](int p, double z, double y) { if (rhs) { Zt[p][col] = z; Yn[p][col] = y; } };
    bool ok = true;
    if (MODE == 0) ok = gauss_jordan_wave<ROWS>(v, nt);
    if (MODE == 1) { ok = forward_eliminate_wave<ROWS>(v, nt, emit); }
    if (MODE == 2) { ok = forward_eliminate_rows<ROWS>(v, nt, emit); }
    if (MODE == 6) { ok = forward_eliminate_rows<ROWS, true>(v, nt, emit); }
    if (MODE == 7) { auto emit2 = [&](int p, double z, double y) { ZY[p][0][ecol] = z; ZY[p][1][ecol] = y; }; ok = forward_eliminate_rows<ROWS, true>(v, nt, emit2); }
    if (MODE == 8) { auto emit3 = [&](int p, double z, double y) { Zt[p][ecol] = z; Yn[p][ecol] = y; }; ok = forward_eliminate_rows<ROWS, true>(v, nt, emit3); }
    if (MODE == 3) { ok = gauss_jordan_rows<ROWS>(v, nt); }
    if (MODE == 4) { ok = forward_eliminate_rows<ROWS>(v, nt, emit); back_substitute_rows<ROWS>(v, nt); }
    if (MODE == 5) { ok = forward_eliminate_wave<ROWS>(v, nt, emit); back_substitute_wave<ROWS>(v, nt); }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc += v[i];
    acc += ok ? 0.0 : 1.0;
    __syncthreads();
  }
  long long t1 = clock64();
  out[l] = acc + Zt[0][l & 15] + Yn[1][l & 15] + ZY[2][1][l & 15];
  if (l == 0) cyc[0] = (t1 - t0) / reps;
}

int main() {
  const int nt = 9;
  std::vector<double> H(16 * 50, 0.0);
  for (int i = 0; i < nt; ++i) {
    for (int j = 0; j < 23; ++j) H[i * 50 + j] = 0.01 * ((i * 7 + j * 3) % 11) - 0.05;
    for (int j = 0; j < nt; ++j) H[i * 50 + 23 + j] = (i == j ? 4.0 : 0.0) + 0.1 / (1 + i + j);
  }
  double *dH, *out; long long* cyc;
  hipMalloc(&dH, H.size() * 8); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice);
  const char* names[] = {"Gauss-Jordan, v_readlane", "forward elimination + emit, v_readlane", "forward elimination + emit, DPP rows", "Gauss-Jordan, DPP rows",
                         "forward + backward, DPP rows", "forward + backward, v_readlane", "forward elimination + emit, DPP rows, exact", "... emit into interleaved Z / Yn, every lane", "... emit into Z, Yn, every lane"};
  for (int mode = 0; mode < 9; ++mode) {
    for (int r = 0; r < 2; ++r) {
      switch (mode) {
        case 0: k<0, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break; case 1: k<1, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break;
        case 2: k<2, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break; case 3: k<3, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break;
        case 4: k<4, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break; case 5: k<5, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break;
        case 6: k<6, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break; case 7: k<7, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break; case 8: k<8, 9><<<1, 64>>>(dH, out, cyc, nt, 200); break;
      }
      hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double ho[64]; hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost); double cs = 0; for (int i = 0; i < 64; ++i) cs += ho[i] * (i + 1);
    printf("%-60s %lld cycles (nt = %d, incl. the LDS loads of the columns)  checksum %.17g\n", names[mode], c, nt, cs);
  }
  return 0;
}
