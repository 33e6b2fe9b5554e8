// What a write-only stream achieves on this part: the calibration of `roofline.frac` for the lineariser, a kernel whose HBM traffic is 97 %
// stores (profiles/r05_traffic.json: 642 MB written, 17 MB read per launch).  No arithmetic, no loads, same bytes:
//   fill16   : plain fill, 16 B per lane, whole 128-byte lines per 8 lanes (the best case of a store stream)
//   fill8    : plain fill, 8 B per lane
//   pattern  : the lineariser's own store pattern - 4 nodes per wave, 16 lanes per node, per output row of [A|B], [C|D], [Q|R] three store
//              instructions of 8 B per lane through the role pointers of linearize_fast.h:RoleSlots (unaligned 48 .. 128-byte pieces of
//              176-byte rows, the dump line for lanes without a role), then e, b, q, r, the compact Q / R record and the scalars - at the
//              lineariser's geometry (256-thread workgroups, 76 KB of LDS each: two per CU) and without the LDS (occupancy by registers only)
// Standalone:  hipcc --offload-arch=gfx950 -O3 tools/probes/write_roof.hip -o /tmp/write_roof && /tmp/write_roof [batch] [nodes per problem] [nx: 22 | 24]
// As a library (bench.py loads it over ctypes and reports `write_roof_GBs` beside roofline.frac):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -DWRITE_ROOF_LIB tools/probes/write_roof.hip -o tools/probes/libwrite_roof.so
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

constexpr int EQ = 16, QRD = 40;      // NX = NU = 22 (H1: the shape the roofline unit is defined on) or 24 (G1 class)
constexpr int kDumpNodes = 1024;

__global__ __launch_bounds__(256) void k_fill16(double2* p, size_t n2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = i; j < n2; j += stride) p[j] = double2{(double)j, 1.0};
}
__global__ __launch_bounds__(256) void k_fill8(double* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = i; j < n; j += stride) p[j] = (double)j;
}

struct Out {
  double *A, *B, *b, *Q, *R, *q, *r, *c, *C, *D, *e, *perf, *qrd, *dump;
  int* nc;
};

// the role pointers of an unpacked node (16 coordinates): slot 0 = x column 6 + ln | slot 1 = momentum column (lanes 0..5) or joint-velocity
// column | slot 2 = force column (lanes 0..11) or the dump
template <int LDS_BYTES, int NX, bool NT = false>
__global__ __launch_bounds__(256) void k_pattern(Out o, int nodes) {
  constexpr int NU = NX;
  __shared__ char pad[LDS_BYTES > 0 ? LDS_BYTES : 1];
  if (LDS_BYTES > 0 && threadIdx.x == 0) pad[blockIdx.x % LDS_BYTES] = 1;      // keep the allocation
  const int ln = threadIdx.x % 16;
  const size_t s = (size_t)blockIdx.x * 16 + threadIdx.x / 16;
  if (s >= (size_t)nodes) return;
  double* dump = o.dump + ((s & (size_t)(kDumpNodes - 1)) * 16 + ln);
  const double v = (double)(s + ln);
  auto rows = [&](double* X, double* U, int nrows) {
    if constexpr (NX == 22) {
      double* p0 = X + (6 + ln);
      double* p1 = ln < 6 ? X + ln : U + (12 + ln - 6);
      double* p2 = ln < 12 ? U + ln : dump;
#pragma unroll
      for (int r = 0; r < nrows; ++r) {
        if constexpr (NT) { __builtin_nontemporal_store(v, p0 + r * NX); __builtin_nontemporal_store(v, p1 + r * NX); __builtin_nontemporal_store(v, p2 + r * NX); }
        else { p0[r * NX] = v; p1[r * NX] = v; p2[r * NX] = v; }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {                      // packed lanes (linearize_fast.h RoleSlots, G0 = 3): lane ln carries coordinate g = ln + 3; four role slots
      const int g = ln + 3, G = NX - 6;
      double* p0 = g < G ? X + (6 + g) : dump;
      double* p1 = ln < 3 ? X + (6 + ln) : (g < G ? U + (12 + g - 6) : dump);
      double* p2 = ln < 12 ? U + ln : dump;
      double* p3 = ln < 6 ? X + ln : dump;
#pragma unroll
      for (int r = 0; r < nrows; ++r) {
        p0[r * NX] = v; p1[r * NX] = v; p2[r * NX] = v; p3[r * NX] = v;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  rows(o.C + s * (EQ * NX), o.D + s * (EQ * NU), EQ);
  o.e[s * EQ + ln] = v;
  rows(o.A + s * (NX * NX), o.B + s * (NX * NU), NX);
  o.b[s * NX + 6 + ln] = v;
  if (ln < 6) o.b[s * NX + ln] = v;
  if (NX > 22 && ln < NX - 22) o.b[s * NX + 22 + ln] = v;
  rows(o.Q + s * (NX * NX), o.R + s * (NU * NU), NX);
  if (ln < 12) for (int a = 0; a < 3; ++a) o.qrd[s * QRD + 1 + 3 * ln + a] = v;
  o.q[s * NX + 6 + ln] = v;
  if (ln < 6) o.q[s * NX + ln] = v;
  if (NX > 22 && ln < NX - 22) { o.q[s * NX + 22 + ln] = v; o.r[s * NU + 22 + ln] = v; }
  if (ln < 12) o.r[s * NU + ln] = v;
  if (ln >= 6) o.r[s * NU + 12 + ln - 6] = v;
  if (ln == 0) { o.qrd[s * QRD] = v; o.c[s] = v; o.nc[s] = 12; o.perf[s * 3] = v; o.perf[s * 3 + 1] = v; o.perf[s * 3 + 2] = v; }
}

struct Roof {
  double fill16_GBs, fill8_GBs, pattern_GBs, pattern_free_GBs, pattern_nt_GBs;
  double bytes_fill, bytes_pattern;
  double pattern_ms, pattern_free_ms, fill16_ms;
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class F>
int time_ms(F&& launch, int reps, float* ms) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  CK(hipEventElapsedTime(ms, a, b));
  *ms /= reps;
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return 0;
}

template <int NX>
int measure(int batch, int nodes_per_problem, Roof* out) {
  constexpr int NU = NX;
  const size_t nodes = (size_t)batch * nodes_per_problem;
  // bytes a node's stores carry (dump lanes excluded): the lineariser's output, 21 784 B at H1 + the 320-byte record's written part
  const size_t per_node = (size_t)(EQ * (NX + NU) + EQ + NX * (NX + NU) + NX + NX * (NX + NU) + NX + NU + 37 + 4) * 8 + 4;
  const size_t total = nodes * per_node;
  Out o;
  auto alloc = [&](double** p, size_t per) { return hipMalloc(p, nodes * per * sizeof(double)); };
  CK(alloc(&o.A, NX * NX)); CK(alloc(&o.B, NX * NU)); CK(alloc(&o.b, NX)); CK(alloc(&o.Q, NX * NX)); CK(alloc(&o.R, NU * NU));
  CK(alloc(&o.q, NX)); CK(alloc(&o.r, NU)); CK(alloc(&o.c, 1)); CK(alloc(&o.C, EQ * NX)); CK(alloc(&o.D, EQ * NU)); CK(alloc(&o.e, EQ));
  CK(alloc(&o.perf, 3)); CK(alloc(&o.qrd, QRD));
  CK(hipMalloc(&o.dump, ((size_t)kDumpNodes * 16 + 1024) * sizeof(double)));
  CK(hipMalloc(&o.nc, nodes * sizeof(int)));
  double* flat;
  CK(hipMalloc(&flat, total));
  const int reps = 20;
  float ms;
  const int fill_grid = 256 * 8;
  if (time_ms([&] { hipLaunchKernelGGL(k_fill16, dim3(fill_grid), dim3(256), 0, 0, reinterpret_cast<double2*>(flat), total / 16); }, reps, &ms)) return 1;
  out->fill16_ms = ms; out->fill16_GBs = total / 16 * 16 / (ms * 1e6);
  if (time_ms([&] { hipLaunchKernelGGL(k_fill8, dim3(fill_grid), dim3(256), 0, 0, flat, total / 8); }, reps, &ms)) return 1;
  out->fill8_GBs = total / 8 * 8 / (ms * 1e6);
  const int grid = (int)((nodes + 15) / 16);
  if (time_ms([&] { hipLaunchKernelGGL((k_pattern<76264, NX>), dim3(grid), dim3(256), 0, 0, o, (int)nodes); }, reps, &ms)) return 1;
  out->pattern_ms = ms; out->pattern_GBs = total / (ms * 1e6);
  if (time_ms([&] { hipLaunchKernelGGL((k_pattern<0, NX>), dim3(grid), dim3(256), 0, 0, o, (int)nodes); }, reps, &ms)) return 1;
  out->pattern_free_ms = ms; out->pattern_free_GBs = total / (ms * 1e6);
  if (time_ms([&] { hipLaunchKernelGGL((k_pattern<76264, NX, true>), dim3(grid), dim3(256), 0, 0, o, (int)nodes); }, reps, &ms)) return 1;
  out->pattern_nt_GBs = total / (ms * 1e6);
  out->bytes_fill = (double)total; out->bytes_pattern = (double)total;
  for (double* p : {o.A, o.B, o.b, o.Q, o.R, o.q, o.r, o.c, o.C, o.D, o.e, o.perf, o.qrd, o.dump, flat}) CK(hipFree(p));
  CK(hipFree(o.nc));
  return 0;
}

}  // namespace

// out[0..8]: fill16 GB/s, fill8 GB/s, pattern GB/s at the lineariser's occupancy, pattern GB/s at free occupancy, bytes per launch,
// pattern ms, pattern ms (free), fill16 ms, pattern GB/s with non-temporal stores (NX = 22 rows only)
extern "C" int write_roof_measure(int batch, int nodes_per_problem, int nx, double* out) {
  Roof r{};
  if (nx != 22 && nx != 24) return 2;
  if (nx == 22 ? measure<22>(batch, nodes_per_problem, &r) : measure<24>(batch, nodes_per_problem, &r)) return 1;
  out[0] = r.fill16_GBs; out[1] = r.fill8_GBs; out[2] = r.pattern_GBs; out[3] = r.pattern_free_GBs; out[4] = r.bytes_pattern;
  out[5] = r.pattern_ms; out[6] = r.pattern_free_ms; out[7] = r.fill16_ms; out[8] = r.pattern_nt_GBs;
  return 0;
}

#ifndef WRITE_ROOF_LIB
int main(int argc, char** argv) {
  const int batch = argc > 1 ? std::atoi(argv[1]) : 256, npp = argc > 2 ? std::atoi(argv[2]) : 103, nx = argc > 3 ? std::atoi(argv[3]) : 22;
  double o[9];
  if (write_roof_measure(batch, npp, nx, o)) return 1;
  std::printf("{\"batch\": %d, \"nodes_per_problem\": %d, \"bytes_per_launch\": %.0f, \"fill16_GBs\": %.1f, \"fill8_GBs\": %.1f, "
              "\"pattern_GBs\": %.1f, \"pattern_ms\": %.4f, \"pattern_free_occupancy_GBs\": %.1f, \"pattern_free_occupancy_ms\": %.4f, \"fill16_ms\": %.4f, \"pattern_nontemporal_GBs\": %.1f}\n",
              batch, npp, o[4], o[0], o[1], o[2], o[5], o[3], o[6], o[7], o[8]);
  return 0;
}
#endif
