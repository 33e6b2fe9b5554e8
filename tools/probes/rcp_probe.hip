// Accuracy of v_rcp_f64 and of one / two Newton refinements against the correctly rounded reciprocal (diagnostic).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double* x, double* e, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(fma(-v, r0, 1.0), r0, r0);
  double r2 = fma(fma(-v, r1, 1.0), r1, r1);
  double ex = 1.0 / v;
  e[3 * i] = fabs(r0 - ex) / fabs(ex); e[3 * i + 1] = fabs(r1 - ex) / fabs(ex); e[3 * i + 2] = fabs(r2 - ex) / fabs(ex);
}
int main() {
  const int n = 1 << 20;
  double *x, *e;
  hipMallocManaged(&x, n * sizeof(double)); hipMallocManaged(&e, 3 * n * sizeof(double));
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (double)(s >> 11) / 9007199254740992.0; x[i] = std::exp((u - 0.5) * 40.0) * ((s & 1) ? 1 : -1); }
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, e, n);
  hipDeviceSynchronize();
  double m[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) for (int q = 0; q < 3; ++q) m[q] = std::fmax(m[q], e[3 * i + q]);
  printf("max relative error: v_rcp_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e\n", m[0], m[1], m[2]);
  return 0;
}
