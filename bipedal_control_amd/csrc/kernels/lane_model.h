// Execution-model shim for the kernel bodies.
//
// Every kernel body in this directory is written as a sequence of PHASES.  Inside a phase each lane works on its
// own items; all data that crosses lanes lives in the workgroup's LDS workspace and phases are separated by
// BP_SYNC().  No per-lane value is carried across a BP_SYNC() in registers.  With that discipline the same source
// has two compilations:
//   * HIP / gfx950 (the product): BP_LANES binds `tid` to threadIdx.x, BP_SYNC is a workgroup barrier (free for
//     the one-wavefront workgroups used by the per-node kernels);
//   * BPMPC_HOST_EMULATION (tests/hostemu only): BP_LANES is a loop over the lanes and BP_SYNC is empty, so g++
//     runs the identical arithmetic on the CPU.  This exists so the kernel logic can be checked against the oracle
//     in CPU-only CI; it is never linked into libbpmpc.so and is not a fallback.
#pragma once

#if defined(BPMPC_HOST_EMULATION)
#include <cmath>
#define BP_DEVICE inline
#define BP_LANES(tid, nthreads) for (int tid = 0; tid < (nthreads); ++tid)
#define BP_SYNC() ((void)0)
#define BP_RESTRICT
#else
#include <hip/hip_runtime.h>
#define BP_DEVICE __device__ __forceinline__
#define BP_LANES(tid, nthreads) for (int tid = threadIdx.x, bp_once_ = 1; bp_once_; bp_once_ = 0)
#define BP_SYNC() __syncthreads()
#define BP_RESTRICT __restrict__
#endif

namespace bpmpc {

constexpr int kWave = 64;

BP_DEVICE void cross3(const double* a, const double* b, double* c) {
  const double c0 = a[1] * b[2] - a[2] * b[1];
  const double c1 = a[2] * b[0] - a[0] * b[2];
  const double c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
// y = M v for a symmetric 3x3 stored as xx,xy,xz,yy,yz,zz
BP_DEVICE void sym3_mul(const double* s, const double* v, double* y) {
  const double y0 = s[0] * v[0] + s[1] * v[1] + s[2] * v[2];
  const double y1 = s[1] * v[0] + s[3] * v[1] + s[4] * v[2];
  const double y2 = s[2] * v[0] + s[4] * v[1] + s[5] * v[2];
  y[0] = y0; y[1] = y1; y[2] = y2;
}
BP_DEVICE void mat3_mul(const double* A, const double* B, double* C) {  // C = A B, no aliasing
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
BP_DEVICE void mat3_vec(const double* A, const double* v, double* y) {
  const double y0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  const double y1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  const double y2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  y[0] = y0; y[1] = y1; y[2] = y2;
}
// mode id -> contact flag of contact point c (0,1 left foot; 2,3 right foot)
BP_DEVICE bool stance_flag(int mode, int c) { return c < 2 ? (mode == 1 || mode == 3) : (mode == 2 || mode == 3); }

}  // namespace bpmpc
