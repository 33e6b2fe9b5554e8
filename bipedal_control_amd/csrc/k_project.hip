// Constraint elimination (kernels/project_lu_s.h) and change of variables (kernels/project_mfma.h); reference body kernels/project_node.h.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernel_launchers.h"
#include "launch.h"
#include "kernels/project_lu_s.h"
#include "kernels/project_mfma.h"

namespace bpmpc {

template <int NJ>
__global__ __launch_bounds__(kWave) void k_project(Launch L) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  __shared__ ProjectWorkspace<NJ> ws;
  const int sidx = blockIdx.x, b = sidx / L.N, k = sidx % L.N;
  if (!L.buf.active[b]) return;
  const int g = L.buf.p_grid[b];
  if (k >= L.buf.g_nodes[g]) return;
  const size_t s = sidx;
  ProjectIn in;
  in.kind = L.buf.g_kind[(size_t)g * L.N + k];
  in.nc = L.buf.nc[s];
  in.C = L.buf.C + s * kMaxEqRows * NX; in.D = L.buf.D + s * kMaxEqRows * NU; in.e = L.buf.e + s * kMaxEqRows;
  in.A = L.buf.A + s * NX * NX; in.B = L.buf.B + s * NX * NU; in.b = L.buf.b + s * NX;
  in.Q = L.buf.Q + s * NX * NX; in.R = L.buf.R + s * NU * NU; in.P = L.buf.P + s * NU * NX; in.q = L.buf.q + s * NX; in.r = L.buf.r + s * NU;
  ProjectOut out;
  out.Px = L.buf.Px + s * NU * NX; out.Pu = L.buf.Pu + s * NU * NU; out.Pe = L.buf.Pe + s * NU; out.nut = L.buf.nut + s;
  out.At = L.buf.At + s * NX * NX; out.Bt = L.buf.Bt + s * NX * NU; out.bt = L.buf.bt + s * NX;
  out.Qt = L.buf.Qt + s * NX * NX; out.Rt = L.buf.Rt + s * NU * NU; out.Pt = L.buf.Pt + s * NU * NX; out.qt = L.buf.qt + s * NX;
  out.rt = L.buf.rt + s * NU;
  project_node<NJ>(ws, in, out);
}

// 68 registers; 9 KB of LDS per wave: four waves per SIMD (0.107 -> 0.095 ms), 5.9 KB with the compact tile of the packed outputs: six
template <int NJ, int RM, bool PK>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(4, 6))) void k_project_lu_s(Launch L) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ, WP = PackedLq<NJ>::WP;
  __shared__ ProjectLuSLds<NJ, PK> lds[kLuNodes];
  const int sub = threadIdx.x / kLuLanes, j = threadIdx.x % kLuLanes;
  const int widx = blockIdx.x * kLuNodes + sub;
  bool valid = widx < L.batch * L.klen;
  const int b = valid ? widx / L.klen : 0, k = valid ? L.k0 + widx % L.klen : 0;
  // the node's facts one memory round trip away (n_info, written by k_prepare): a wave of this kernel lives ~40 k cycles and every
  // dependent load before its data loads costs it ~5 k
  const int act = L.buf.active[b];
  const int info = L.buf.n_info[(size_t)b * L.N + k];
  const int kind = info & 1, gmode = (info >> 1) & 3;
  valid = valid && act != 0 && info != 0;
  const size_t s = valid ? (size_t)b * L.N + k : 0;
  double* Px = L.buf.Px + s * NU * NX;
  double* Pu = L.buf.Pu + s * NU * NU;
  double* Pe = L.buf.Pe + s * NU;
  double* Vt = L.buf.Vt + s * NJ * WP;
  if (valid && kind == 1) {   // event node: no input
    if constexpr (PK) {
      for (int idx = j; idx < NJ * 32; idx += kLuLanes) Vt[(idx >> 5) * WP + (idx & 31)] = 0.0;      // the first two block columns: what a reader with nut = 0 loads
    } else {
      for (int idx = j; idx < NU * NX; idx += kLuLanes) { Px[idx] = 0.0; Pu[idx] = 0.0; }
    }
    for (int idx = j; idx < NU; idx += kLuLanes) Pe[idx] = 0.0;
    if (j == 0) L.buf.nut[s] = 0;
    valid = false;
  }
  const int mode = valid ? (gmode & 3) : 3;
  project_lu_s<NJ, RM, PK>(lds[sub], valid, mode, L.buf.D + s * kMaxEqRows * NU, L.buf.C + s * kMaxEqRows * NX, L.buf.e + s * kMaxEqRows, Px, Pu, Pe,
                           L.buf.nut + s, sub, j, Vt,
                           (threadIdx.x == 0 && blockIdx.x < (unsigned)L.batch) ? L.buf.rprof + 8 * blockIdx.x : nullptr);
}

// (amdgpu_waves_per_eu(3): 168 registers and 60 B of scratch instead of 186 registers, three waves per SIMD instead of two - measured
//  slower, 0.272 against 0.255 ms on the same box)
template <int NJ, bool PK, bool WJ>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(PK ? 3 : 2, 4))) void k_project_fast(Launch L) {
  constexpr int NX = 12 + NJ, NU = 12 + NJ;
  __shared__ ProjectMfmaWorkspace<NJ, PK> ws;
  const int b = blockIdx.x / L.klen, k = L.k0 + blockIdx.x % L.klen;
  if (!L.buf.active[b]) return;
  const int g = L.buf.p_grid[b];
  if (k >= L.buf.g_nodes[g]) return;
  const size_t s = (size_t)b * L.N + k;
  ProjectIn in;
  in.kind = L.buf.g_kind[(size_t)g * L.N + k];
  in.nc = L.buf.nc[s];
  in.C = L.buf.C + s * kMaxEqRows * NX; in.D = L.buf.D + s * kMaxEqRows * NU; in.e = L.buf.e + s * kMaxEqRows;
  in.A = L.buf.A + s * NX * NX; in.B = L.buf.B + s * NX * NU; in.b = L.buf.b + s * NX;
  in.Q = L.buf.Q + s * NX * NX; in.R = L.buf.R + s * NU * NU; in.P = L.buf.P + s * NU * NX; in.q = L.buf.q + s * NX; in.r = L.buf.r + s * NU;
  ProjectOut out;
  out.Px = L.buf.Px + s * NU * NX; out.Pu = L.buf.Pu + s * NU * NU; out.Pe = L.buf.Pe + s * NU; out.nut = L.buf.nut + s;
  out.At = L.buf.At + s * NX * NX; out.Bt = L.buf.Bt + s * NX * NU; out.bt = L.buf.bt + s * NX;
  out.Qt = L.buf.Qt + s * NX * NX; out.Rt = L.buf.Rt + s * NU * NU; out.Pt = L.buf.Pt + s * NU * NX; out.qt = L.buf.qt + s * NX;
  out.rt = L.buf.rt + s * NU;
  in.qrd = L.buf.qrd + s * kQrdStride;
  in.r_shift_at = L.ilqr ? kQrdRShift : 0;
  const double dt = L.buf.g_dt[(size_t)g * L.N + k];
  out.Wt = L.buf.Wt + s * PackedLq<NJ>::W_SIZE; out.Qp = L.buf.Qp + s * PackedLq<NJ>::Q_SIZE; out.Mt = L.buf.Mt + s * PackedLq<NJ>::M_SIZE;
  if constexpr (PK) { in.zero = L.buf.zero_page; in.Vt = L.buf.Vt + s * NJ * PackedLq<NJ>::WP; in.mode = L.buf.g_mode[(size_t)g * L.N + k] & 3; }   // written by the structured elimination
  else out.Vt = L.buf.Vt + s * NJ * PackedLq<NJ>::WP;   // FullPivLU elimination (Px, Pu, Pe): this kernel packs the joint rows for the sweep's loaders
  project_apply_mfma<NJ, PK, WJ>(ws, in, out, dt, dt * (1.0 / L.model->robot_mass), L.model->Q, L.model->R, L.reg_prim);   // as written by linearize_fast
}

#define KL_NJ(nj, ...)                                                          \
  do {                                                                          \
    if ((nj) == 10) { constexpr int NJ = 10; __VA_ARGS__; }                     \
    else if ((nj) == 12) { constexpr int NJ = 12; __VA_ARGS__; }                \
    else throw std::runtime_error("unsupported joint count");                   \
  } while (0)

namespace kl {

void project_reference(int nj, int slots, hipStream_t st, const Launch& L) { KL_NJ(nj, hipLaunchKernelGGL(k_project<NJ>, dim3(slots), dim3(kWave), 0, st, L)); }
// max_vel_rows: largest number of rows that constrain a contact velocity over the nodes of the setup (instantiation 8 or 12);
// packed: only the joint rows Vt leave the kernel (the structured change of variables / the sweeps generate the force rows)
void project_lu_s(int nj, int max_vel_rows, bool packed, int nodes, hipStream_t st, const Launch& L) {
  const int grid = (nodes + kLuNodes - 1) / kLuNodes;
  KL_NJ(nj, {
    if (packed) {
      if (max_vel_rows <= 8) hipLaunchKernelGGL((k_project_lu_s<NJ, 8, true>), dim3(grid), dim3(kWave), 0, st, L);
      else hipLaunchKernelGGL((k_project_lu_s<NJ, 12, true>), dim3(grid), dim3(kWave), 0, st, L);
    } else {
      if (max_vel_rows <= 8) hipLaunchKernelGGL((k_project_lu_s<NJ, 8, false>), dim3(grid), dim3(kWave), 0, st, L);
      else hipLaunchKernelGGL((k_project_lu_s<NJ, 12, false>), dim3(grid), dim3(kWave), 0, st, L);
    }
  });
}
// joint_rows false: the joint rows of Wt are left to the sweep (packed operands and a sweep that completes them only: riccati_wave2.h)
void project_fast(int nj, bool packed, bool joint_rows, int nodes, hipStream_t st, const Launch& L) {
  KL_NJ(nj, {
    if (packed && !joint_rows) hipLaunchKernelGGL((k_project_fast<NJ, true, false>), dim3(nodes), dim3(kWave), 0, st, L);
    else if (packed) hipLaunchKernelGGL((k_project_fast<NJ, true, true>), dim3(nodes), dim3(kWave), 0, st, L);
    else hipLaunchKernelGGL((k_project_fast<NJ, false, true>), dim3(nodes), dim3(kWave), 0, st, L);
  });
}

}  // namespace kl
}  // namespace bpmpc
