// Host-side entry points of the kernel translation units (k_node.hip, k_project.hip, k_riccati.hip, k_riccati_wave.hip): the solver
// (solver.hip, host code) launches every kernel through these, so the kernel families compile in parallel and a change to one of them
// rebuilds one translation unit.  `nj` selects the instantiation (10: nx = nu = 22, 12: nx = nu = 24); every function only enqueues.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace bpmpc {

struct Launch;
struct DeviceModel;
struct RolloutArgs;
struct DdpBuffers;

namespace kl {

// ---- k_node.hip: per-node kernels in the lane-per-coordinate mapping, line search, warm start, policy rollout
void prepare(int nj, int slots, hipStream_t st, const Launch& L);
void linearize_reference(int nj, int slots, hipStream_t st, const Launch& L);
void linearize_fast(int nj, bool materialise, int nodes, hipStream_t st, const Launch& L, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void warm_shift(int nj, int slots, hipStream_t st, const Launch& L);
void ls_begin(int nj, int batch, hipStream_t st, const Launch& L);
void trial_reference(int nj, int slots, hipStream_t st, const Launch& L);
int trial_fast_workgroups(int nj, int nodes);
void trial_fast(int nj, int nodes, hipStream_t st, const Launch& L);
void ls_decide(int nj, int batch, hipStream_t st, const Launch& L, bool look_first = false);
void ls_tail(int nj, int batch, hipStream_t st, const Launch& L, int first_round, int max_trials, bool pending);
void constraint_values(int nj, int nodes, hipStream_t st, const Launch& L, double* eqv);
void rollout(int nj, bool serial_legs, int batch, hipStream_t st, const DeviceModel* model, const RolloutArgs& a);
void copy_pairs(int grid, hipStream_t st, const double* a_src, double* a_dst, size_t na, const double* b_src, double* b_dst, size_t nb,
                int* iterations, int* active, int batch);

// ---- k_project.hip: constraint elimination and change of variables
void project_reference(int nj, int slots, hipStream_t st, const Launch& L);
void project_lu_s(int nj, int max_vel_rows, bool packed, int nodes, hipStream_t st, const Launch& L);
void project_fast(int nj, bool packed, bool joint_rows, int nodes, hipStream_t st, const Launch& L);

// ---- k_riccati.hip: workgroup-per-problem sweeps
void riccati_reference(int nj, int batch, hipStream_t st, const Launch& L);
void riccati_fast(int nj, bool double_buffered, bool joint_rows, int batch, hipStream_t st, const Launch& L);
void riccati_fast8(int nj, bool joint_rows, int batch, hipStream_t st, const Launch& L);     // joint_rows: Wt holds them (off: completed from Vt by the loaders)

// ---- k_riccati_wave.hip: wave-per-problem sweeps and their roll-out
void riccati_wave(int nj, bool two_per_simd, bool joint_rows, int batch, hipStream_t st, const Launch& L);
void riccati_rollout(int nj, int batch, hipStream_t st, const Launch& L);

// ---- k_ddp.hip: the DDP slice (one ILQR iteration; backward pass on the reference kernel set, line search over policy roll-outs)
void ddp_policy(int nj, int batch, hipStream_t st, const Launch& L, const DdpBuffers& d);
void ddp_cost(int nj, int batch, bool fast, hipStream_t st, const Launch& L, const DdpBuffers& d);
void ddp_select(int nj, int batch, hipStream_t st, const Launch& L, const DdpBuffers& d, double armijo);
void ddp_nominal(int nj, int batch, hipStream_t st, const Launch& L, const DdpBuffers& d);
void ddp_finish(int nj, int batch, hipStream_t st, const Launch& L, const DdpBuffers& d);
void ddp_keep_times(int batch, int N, hipStream_t st, const DdpBuffers& d, double* tp_time, int* tp_kind, int* tp_nodes, int* tp_grid);

}  // namespace kl
}  // namespace bpmpc
