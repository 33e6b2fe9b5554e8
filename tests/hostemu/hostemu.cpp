// TEST-ONLY: CPU lane-emulation build of the HIP kernel bodies (see bipedal_control_amd/csrc/kernels/lane_model.h).
// Compiled by tests/conftest.py with g++ -DBPMPC_HOST_EMULATION into tests/hostemu/libbpmpc_hostemu.so so that the
// CPU-only test tier can check the kernel arithmetic against the oracle.  Never part of libbpmpc.so, not a fallback.
#include <cstring>
#include <memory>
#include <string>

#include "../../bipedal_control_amd/csrc/device_model.h"
#include "../../bipedal_control_amd/csrc/kernels/node_lq.h"
#include "../../bipedal_control_amd/csrc/robot_model.h"

using namespace bpmpc;

struct EmuModel { RobotModel rm; DeviceModel dm; };

template <int NJ>
static void run_linearize(const DeviceModel& dm, const NodeInputs& in, const NodeLQOut& out) {
  auto ws = std::make_unique<NodeWorkspace<NJ>>();
  std::memset(ws.get(), 0, sizeof(*ws));
  linearize_node<NJ>(dm, *ws, in, out);
}
template <int NJ>
static void run_perf(const DeviceModel& dm, const NodeInputs& in, double* perf) {
  auto ws = std::make_unique<NodeWorkspace<NJ>>();
  std::memset(ws.get(), 0, sizeof(*ws));
  node_performance<NJ>(dm, *ws, in, perf);
}

extern "C" {

void* emu_model_create(const char* urdf, const char* task, const char* ref) {
  try {
    auto m = std::make_unique<EmuModel>();
    m->rm = load_robot_model(urdf, task, ref);
    m->dm = make_device_model(m->rm);
    return m.release();
  } catch (const std::exception&) {
    return nullptr;
  }
}
void emu_model_destroy(void* m) { delete static_cast<EmuModel*>(m); }

int emu_linearize_node(void* mv, int kind, int mode, double dt, const double* x, const double* u, const double* xnext, const double* xref,
                       const double* zref, const double* zdref, double* A, double* B, double* b, double* Q, double* R, double* P, double* q,
                       double* r, double* c, double* C, double* D, double* e, int* nc, double* perf) {
  EmuModel* m = static_cast<EmuModel*>(mv);
  NodeInputs in{kind, mode, dt, x, u, xnext, xref, zref, zdref};
  NodeLQOut out{A, B, b, Q, R, P, q, r, c, C, D, e, nc, perf};
  if (m->dm.nj == 10) run_linearize<10>(m->dm, in, out);
  else if (m->dm.nj == 12) run_linearize<12>(m->dm, in, out);
  else return -1;
  return 0;
}

int emu_node_performance(void* mv, int kind, int mode, double dt, const double* x, const double* u, const double* xnext, const double* xref,
                         const double* zref, const double* zdref, double* perf) {
  EmuModel* m = static_cast<EmuModel*>(mv);
  NodeInputs in{kind, mode, dt, x, u, xnext, xref, zref, zdref};
  if (m->dm.nj == 10) run_perf<10>(m->dm, in, perf);
  else if (m->dm.nj == 12) run_perf<12>(m->dm, in, perf);
  else return -1;
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
#include <vector>

#include "../../bipedal_control_amd/csrc/kernels/project_node.h"
#include "../../bipedal_control_amd/csrc/kernels/riccati.h"

template <int NJ>
static void run_project(const ProjectIn& in, const ProjectOut& out) {
  auto ws = std::make_unique<ProjectWorkspace<NJ>>();
  std::memset(ws.get(), 0, sizeof(*ws));
  project_node<NJ>(*ws, in, out);
}
template <int NJ>
static void run_riccati(const RiccatiIO& io) {
  auto ws = std::make_unique<RiccatiWorkspace<NJ>>();
  std::memset(ws.get(), 0, sizeof(*ws));
  riccati_problem<NJ>(*ws, io);
}

extern "C" {

int emu_project_node(void* mv, int kind, int nc, const double* C, const double* D, const double* e, const double* A, const double* B,
                     const double* b, const double* Q, const double* R, const double* P, const double* q, const double* r, double* Px,
                     double* Pu, double* Pe, int* nut, double* At, double* Bt, double* bt, double* Qt, double* Rt, double* Pt, double* qt,
                     double* rt) {
  EmuModel* m = static_cast<EmuModel*>(mv);
  ProjectIn in{kind, nc, C, D, e, A, B, b, Q, R, P, q, r};
  ProjectOut out{Px, Pu, Pe, nut, At, Bt, bt, Qt, Rt, Pt, qt, rt};
  if (m->dm.nj == 10) run_project<10>(in, out);
  else if (m->dm.nj == 12) run_project<12>(in, out);
  else return -1;
  return 0;
}

// Full QP step of one problem through the emulated kernels: linearize -> project -> riccati.
int emu_qp_step(void* mv, int N, const int* kind, const double* dt, const int* mode, const double* zref, const double* zdref,
                const double* xref, const double* x0, const double* x, const double* u, double* dx, double* du, double* K, double* summary,
                double* perf_sum) {
  EmuModel* m = static_cast<EmuModel*>(mv);
  const int nx = m->dm.nj + 12, nu = nx;
  std::vector<double> A(N * nx * nx), B(N * nx * nu), b(N * nx), Q(N * nx * nx), R(N * nu * nu), P(N * nu * nx), q(N * nx), r(N * nu), c(N),
      C(N * 16 * nx), D(N * 16 * nu), e(N * 16), perf(N * 3);
  std::vector<int> nc(N), nut(N);
  std::vector<double> Px(N * nu * nx), Pu(N * nu * nu), Pe(N * nu), At(N * nx * nx), Bt(N * nx * nu), bt(N * nx), Qt(N * nx * nx),
      Rt(N * nu * nu), Pt(N * nu * nx), qt(N * nx), rt(N * nu), Kt(N * nu * nx), kt(N * nu), dx0(nx);
  perf_sum[0] = perf_sum[1] = perf_sum[2] = 0;
  for (int k = 0; k < N; ++k) {
    NodeInputs in{kind[k], mode[k], dt[k], x + k * nx, u + k * nu, x + (k + 1) * nx, xref + k * nx, zref + 4 * k, zdref + 4 * k};
    NodeLQOut out{&A[k * nx * nx], &B[k * nx * nu], &b[k * nx], &Q[k * nx * nx], &R[k * nu * nu], &P[k * nu * nx], &q[k * nx], &r[k * nu],
                  &c[k], &C[k * 16 * nx], &D[k * 16 * nu], &e[k * 16], &nc[k], &perf[3 * k]};
    ProjectIn pin{kind[k], 0, out.C, out.D, out.e, out.A, out.B, out.b, out.Q, out.R, out.P, out.q, out.r};
    ProjectOut pout{&Px[k * nu * nx], &Pu[k * nu * nu], &Pe[k * nu], &nut[k], &At[k * nx * nx], &Bt[k * nx * nu], &bt[k * nx],
                    &Qt[k * nx * nx], &Rt[k * nu * nu], &Pt[k * nu * nx], &qt[k * nx], &rt[k * nu]};
    if (m->dm.nj == 10) { run_linearize<10>(m->dm, in, out); pin.nc = nc[k]; run_project<10>(pin, pout); }
    else { run_linearize<12>(m->dm, in, out); pin.nc = nc[k]; run_project<12>(pin, pout); }
    for (int i = 0; i < 3; ++i) perf_sum[i] += perf[3 * k + i];
  }
  for (int i = 0; i < nx; ++i) dx0[i] = x0[i] - x[i];
  RiccatiIO io{N, nut.data(), At.data(), Bt.data(), bt.data(), Qt.data(), Rt.data(), Pt.data(), qt.data(), rt.data(), Px.data(), Pu.data(),
               Pe.data(), dx0.data(), Kt.data(), kt.data(), dx, du, K, summary};
  if (m->dm.nj == 10) run_riccati<10>(io); else run_riccati<12>(io);
  return 0;
}

// One SQP iteration of one problem through the emulated kernel bodies, as far as a timing baseline needs it (bench.py
// `cpu_baseline_analytic`): QP step (linearise with ANALYTIC derivatives -> project -> Riccati), then the full-step trial
// x + dx, u + du evaluated on every node (what the filter line search looks at first; every problem of the benchmark accepts it).
// perf_before / perf_after: {cost, dynamics SSE, equality SSE} summed over the nodes.
int emu_solve_iteration(void* mv, int N, const int* kind, const double* dt, const int* mode, const double* zref, const double* zdref,
                        const double* xref, const double* x0, const double* x, const double* u, double* x_new, double* u_new, double* K,
                        double* perf_before, double* perf_after) {
  EmuModel* m = static_cast<EmuModel*>(mv);
  const int nx = m->dm.nj + 12, nu = nx;
  std::vector<double> dx((N + 1) * nx), du(N * nu);
  double summary[4];
  if (emu_qp_step(mv, N, kind, dt, mode, zref, zdref, xref, x0, x, u, dx.data(), du.data(), K, summary, perf_before) != 0) return -1;
  if (summary[3] != 0.0) return 2;
  for (int i = 0; i < (N + 1) * nx; ++i) x_new[i] = x[i] + dx[i];
  for (int i = 0; i < N * nu; ++i) u_new[i] = u[i] + du[i];
  perf_after[0] = perf_after[1] = perf_after[2] = 0.0;
  for (int k = 0; k < N; ++k) {
    double perf[3];
    NodeInputs in{kind[k], mode[k], dt[k], x_new + k * nx, u_new + k * nu, x_new + (k + 1) * nx, xref + k * nx, zref + 4 * k, zdref + 4 * k};
    if (m->dm.nj == 10) run_perf<10>(m->dm, in, perf); else run_perf<12>(m->dm, in, perf);
    for (int i = 0; i < 3; ++i) perf_after[i] += perf[i];
  }
  return 0;
}

}  // extern "C"
