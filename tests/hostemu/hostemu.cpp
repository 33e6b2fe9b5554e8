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
