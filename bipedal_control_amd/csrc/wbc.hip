// Host side of the batched whole-body controller (kernels/wbc.h): settings ingest, device buffers, the C ABI (include/bpmpc.h).
//   WeightedWbc construction + loadTasksSetting     bipedal_controllers/src/BipedalController.cpp:97-100, bipedal_wbc/src/WbcBase.cpp:405-447,
//                                                   bipedal_wbc/src/WeightedWbc.cpp:100-116
//   WeightedWbc::update                             bipedal_controllers/src/BipedalController.cpp:229
#include <hip/hip_runtime.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bpmpc.h"
#include "capi_internal.h"
#include "device_model.h"
#include "info_tree.h"
#include "kernels/wbc.h"

namespace bpmpc {

template <int NJ>
__global__ __launch_bounds__(kWave) void k_wbc(const DeviceModel* model, WbcSettings st, WbcArgs a) {
  __shared__ WbcLds<NJ> w;
  const int b = blockIdx.x;
  if (b >= a.batch) return;
  wbc_robot<NJ>(*model, st, w, a, b, threadIdx.x);
}

}  // namespace bpmpc

using namespace bpmpc;

struct bpmpc_wbc {
  RobotModel rm;
  DeviceModel dm;
  DeviceModel* d_model = nullptr;
  WbcSettings st{};
  int device = 0, max_batch = 0, nv = 0, n = 0;
  hipStream_t stream = nullptr;
  double *d_x = nullptr, *d_u = nullptr, *d_rbd = nullptr, *d_sol = nullptr, *d_debug = nullptr;
  int *d_mode = nullptr, *d_status = nullptr;
};

namespace {
struct WbcError : std::runtime_error { using std::runtime_error::runtime_error; };
void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw WbcError(std::string(what) + ": " + hipGetErrorString(e));
}
#define WBC_HIP(expr) hip_check((expr), #expr)

WbcSettings load_wbc_settings(const std::string& task_info, int nj) {
  const auto t = read_info_file(task_info);
  WbcSettings s{};
  const std::vector<double> lim = load_matrix(*t, "torqueLimitsTask", nj / 2, 1);
  for (int i = 0; i < nj / 2; ++i) s.torque_limits[i] = lim[i];
  auto need = [&](const char* key, double* out) { if (!t->get(key, out)) throw std::runtime_error(std::string("task.info: missing ") + key); };
  need("frictionConeTask.frictionCoefficient", &s.friction);
  need("swingLegTask.kp", &s.swing_kp);
  need("swingLegTask.kd", &s.swing_kd);
  const std::vector<double> kp = load_matrix(*t, "baseAccelPDTask.baseKp", 6, 1), kd = load_matrix(*t, "baseAccelPDTask.baseKd", 6, 1);
  for (int i = 0; i < 6; ++i) { s.base_kp[i] = kp[i]; s.base_kd[i] = kd[i]; }
  // [OCS2-upstream] loadData::loadPtreeValue keeps the member's initial value when the key is absent (it only warns): the Hunter
  // configuration has no noContactMotionTask section, and WbcBase.h:131 initialises noContactMotionTolerance_{} = 0
  s.contact_tolerance = 0.0;
  (void)t->get("noContactMotionTask.tolerance", &s.contact_tolerance);
  need("weight.swingLeg", &s.w_swing);
  need("weight.baseAccel", &s.w_base);
  need("weight.contactForce", &s.w_force);
  s.max_working_set_changes = 20;      // int nWsr = 20, WeightedWbc.cpp:57
  return s;
}

int translate(const std::exception& e) {
  set_last_error(e.what());
  if (dynamic_cast<const WbcError*>(&e)) return BPMPC_ERR_DEVICE;
  if (dynamic_cast<const std::invalid_argument*>(&e)) return BPMPC_ERR_INVALID_ARGUMENT;
  if (dynamic_cast<const std::length_error*>(&e)) return BPMPC_ERR_CAPACITY;
  return BPMPC_ERR_IO;
}
}  // namespace

extern "C" {

int bpmpc_wbc_create(const bpmpc_model* model, const char* task_info_path, int device, int max_batch, bpmpc_wbc** out) {
  if (!model || !task_info_path || !out || max_batch < 1) { set_last_error("bpmpc_wbc_create: bad argument"); return BPMPC_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) {
    set_last_error("bpmpc_wbc_create: no usable HIP device (this engine has no CPU path)");
    return BPMPC_ERR_NO_DEVICE;
  }
  std::unique_ptr<bpmpc_wbc> w(new bpmpc_wbc);
  try {
    w->rm = model_of(model);
    if (w->rm.nj != 10 && w->rm.nj != 12) { set_last_error("only 10- and 12-joint bipeds are instantiated"); return BPMPC_ERR_UNSUPPORTED; }
    w->dm = make_device_model(w->rm);
    w->st = load_wbc_settings(task_info_path, w->rm.nj);
    w->device = device; w->max_batch = max_batch; w->nv = 6 + w->rm.nj; w->n = w->nv + 12 + w->rm.nj;
    WBC_HIP(hipSetDevice(device));
    WBC_HIP(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    WBC_HIP(hipMalloc(reinterpret_cast<void**>(&w->d_model), sizeof(DeviceModel)));
    WBC_HIP(hipMemcpy(w->d_model, &w->dm, sizeof(DeviceModel), hipMemcpyHostToDevice));
    const size_t B = max_batch;
    WBC_HIP(hipMalloc(reinterpret_cast<void**>(&w->d_x), B * w->rm.nx * sizeof(double)));
    WBC_HIP(hipMalloc(reinterpret_cast<void**>(&w->d_u), B * w->rm.nu * sizeof(double)));
    WBC_HIP(hipMalloc(reinterpret_cast<void**>(&w->d_rbd), B * 2 * w->nv * sizeof(double)));
    WBC_HIP(hipMalloc(reinterpret_cast<void**>(&w->d_sol), B * w->n * sizeof(double)));
    WBC_HIP(hipMalloc(reinterpret_cast<void**>(&w->d_debug), B * kWbcDebugStride * sizeof(double)));
    WBC_HIP(hipMalloc(reinterpret_cast<void**>(&w->d_mode), B * sizeof(int)));
    WBC_HIP(hipMalloc(reinterpret_cast<void**>(&w->d_status), B * sizeof(int)));
    WBC_HIP(hipMemset(w->d_sol, 0, B * w->n * sizeof(double)));      // lastQpSol_ starts at zero (WeightedWbc.h)
  } catch (const std::exception& e) {
    const int rc = translate(e);
    bpmpc_wbc_destroy(w.release());
    return rc;
  }
  *out = w.release();
  return BPMPC_OK;
}

void bpmpc_wbc_destroy(bpmpc_wbc* w) {
  if (!w) return;
  if (w->stream) { (void)hipStreamSynchronize(w->stream); (void)hipStreamDestroy(w->stream); }
  for (void* p : {(void*)w->d_model, (void*)w->d_x, (void*)w->d_u, (void*)w->d_rbd, (void*)w->d_sol, (void*)w->d_debug, (void*)w->d_mode, (void*)w->d_status})
    if (p) (void)hipFree(p);
  delete w;
}

int bpmpc_wbc_dims(const bpmpc_wbc* w, int* n_decision, int* nv) {
  if (!w) { set_last_error("null wbc handle"); return BPMPC_ERR_INVALID_ARGUMENT; }
  if (n_decision) *n_decision = w->n;
  if (nv) *nv = w->nv;
  return BPMPC_OK;
}

int bpmpc_wbc_update(bpmpc_wbc* w, int batch, const double* state_desired, const double* input_desired, const double* rbd_state_measured,
                     const int* mode, double period, double* solution, int* status, double* debug) {
  (void)period;      // the joint-acceleration feed-forward that used it is commented out in the reference (WbcBase.cpp:242-243)
  if (!w || !state_desired || !input_desired || !rbd_state_measured || !mode || !solution) { set_last_error("bpmpc_wbc_update: null argument"); return BPMPC_ERR_INVALID_ARGUMENT; }
  try {
    if (batch < 1 || batch > w->max_batch) throw std::length_error("bpmpc_wbc_update: batch exceeds max_batch");
    for (int b = 0; b < batch; ++b) if (mode[b] < 0 || mode[b] > 3) throw std::invalid_argument("bpmpc_wbc_update: mode must be 0..3");
    WBC_HIP(hipSetDevice(w->device));
    const size_t B = batch;
    WBC_HIP(hipMemcpyAsync(w->d_x, state_desired, B * w->rm.nx * sizeof(double), hipMemcpyHostToDevice, w->stream));
    WBC_HIP(hipMemcpyAsync(w->d_u, input_desired, B * w->rm.nu * sizeof(double), hipMemcpyHostToDevice, w->stream));
    WBC_HIP(hipMemcpyAsync(w->d_rbd, rbd_state_measured, B * 2 * w->nv * sizeof(double), hipMemcpyHostToDevice, w->stream));
    WBC_HIP(hipMemcpyAsync(w->d_mode, mode, B * sizeof(int), hipMemcpyHostToDevice, w->stream));
    WbcArgs a{};
    a.batch = batch; a.nx = w->rm.nx; a.state_des = w->d_x; a.input_des = w->d_u; a.rbd_meas = w->d_rbd; a.mode = w->d_mode;
    a.sol = w->d_sol; a.status = w->d_status; a.debug = debug ? w->d_debug : nullptr;
    if (w->rm.nj == 10) hipLaunchKernelGGL(k_wbc<10>, dim3(batch), dim3(kWave), 0, w->stream, w->d_model, w->st, a);
    else hipLaunchKernelGGL(k_wbc<12>, dim3(batch), dim3(kWave), 0, w->stream, w->d_model, w->st, a);
    WBC_HIP(hipGetLastError());
    WBC_HIP(hipMemcpyAsync(solution, w->d_sol, B * w->n * sizeof(double), hipMemcpyDeviceToHost, w->stream));
    if (status) WBC_HIP(hipMemcpyAsync(status, w->d_status, B * sizeof(int), hipMemcpyDeviceToHost, w->stream));
    if (debug) WBC_HIP(hipMemcpyAsync(debug, w->d_debug, B * kWbcDebugStride * sizeof(double), hipMemcpyDeviceToHost, w->stream));
    WBC_HIP(hipStreamSynchronize(w->stream));
  } catch (const std::exception& e) { return translate(e); }
  return BPMPC_OK;
}

int bpmpc_wbc_reset(bpmpc_wbc* w) {
  if (!w) { set_last_error("null wbc handle"); return BPMPC_ERR_INVALID_ARGUMENT; }
  try {
    WBC_HIP(hipSetDevice(w->device));
    WBC_HIP(hipMemsetAsync(w->d_sol, 0, (size_t)w->max_batch * w->n * sizeof(double), w->stream));
    WBC_HIP(hipStreamSynchronize(w->stream));
  } catch (const std::exception& e) { return translate(e); }
  return BPMPC_OK;
}

}  // extern "C"
