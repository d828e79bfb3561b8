// Library-level entry points of libvc_hip: version, last-error string.
#include <stdarg.h>

#include <atomic>
#include <mutex>

#include "vc_common.h"

namespace {
thread_local char g_err[512] = "";

constexpr int MAX_DEV = VC_MAX_DEVICES;
// Deterministic-mode state: the flag is atomic and the per-device tables are read and written under g_det_mu, so a
// host thread that switches the mode never hands another thread a half-updated (workspace, size) pair.  A launcher that
// has already fetched its workspace keeps using it: the caller must not free a workspace while calls are in flight.
std::atomic<int> g_det{0};
std::mutex g_det_mu;
float* g_det_ws[MAX_DEV] = {};
size_t g_det_floats[MAX_DEV] = {};
std::atomic<int> g_cus[MAX_DEV];  // compute units per device, 0 = not asked yet

int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    (void)hipGetLastError();
    return -1;
  }
  return dev;
}

// out[i] += the sum over the slots of ws[slot][i], in an order that depends on (n, slots) only: the slots are cut
// into 8 equal runs, a thread adds one run front to back (four loads in flight), and the 8 run sums of an element are
// added first to last.  256 threads = 32 elements x 8 runs.
__global__ __launch_bounds__(256) void det_reduce_kernel(float* __restrict__ out, const float* __restrict__ ws, size_t n,
                                                         int slots) {
  __shared__ float part[8][32];
  const int e = threadIdx.x & 31, run = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + e;
  const int per = (slots + 7) / 8;
  const int s0 = run * per, s1 = min(slots, s0 + per);
  float a = 0.f;
  if (i < n) {
    int s = s0;
    for (; s + 3 < s1; s += 4) {
      const float v0 = ws[(size_t)s * n + i], v1 = ws[(size_t)(s + 1) * n + i], v2 = ws[(size_t)(s + 2) * n + i],
                  v3 = ws[(size_t)(s + 3) * n + i];
      a = ((a + v0) + v1) + v2;
      a += v3;
    }
    for (; s < s1; ++s) a += ws[(size_t)s * n + i];
  }
  part[run][e] = a;
  __syncthreads();
  if (run == 0 && i < n) {
    float t = part[0][e];
#pragma unroll
    for (int r = 1; r < 8; ++r) t += part[r][e];
    out[i] += t;
  }
}
}  // namespace

bool vc_deterministic() { return g_det != 0; }

size_t vc_det_capacity() {
  const int dev = current_device();
  if (dev < 0) return 0;
  std::lock_guard<std::mutex> lock(g_det_mu);
  return g_det_floats[dev];
}

float* vc_det_workspace(size_t floats, hipStream_t stream, const char* who) {
  const int dev = current_device();
  float* ws = nullptr;
  size_t have = 0;
  if (dev >= 0) {  // one consistent (pointer, size) pair
    std::lock_guard<std::mutex> lock(g_det_mu);
    ws = g_det_ws[dev];
    have = g_det_floats[dev];
  }
  if (!ws || have < floats) {
    vc_set_error("%s: deterministic mode needs a workspace of %zu floats on this device, vc_set_deterministic registered %zu",
                 who, floats, have);
    return nullptr;
  }
  if (hipMemsetAsync(ws, 0, floats * sizeof(float), stream) != hipSuccess) {
    vc_set_error("%s: clearing the deterministic-mode workspace failed: %s", who, hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return ws;
}

int vc_det_reduce(float* out, const float* ws, size_t n, int slots, hipStream_t stream) {
  hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, stream, out, ws, n, slots);
  VC_LAUNCH_CHECK("vc_det_reduce");
  return VC_OK;
}

void vc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Compute units of the current device (the bound on how many one-per-CU workgroups can be resident at once);
// 256 (MI355X) when no device can be asked, e.g. on a build host.
int vc_device_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 256;
  }
  const bool cached = dev >= 0 && dev < MAX_DEV;
  if (cached && (n = g_cus[dev].load(std::memory_order_relaxed)) > 0) return n;  // one attribute query per device
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  if (cached) g_cus[dev].store(n, std::memory_order_relaxed);
  return n;
}

extern "C" {

int vc_version(void) { return 101; }

const char* vc_last_error(void) { return g_err; }

// on != 0: every sum that is otherwise combined with floating-point atomics (weight and bias gradients, BatchNorm
// statistics and backward reductions, the loss) is added in a fixed order, so that a step repeats bit for bit.
// `workspace` (device memory of the CURRENT device, `workspace_floats` floats, owned by the caller and kept alive while
// the mode is on) holds the per-block partial sums; call once per device.  Calls that use it must be issued on one
// stream at a time per device.  on == 0 switches the mode off for every device and forgets the workspaces.
int vc_set_deterministic(int on, float* workspace, size_t workspace_floats) {
  std::lock_guard<std::mutex> lock(g_det_mu);
  if (!on) {
    g_det = 0;
    for (int d = 0; d < MAX_DEV; ++d) {
      g_det_ws[d] = nullptr;
      g_det_floats[d] = 0;
    }
    return VC_OK;
  }
  const int dev = current_device();
  VC_CHECK_ARG(dev >= 0, "vc_set_deterministic: no current device");
  VC_CHECK_ARG(workspace && workspace_floats >= (1u << 20), "vc_set_deterministic: needs a workspace of at least 2^20 floats");
  g_det_ws[dev] = workspace;
  g_det_floats[dev] = workspace_floats;
  g_det = 1;
  return VC_OK;
}

int vc_get_deterministic(void) { return g_det; }

}  // extern "C"
