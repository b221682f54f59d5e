// Error reporting and version of the C ABI (include/vsseg_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#include "../../include/vsseg_hip.h"

static thread_local char g_err[512] = "";

extern "C" void vsseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* vsseg_last_error(void) { return g_err; }
extern "C" int vsseg_version(void) { return 7; }

// Forks without a marker packet (common.h, vsseg_launch_kernel): between vsseg_fork_arm(ev) and vsseg_fork_disarm() every kernel the library launches on the calling
// thread carries `ev` as the stop event of its own dispatch; a later record replaces an earlier one, so after a launch record of several kernels the event stands for the
// last of them.  vsseg_fork_disarm returns how many kernels carried it (0: the record launched none — a memset — and the caller forks with a plain event record).
static thread_local hipEvent_t g_fork_ev = nullptr;
static thread_local int g_fork_n = 0;
hipEvent_t vsseg_fork_event() {
  if (g_fork_ev) ++g_fork_n;
  return g_fork_ev;
}
// (the events order streams of ONE device: no system-scope fence)
extern "C" void* vsseg_fork_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) { vsseg_set_error("vsseg_fork_event_create: hipEventCreateWithFlags failed"); return nullptr; }
  return ev;
}
extern "C" int vsseg_fork_event_destroy(void* ev) { return ev && hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)) == hipSuccess ? VSSEG_OK : VSSEG_EINVAL; }
extern "C" int vsseg_fork_arm(void* ev) {
  if (!ev) { vsseg_set_error("vsseg_fork_arm: null event"); return VSSEG_EINVAL; }
  g_fork_ev = reinterpret_cast<hipEvent_t>(ev);
  g_fork_n = 0;
  return VSSEG_OK;
}
extern "C" int vsseg_fork_disarm(void) {
  const int n = g_fork_ev ? g_fork_n : 0;
  g_fork_ev = nullptr;
  g_fork_n = 0;
  return n;
}
extern "C" int vsseg_stream_wait_event(void* stream, void* ev) {
  if (!ev || hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(ev), 0) != hipSuccess) { vsseg_set_error("vsseg_stream_wait_event: hipStreamWaitEvent failed"); return VSSEG_ELAUNCH; }
  return VSSEG_OK;
}

// Sticky flag of the fixed-point accumulators (csrc/common.h, vsseg_fx_add): one word of device memory PER DEVICE (the current device of the calling thread: one
// process per GPU is the product's layout, a second device in the same process gets its own word), allocated once under a lock.  nullptr only if the allocation
// failed: every launcher checks (VSSEG_FX_FLAG in common.h) instead of handing a null pointer to a kernel.
#include <mutex>
unsigned* vsseg_fx_flag() {
  static std::mutex mu;
  static unsigned* flags[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!flags[dev]) {
    unsigned* f = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&f), 256) == hipSuccess && hipMemset(f, 0, 256) == hipSuccess) flags[dev] = f;
  }
  return flags[dev];
}
extern "C" int vsseg_fx_status(int32_t reset, void* stream) {
  unsigned* f = vsseg_fx_flag();
  if (!f) { vsseg_set_error("vsseg_fx_status: could not allocate the flag word"); return VSSEG_ELAUNCH; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  unsigned h = 0;
  hipError_t e = hipMemcpyAsync(&h, f, sizeof(h), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess && reset && h) e = hipMemsetAsync(f, 0, sizeof(h), s);
  if (e != hipSuccess) { vsseg_set_error("vsseg_fx_status: %s", hipGetErrorString(e)); return VSSEG_ELAUNCH; }
  return (int)(h & 1u);
}

// Zero-fill / device copy on the caller's stream (replace the torch fill / clone kernels the step used to issue: per-step
// statistics, the flat gradient buffer, gradient slices that receive their first contribution as a partial write).
extern "C" int vsseg_memset_zero(void* dst, int64_t bytes, void* stream) {
  if (!dst || bytes < 0) { vsseg_set_error("vsseg_memset_zero: bad arguments"); return VSSEG_EINVAL; }
  if (bytes == 0) return VSSEG_OK;
  hipError_t e = hipMemsetAsync(dst, 0, (size_t)bytes, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { vsseg_set_error("vsseg_memset_zero: %s", hipGetErrorString(e)); return VSSEG_ELAUNCH; }
  return VSSEG_OK;
}
extern "C" int vsseg_copy_bytes(const void* src, void* dst, int64_t bytes, void* stream) {
  if (!src || !dst || bytes < 0) { vsseg_set_error("vsseg_copy_bytes: bad arguments"); return VSSEG_EINVAL; }
  if (bytes == 0) return VSSEG_OK;
  hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { vsseg_set_error("vsseg_copy_bytes: %s", hipGetErrorString(e)); return VSSEG_ELAUNCH; }
  return VSSEG_OK;
}
