// Shared helpers for the libpfk.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PFK_WAVE 64

static inline bool pfk_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipFuncSetAttribute is per DEVICE: a process that drives several GPUs (model.to("cuda:1"), DataParallel) must raise the
// dynamic-LDS limit of a kernel on each of them.  One flag per (kernel instantiation, device), lock-free after the first call.
#include <atomic>
constexpr int PFK_MAX_DEVICES = 64;
struct pfk_device_once {
  std::atomic<unsigned char> done[PFK_MAX_DEVICES] = {};
  template <class F>
  void run(F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PFK_MAX_DEVICES) { f(); return; }
    if (done[dev].load(std::memory_order_acquire)) return;
    f();   // idempotent: two threads racing here both set the same attribute value
    done[dev].store(1, std::memory_order_release);
  }
};

// pfk_debug_set_* are inert unless the process opted in (include/pfk.h)
#include <stdlib.h>
static inline bool pfk_debug_knobs_enabled() {
  const char* e = getenv("PFK_DEBUG_KNOBS");
  return e && e[0] == '1' && e[1] == 0;
}

static inline int pfk_launch_status() {
  return hipGetLastError() == hipSuccess ? PFK_OK : PFK_ERR_LAUNCH;
}

// Bijective XCD-aware block remap (8 XCDs, dispatcher places block b on XCD b % 8): gives each
// XCD one contiguous range of tile ids so tiles that share operand panels share an L2.
__device__ __forceinline__ int pfk_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, local = bid >> 3;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + local;
}
