// Shared helpers for the libpfk.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PFK_WAVE 64

static inline bool pfk_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

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
