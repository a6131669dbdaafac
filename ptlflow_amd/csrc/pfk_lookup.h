// Coordinate arithmetic shared by the pyramid lookup (pfk_corr.hip) and its backward (pfk_bwd.hip): both must derive the
// SAME tap indices and weights from a coordinate, so the chain lives in one place.  Every translation unit that includes
// this header is compiled with -ffp-contract=off (see pfk_corr.hip's header for why).
#pragma once
#include "pfk_common.h"

namespace {

constexpr int PATCH = 12;      // staged window: (2r+2) taps + one ring for the +-1 index wobble of the round trip (r <= 4)
constexpr int PATCH_LD = 13;

// pixel -> normalised -> pixel, every step rounded (see file header).
__device__ __forceinline__ float roundtrip(float p, float size_m1, float half_span) {
  float g = 2.0f * p;
  g = g / size_m1;       // correctly rounded IEEE division (hipcc default for fp32)
  g = g - 1.0f;
  float ix = g + 1.0f;
  return ix * half_span;
}

__device__ __forceinline__ int safe_base(float v) {
  // integer-valued float -> int; anything non-finite or absurd maps far outside every map so the
  // patch is staged as zeros (and the weights carry the NaN, as in the reference).
  return (fabsf(v) < 1.0e9f) ? (int)v : -(1 << 30);
}

// LDS operations of one wave complete in order: a wave that only talks to itself through LDS needs no workgroup barrier.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

}  // namespace
