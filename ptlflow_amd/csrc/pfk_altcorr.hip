// On-demand ("alternate") correlation: the local (2r+1)^2 correlation window of every pixel computed straight
// from the feature maps, without materialising the all-pairs volume — the job of the reference's optional
// alt_cuda_corr extension (ptlflow/utils/external/alt_cuda_corr/correlation_kernel.cu:18-119), rewritten for
// wave64 / gfx950:
//
//   for pixel p with target coordinate (x, y):  fx = floor(x), fy = floor(y), dx = x - fx, dy = y - fy
//   s[iy][ix] = <fmap1[p, :], fmap2[fy - r + iy, fx - r + ix, :]>      iy, ix in [0, 2r+1]   (0 outside the map)
//   out[oy + (2r+1)*ox] = s[oy][ox](1-dy)(1-dx) + s[oy][ox+1](1-dy)dx + s[oy+1][ox]dy(1-dx) + s[oy+1][ox+1]dy dx
//
// (x-offset-major window, unscaled: the caller divides by sqrt(C), raft/corr.py:101.)
// One wave per pixel: the 64 lanes form 4 groups of 16; a group takes one tap, its 16 lanes split the channels
// in float4 (one coalesced 256-byte segment of the NHWC fmap2 row per load), a 4-step butterfly finishes the dot
// product.  fmap1's row lives in registers.  The (2r+2)^2 tap values go through LDS, then the lanes write the
// (2r+1)^2 interpolated cells.  HBM/L2-bound gather: (2r+2)^2 * C * 4 bytes read per pixel.
#include "pfk_common.h"

namespace {

constexpr int MAX_C4 = 8;   // channels <= 16 lanes * 4 floats * MAX_C4 = 512

__global__ __launch_bounds__(256) void altcorr_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                          const float* __restrict__ coords, float* __restrict__ out,
                                                          long long M, int H1, int W1, int H2, int W2, int C, int r) {
  __shared__ float s_tap[4][104];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wid;
  const bool live = p < M;
  const int g = lane >> 4, cl = lane & 15;
  const int rd = 2 * r + 1, n = rd + 1, ntaps = n * n;

  float x = 0.f, y = 0.f;
  long long b = 0;
  int pix = 0;
  if (live) {
    x = coords[p * 2 + 0];
    y = coords[p * 2 + 1];
    const long long hw = (long long)H1 * W1;
    b = p / hw;
    pix = (int)(p - b * hw);
  }
  const float fx = floorf(x), fy = floorf(y);
  const float dx = x - fx, dy = y - fy;
  // non-finite / absurd coordinates: every tap out of bounds (the weights still carry the NaN)
  const int x0 = (fabsf(fx) < 1.0e9f) ? (int)fx - r : -(1 << 30);
  const int y0 = (fabsf(fy) < 1.0e9f) ? (int)fy - r : -(1 << 30);

  f32x4 a[MAX_C4];
#pragma unroll
  for (int i = 0; i < MAX_C4; ++i) {
    const int c = cl * 4 + i * 64;
    a[i] = (live && c < C) ? *reinterpret_cast<const f32x4*>(f1 + p * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int t0 = 0; t0 < ntaps; t0 += 4) {
    const int t = t0 + g;
    const int iy = t / n, ix = t - iy * n;
    const int yy = y0 + iy, xx = x0 + ix;
    const bool ok = live && t < ntaps && (unsigned)yy < (unsigned)H2 && (unsigned)xx < (unsigned)W2;
    float acc = 0.f;
    if (ok) {
      const float* row = f2 + ((b * H2 + yy) * (long long)W2 + xx) * C;
#pragma unroll
      for (int i = 0; i < MAX_C4; ++i) {
        const int c = cl * 4 + i * 64;
        if (c < C) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
          acc = fmaf(a[i].x, v.x, fmaf(a[i].y, v.y, fmaf(a[i].z, v.z, fmaf(a[i].w, v.w, acc))));
        }
      }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (cl == 0 && t < ntaps) s_tap[wid][t] = acc;
  }
  __syncthreads();
  if (!live) return;
  const float ex = 1.0f - dx, ey = 1.0f - dy;
  const long long hw = (long long)H1 * W1;
  float* o = out + b * rd * rd * hw + pix;
  for (int k = lane; k < rd * rd; k += 64) {
    const int ox = k / rd, oy = k - ox * rd;
    const float s00 = s_tap[wid][oy * n + ox], s01 = s_tap[wid][oy * n + ox + 1];
    const float s10 = s_tap[wid][(oy + 1) * n + ox], s11 = s_tap[wid][(oy + 1) * n + ox + 1];
    o[k * hw] = s00 * ey * ex + s01 * ey * dx + s10 * dy * ex + s11 * dy * dx;
  }
}

// Backward of the above w.r.t. both feature maps (correlation_kernel.cu:122-256): per pixel, the gradient of the
// (2r+1)^2 cells is folded back onto the (2r+2)^2 taps (transpose of the bilinear combine), then
//   fmap1_grad[p]   = sum_t gs[t] * fmap2[tap t]          (wave-local, one butterfly across the 4 tap groups)
//   fmap2_grad[tap] += gs[t] * fmap1[p]                    (fp32 atomics, like the reference's atomicAdd at :237;
//                                                           summation order over pixels is therefore not fixed)
// The reference allocates coords_grad but never writes it (:307); neither do we.
__global__ __launch_bounds__(256) void altcorr_bwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                          const float* __restrict__ coords,
                                                          const float* __restrict__ grad, float* __restrict__ f1g,
                                                          float* __restrict__ f2g, long long M, int H1, int W1, int H2,
                                                          int W2, int C, int r) {
  __shared__ float s_g[4][84];
  __shared__ float s_gs[4][104];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wid;
  const bool live = p < M;
  const int g = lane >> 4, cl = lane & 15;
  const int rd = 2 * r + 1, n = rd + 1, ntaps = n * n;
  const long long hw = (long long)H1 * W1;

  float x = 0.f, y = 0.f;
  long long b = 0;
  int pix = 0;
  if (live) {
    x = coords[p * 2 + 0];
    y = coords[p * 2 + 1];
    b = p / hw;
    pix = (int)(p - b * hw);
    for (int k = lane; k < rd * rd; k += 64) s_g[wid][k] = grad[(b * rd * rd + k) * hw + pix];
  }
  __syncthreads();
  const float fx = floorf(x), fy = floorf(y);
  const float dx = x - fx, dy = y - fy, ex = 1.0f - dx, ey = 1.0f - dy;
  const int x0 = (fabsf(fx) < 1.0e9f) ? (int)fx - r : -(1 << 30);
  const int y0 = (fabsf(fy) < 1.0e9f) ? (int)fy - r : -(1 << 30);
  if (live) {
    for (int t = lane; t < ntaps; t += 64) {
      const int iy = t / n, ix = t - iy * n;
      float v = 0.f;
      if (iy < rd && ix < rd) v += s_g[wid][iy + rd * ix] * ey * ex;
      if (iy < rd && ix > 0) v += s_g[wid][iy + rd * (ix - 1)] * ey * dx;
      if (iy > 0 && ix < rd) v += s_g[wid][(iy - 1) + rd * ix] * dy * ex;
      if (iy > 0 && ix > 0) v += s_g[wid][(iy - 1) + rd * (ix - 1)] * dy * dx;
      s_gs[wid][t] = v;
    }
  }
  __syncthreads();

  f32x4 a[MAX_C4], ga[MAX_C4];
#pragma unroll
  for (int i = 0; i < MAX_C4; ++i) {
    const int c = cl * 4 + i * 64;
    a[i] = (live && c < C) ? *reinterpret_cast<const f32x4*>(f1 + p * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    ga[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int t0 = 0; t0 < ntaps; t0 += 4) {
    const int t = t0 + g;
    const int iy = t / n, ix = t - iy * n;
    const int yy = y0 + iy, xx = x0 + ix;
    const bool ok = live && t < ntaps && (unsigned)yy < (unsigned)H2 && (unsigned)xx < (unsigned)W2;
    if (ok) {
      const float wgt = s_gs[wid][t];
      const long long rowoff = ((b * H2 + yy) * (long long)W2 + xx) * C;
#pragma unroll
      for (int i = 0; i < MAX_C4; ++i) {
        const int c = cl * 4 + i * 64;
        if (c < C) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(f2 + rowoff + c);
          ga[i].x = fmaf(wgt, v.x, ga[i].x); ga[i].y = fmaf(wgt, v.y, ga[i].y);
          ga[i].z = fmaf(wgt, v.z, ga[i].z); ga[i].w = fmaf(wgt, v.w, ga[i].w);
          float* dst = f2g + rowoff + c;
          atomicAdd(dst + 0, wgt * a[i].x); atomicAdd(dst + 1, wgt * a[i].y);
          atomicAdd(dst + 2, wgt * a[i].z); atomicAdd(dst + 3, wgt * a[i].w);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAX_C4; ++i) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      ga[i].x += __shfl_xor(ga[i].x, off, 64); ga[i].y += __shfl_xor(ga[i].y, off, 64);
      ga[i].z += __shfl_xor(ga[i].z, off, 64); ga[i].w += __shfl_xor(ga[i].w, off, 64);
    }
    const int c = cl * 4 + i * 64;
    if (live && g == 0 && c < C) *reinterpret_cast<f32x4*>(f1g + p * C + c) = ga[i];
  }
}

}  // namespace

extern "C" {

int pfk_altcorr_backward_f32(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                             float* fmap1_grad, float* fmap2_grad, int B, int H1, int W1, int H2, int W2, int C,
                             int radius, pfk_stream_t stream) {
  if (!fmap1 || !fmap2 || !coords || !corr_grad || !fmap1_grad || !fmap2_grad) return PFK_ERR_BAD_ARG;
  if (B <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0) return PFK_ERR_BAD_ARG;
  if (radius < 1 || radius > 4) return PFK_ERR_UNSUPPORTED;
  if ((C & 3) || !pfk_aligned16(fmap1) || !pfk_aligned16(fmap2) || !pfk_aligned16(fmap1_grad)) return PFK_ERR_ALIGNMENT;
  if (C > 64 * MAX_C4) return PFK_ERR_UNSUPPORTED;
  const long long M = (long long)B * H1 * W1;
  const long long blocks = (M + 3) / 4;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(fmap2_grad, 0, (size_t)B * H2 * W2 * C * sizeof(float), st) != hipSuccess) return PFK_ERR_LAUNCH;
  hipLaunchKernelGGL(altcorr_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, fmap1, fmap2, coords, corr_grad,
                     fmap1_grad, fmap2_grad, M, H1, W1, H2, W2, C, radius);
  return pfk_launch_status();
}


int pfk_altcorr_forward_f32(const float* fmap1, const float* fmap2, const float* coords, float* out, int B, int H1,
                            int W1, int H2, int W2, int C, int radius, pfk_stream_t stream) {
  if (!fmap1 || !fmap2 || !coords || !out) return PFK_ERR_BAD_ARG;
  if (B <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0) return PFK_ERR_BAD_ARG;
  if (radius < 1 || radius > 4) return PFK_ERR_UNSUPPORTED;
  if ((C & 3) || !pfk_aligned16(fmap1) || !pfk_aligned16(fmap2)) return PFK_ERR_ALIGNMENT;
  if (C > 64 * MAX_C4) return PFK_ERR_UNSUPPORTED;
  const long long M = (long long)B * H1 * W1;
  const long long blocks = (M + 3) / 4;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(altcorr_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), fmap1,
                     fmap2, coords, out, M, H1, W1, H2, W2, C, radius);
  return pfk_launch_status();
}

}  // extern "C"
