// Backward kernels of the correlation path and of the convex upsampling — what torch.autograd executes for
// ptlflow/models/raft/corr.py:29-64 (grid_sample backward, avg_pool2d backward, matmul backward) and raft/raft.py:112-123
// in a training step (train.py, raft-train1-chairs.yaml).  `coords` arrives detached (raft.py:171), so the only gradients on
// the correlation path are those of the two feature maps.
//
//   lookup backward   d(out)[p][l][i][j]  ->  += into the pixel's own level-l gradient map, through the same four bilinear
//                     weights and tap indices the forward used (pfk_lookup.h: shared arithmetic, -ffp-contract=off);
//   volume backward   level l's volume is  C_l = s * F1 . pool^l(F2)^T  (pooling the target dims of the volume == correlating
//                     with the pooled feature map: both are linear and touch the same index sets), so
//                       dF1      += s * dC_l . F2_l            -> pfk_conv2d_f32 as a plain K-contiguous GEMM (K = target pixels)
//                       dF2_l     = s * dC_l^T . F1            -> pfk_conv_wgrad_f32 (the transposed product, reduction over rows)
//                     and the (tiny) chain F2 -> F2_l is left to the caller; no pass over the N x N buffers besides the two GEMMs.
//   convex upsample backward   softmax + 3x3 convex combination + pixel shuffle: gradient of the mask logits and of the flow.
#include "pfk_common.h"
#include "pfk_lookup.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Lookup backward.  Same decomposition as the forward kernel: one workgroup per PIX source pixels, one wave per level,
// every wave on private LDS.  A sample (i, j) of the (2r+1)^2 window reads patch cells (ry[j] + {0,1}, rx[i] + {0,1}) with
// weights {1-wy_j, wy_j} x {1-wx_i, wx_i}: x depends only on i, y only on j, so the scatter is the separable product
//     cell[yy][xx] = sum_j WY[j][yy] * ( sum_i g[i][j] * WX[i][xx] )
// evaluated as two small gathers (deterministic: no atomics, fixed summation order), then one read-modify-write of the
// 12x12 patch into the pixel's gradient map — the map of a source pixel belongs to exactly one wave of one launch.
// Cells outside the map are dropped (the forward's zero padding has zero gradient).
// ------------------------------------------------------------------------------------------------------------------
struct LookupBwdArgs {
  float* gl[PFK_MAX_LEVELS];
  int lh[PFK_MAX_LEVELS];
  int lw[PFK_MAX_LEVELS];
  long long lld[PFK_MAX_LEVELS];
  int L, r, B, h, w;
  const float* coords;
  const float* gout;
  int gout_ld;
};

constexpr int BPIX = 4;

template <int PIX, int R>
__global__ __launch_bounds__(256) void lookup_bwd_kernel(const LookupBwdArgs a) {
  constexpr int n = 2 * R + 1, nn = n * n;
  __shared__ float s_g[4][PIX][nn + 3];
  __shared__ float s_t[4][PIX][n][PATCH];
  __shared__ float s_wx[4][PIX][12], s_wy[4][PIX][12];
  __shared__ int s_rx[4][PIX][12], s_ry[4][PIX][12];

  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const unsigned M = (unsigned)a.B * (unsigned)a.h * (unsigned)a.w;   // < 2^31 (checked by the host): 32-bit scalar arithmetic
  const unsigned p0 = blockIdx.x * (unsigned)PIX;
  const unsigned N = (unsigned)(a.h * a.w);

  float cx0[PIX], cy0[PIX];
  {
    unsigned b = p0 / N, pix = p0 - b * N;
#pragma unroll
    for (int q = 0; q < PIX; ++q) {
      cx0[q] = 0.f; cy0[q] = 0.f;
      if (p0 + q < M) {
        const float* cb = a.coords + (size_t)b * 2 * N + pix;
        cx0[q] = cb[0];
        cy0[q] = cb[N];
      }
      if (++pix == N) { pix = 0; ++b; }
    }
  }

  const int rounds = (a.L + 3) >> 2;
  for (int it = 0; it < rounds; ++it) {
    const int l = it * 4 + wid;
    const bool active = l < a.L;
    int xbi[PIX], ybi[PIX];
    int Hl = 1, Wl = 1;
    if (active) {
      Hl = a.lh[l]; Wl = a.lw[l];
      const float inv = 1.0f / (float)(1 << l);
#pragma unroll
      for (int q = 0; q < PIX; ++q) {
        const float cx = cx0[q] * inv, cy = cy0[q] * inv;
        const float xb = floorf(cx) - (float)(R + 1);
        const float yb = floorf(cy) - (float)(R + 1);
        if (lane < n) {   // identical to the forward kernel's table (pfk_corr.hip)
          const float off = (float)(lane - R);
          const float ix = roundtrip(cx + off, (float)(Wl - 1), (float)(Wl - 1) * 0.5f);
          const float iy = roundtrip(cy + off, (float)(Hl - 1), (float)(Hl - 1) * 0.5f);
          const float x0 = floorf(ix), y0 = floorf(iy);
          const float dxf = x0 - xb, dyf = y0 - yb;
          s_rx[wid][q][lane] = (dxf >= 0.f && dxf <= (float)(PATCH - 2)) ? (int)dxf : 0;
          s_ry[wid][q][lane] = (dyf >= 0.f && dyf <= (float)(PATCH - 2)) ? (int)dyf : 0;
          s_wx[wid][q][lane] = ix - x0;
          s_wy[wid][q][lane] = iy - y0;
        }
        xbi[q] = safe_base(xb); ybi[q] = safe_base(yb);
      }
      for (int idx = lane; idx < PIX * nn; idx += 64) {
        const int q = idx / nn, k = idx - q * nn;
        s_g[wid][q][k] = (p0 + q < M) ? a.gout[(size_t)(p0 + q) * (unsigned)a.gout_ld + l * nn + k] : 0.f;
      }
    }
    wave_lds_sync();
    if (active) {
      for (int idx = lane; idx < PIX * n * PATCH; idx += 64) {
        const int q = idx / (n * PATCH), rem = idx - q * (n * PATCH);
        const int j = rem / PATCH, xx = rem - j * PATCH;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < n; ++i) {
          const int rx = s_rx[wid][q][i];
          const float wx = s_wx[wid][q][i];
          const float wgt = (xx == rx) ? (1.0f - wx) : ((xx == rx + 1) ? wx : 0.f);
          acc = fmaf(s_g[wid][q][i * n + j], wgt, acc);
        }
        s_t[wid][q][j][xx] = acc;
      }
    }
    wave_lds_sync();
    if (active) {
      for (int idx = lane; idx < PIX * PATCH * PATCH; idx += 64) {
        const int q = idx / (PATCH * PATCH), rem = idx - q * (PATCH * PATCH);
        const int yy = rem / PATCH, xx = rem - yy * PATCH;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < n; ++j) {
          const int ry = s_ry[wid][q][j];
          const float wy = s_wy[wid][q][j];
          const float wgt = (yy == ry) ? (1.0f - wy) : ((yy == ry + 1) ? wy : 0.f);
          acc = fmaf(s_t[wid][q][j][xx], wgt, acc);
        }
        int xb_q = xbi[0], yb_q = ybi[0];
#pragma unroll
        for (int qq = 1; qq < PIX; ++qq) if (q == qq) { xb_q = xbi[qq]; yb_q = ybi[qq]; }
        const int gy = yb_q + yy, gx = xb_q + xx;
        if (p0 + q < M && (unsigned)gy < (unsigned)Hl && (unsigned)gx < (unsigned)Wl) {
          float* dst = a.gl[l] + (size_t)(p0 + q) * (size_t)a.lld[l] + (gy * Wl + gx);
          *dst += acc;
        }
      }
    }
    wave_lds_sync();
  }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long long n, float s) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 v = *reinterpret_cast<f32x4*>(x + i);
  v *= s;
  *reinterpret_cast<f32x4*>(x + i) = v;
}

// ------------------------------------------------------------------------------------------------------------------
// Convex upsampling backward (raft/raft.py:112-123).  Forward, per coarse pixel p and sub-pixel (sy, sx) = lane:
//     w_k = softmax_k(mask[p][k*64 + lane]),   out[c][8y+sy][8x+sx] = sum_k w_k * 8 * flow[c][nbr_k(p)]   (zero outside)
// One wave per coarse pixel, lane = sub-pixel (as the forward kernel):
//     dw_k = g_x * vx_k + g_y * vy_k;   dmask[p][k*64 + lane] = w_k * (dw_k - sum_j w_j dw_j)
//     dflow[c][nbr_k(p)] += 8 * sum_lanes w_k * g_c       -> wave-reduced into part[p][k][c]; a second kernel gathers the <= 9
//                                                            contributions of each flow pixel in a fixed order (no atomics).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void convex_upsample_bwd_kernel(const float* __restrict__ flow, int flow_ld,
                                                                  const float* __restrict__ mask, int mask_ld,
                                                                  const float* __restrict__ gout, float* __restrict__ gmask,
                                                                  int gmask_ld, float* __restrict__ part, long long M, int H,
                                                                  int W) {
  const int lane = threadIdx.x & 63;
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= M) return;
  const long long hw = (long long)H * W;
  const long long b = p / hw;
  const int pix = (int)(p - b * hw);
  const int y = pix / W, x = pix - y * W;
  const float* mrow = mask + p * mask_ld + lane;
  float m[9];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) { m[k] = mrow[k * 64]; mx = fmaxf(mx, m[k]); }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); sum += m[k]; }
  const float inv = 1.0f / sum;
  const int sy = lane >> 3, sx = lane & 7;
  const long long HW8 = hw * 64;
  const long long o = (long long)(8 * y + sy) * (8 * W) + 8 * x + sx;
  const float gx = gout[(b * 2 + 0) * HW8 + o], gy = gout[(b * 2 + 1) * HW8 + o];
  const float* fx = flow + (b * 2 + 0) * hw;
  const float* fy = flow + (b * 2 + 1) * hw;
  float dw[9], dot = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    float vx = 0.f, vy = 0.f;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
      if (flow_ld > 0) {
        const float* f = flow + (b * hw + (long long)yy * W + xx) * flow_ld;
        vx = 8.0f * f[0]; vy = 8.0f * f[1];
      } else {
        vx = 8.0f * fx[yy * W + xx]; vy = 8.0f * fy[yy * W + xx];
      }
    }
    m[k] *= inv;
    dw[k] = gx * vx + gy * vy;
    dot += m[k] * dw[k];
  }
  float* grow = gmask + p * gmask_ld + lane;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    grow[k * 64] = m[k] * (dw[k] - dot);
    float cxv = m[k] * gx, cyv = m[k] * gy;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      cxv += __shfl_xor(cxv, s, 64);
      cyv += __shfl_xor(cyv, s, 64);
    }
    if (lane == 0) {
      part[(p * 9 + k) * 2 + 0] = 8.0f * cxv;
      part[(p * 9 + k) * 2 + 1] = 8.0f * cyv;
    }
  }
}

// gflow[b][c][y][x] = sum_k part[pixel (y - ky + 1, x - kx + 1)][k][c]   (the pixels whose k-th neighbour is (y, x))
__global__ __launch_bounds__(256) void convex_upsample_bwd_gather_kernel(const float* __restrict__ part,
                                                                         float* __restrict__ gflow, long long M, int H, int W) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const long long hw = (long long)H * W;
  const long long b = p / hw;
  const int pix = (int)(p - b * hw);
  const int y = pix / W, x = pix - y * W;
  float ax = 0.f, ay = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y - (k / 3 - 1), xx = x - (k % 3 - 1);
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
      const long long src = b * hw + (long long)yy * W + xx;
      ax += part[(src * 9 + k) * 2 + 0];
      ay += part[(src * 9 + k) * 2 + 1];
    }
  }
  gflow[(b * 2 + 0) * hw + pix] = ax;
  gflow[(b * 2 + 1) * hw + pix] = ay;
}

inline int round_up32(int c) { return (c + 31) & ~31; }

}  // namespace

extern "C" {

int pfk_corr_lookup_bwd_f32(const pfk_lookup_bwd_desc* d, pfk_stream_t stream) {
  if (!d || !d->coords || !d->grad_out) return PFK_ERR_BAD_ARG;
  if (d->num_levels < 1 || d->num_levels > PFK_MAX_LEVELS) return PFK_ERR_BAD_ARG;
  if (d->radius < 1 || d->radius > 4) return PFK_ERR_UNSUPPORTED;
  if (d->B <= 0 || d->h <= 0 || d->w <= 0) return PFK_ERR_BAD_ARG;
  const int n = 2 * d->radius + 1;
  if (d->grad_out_ld < d->num_levels * n * n) return PFK_ERR_BAD_ARG;
  LookupBwdArgs a{};
  for (int l = 0; l < d->num_levels; ++l) {
    if (!d->grad_levels[l] || d->lvl_h[l] <= 0 || d->lvl_w[l] <= 0) return PFK_ERR_BAD_ARG;
    if (d->lvl_ld[l] < (long long)d->lvl_h[l] * d->lvl_w[l]) return PFK_ERR_BAD_ARG;
    a.gl[l] = d->grad_levels[l]; a.lh[l] = d->lvl_h[l]; a.lw[l] = d->lvl_w[l]; a.lld[l] = d->lvl_ld[l];
  }
  a.L = d->num_levels; a.r = d->radius; a.B = d->B; a.h = d->h; a.w = d->w;
  a.coords = d->coords; a.gout = d->grad_out; a.gout_ld = d->grad_out_ld;
  if ((long long)d->B * d->h * d->w >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  const long long blocks = ((long long)d->B * d->h * d->w + BPIX - 1) / BPIX;
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (d->radius) {
    case 1: hipLaunchKernelGGL((lookup_bwd_kernel<BPIX, 1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((lookup_bwd_kernel<BPIX, 2>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((lookup_bwd_kernel<BPIX, 3>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((lookup_bwd_kernel<BPIX, 4>), grid, block, 0, st, a); break;
  }
  return pfk_launch_status();
}

long long pfk_corr_volume_bwd_workspace_bytes(int N1, int ldc, int D) {
  if (N1 <= 0 || ldc <= 0 || D <= 0) return 0;
  pfk_conv_desc w{};
  w.num_src = 1; w.src[0].channels = D; w.src[0].ld = D; w.B = 1; w.H = 1; w.W = N1; w.kh = 1; w.kw = 1; w.cout = ldc;
  return pfk_conv_wgrad_workspace_bytes(&w, 0);
}

int pfk_corr_volume_bwd_f32(const float* dC, int ldc, int N1, int N2, const float* f1, int ld1, const float* f2_cm, int ld2cm,
                            int D, float scale, float* df1, int df1_ld, int accumulate_df1, float* df2, void* workspace,
                            long long workspace_bytes, pfk_stream_t stream) {
  if (!dC || !f1 || !f2_cm || !df1 || !df2 || N1 <= 0 || N2 <= 0 || D <= 0) return PFK_ERR_BAD_ARG;
  if (ldc < N2 || ld1 < D || df1_ld < D) return PFK_ERR_BAD_ARG;
  if ((ldc & 3) || (ld1 & 3) || (D & 3) || (df1_ld & 3)) return PFK_ERR_ALIGNMENT;
  if (ld2cm != round_up32(ldc)) return PFK_ERR_UNSUPPORTED;   // f2_cm rows are read as packed weight rows of the GEMM below
  // dF1 (+)= scale * dC . F2 : 1x1 "convolution" over the N1 source pixels, K = target pixels (contiguous in dC rows)
  pfk_conv_desc g{};
  g.num_src = 1; g.src[0].ptr = dC; g.src[0].ld = ldc; g.src[0].channels = ldc;
  g.B = 1; g.H = 1; g.W = N1; g.kh = 1; g.kw = 1;
  g.weight = f2_cm; g.bias = nullptr; g.cout = D; g.epilogue = PFK_EPI_LINEAR; g.relu = 0; g.scale = scale;
  g.out = df1; g.out_ld = df1_ld; g.out_coff = 0;
  if (accumulate_df1) { g.residual = df1; g.residual_ld = df1_ld; }
  int rc = pfk_conv2d_f32(&g, stream);
  if (rc != PFK_OK) return rc;
  // dF2 = scale * dC^T . F1 : the transposed product (reduction over the N1 rows) = the weight-gradient kernel with
  // "output channels" = target pixels and "input channels" = feature channels; result [ldc][round_up32(D)]
  pfk_conv_desc w{};
  w.num_src = 1; w.src[0].ptr = f1; w.src[0].ld = ld1; w.src[0].channels = D;
  w.B = 1; w.H = 1; w.W = N1; w.kh = 1; w.kw = 1; w.cout = ldc;
  rc = pfk_conv_wgrad_f32(&w, dC, ldc, df2, 0, workspace, workspace_bytes, stream);
  if (rc != PFK_OK) return rc;
  const long long n = (long long)ldc * round_up32(D);
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), df2, n,
                     scale);
  return pfk_launch_status();
}

long long pfk_convex_upsample_bwd_workspace_bytes(int B, int H, int W) {
  return (long long)B * H * W * 18 * (long long)sizeof(float);
}

int pfk_convex_upsample_bwd_f32(const float* flow, int flow_ld, const float* mask, int mask_ld, const float* grad_out,
                                float* grad_mask, int grad_mask_ld, float* grad_flow, void* workspace, long long workspace_bytes,
                                int B, int H, int W, pfk_stream_t stream) {
  if (!flow || !mask || !grad_out || !grad_mask || !grad_flow || !workspace || B <= 0 || H <= 0 || W <= 0) return PFK_ERR_BAD_ARG;
  if (mask_ld < 576 || grad_mask_ld < 576 || flow_ld < 0 || flow_ld == 1) return PFK_ERR_BAD_ARG;
  if (workspace_bytes < pfk_convex_upsample_bwd_workspace_bytes(B, H, W)) return PFK_ERR_BAD_ARG;
  const long long M = (long long)B * H * W;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(convex_upsample_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, flow, flow_ld, mask, mask_ld,
                     grad_out, grad_mask, grad_mask_ld, part, M, H, W);
  hipLaunchKernelGGL(convex_upsample_bwd_gather_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, part, grad_flow, M, H, W);
  return pfk_launch_status();
}

}  // extern "C"
