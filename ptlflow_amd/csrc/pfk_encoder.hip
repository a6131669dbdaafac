// Feature / context encoder pieces that are not implicit-GEMM convolutions (reference:
// ptlflow/models/raft/extractor.py:122-194 BasicEncoder):
//   * the stem: Conv2d(3, C, 7, stride 2, padding 3) read straight from the NCHW image (K = 147 is too thin for the
//     32-channel K-steps of the MFMA kernels), VALU FMA kernel, output pixel-major;
//   * instance-norm statistics (one pass: sum and sum of squares per (image, channel) in double, deterministic: per-chunk
//     partials reduced in a fixed order, no atomics), and their backward;
//   * normalise + relu (+ residual add + relu) elementwise pass.
// The 3x3 / 1x1 convolutions of the residual blocks (stride 1 and 2) run on pfk_conv2d_f32 / pfk_conv2d_bf16s;
// batch-norm in eval mode is folded into their weights by the host (ptlflow_amd/encoder.py).
#include "pfk_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Stem.  One workgroup = 32 consecutive output pixels of one output row x all output channels; thread = one channel x
// 8 pixels.  The 3 x 7 x 69 input patch (zero padding applied here) is staged once in LDS with row-contiguous loads;
// per (channel, ky) a thread pulls its 21 patch columns into registers (LDS broadcast reads: the 64 lanes of a wave
// share them) and reuses them across the 7 kx, so the inner loop is 56 FMAs per 7 coalesced weight loads.
// ------------------------------------------------------------------------------------------------------------------
constexpr int STEM_K = 7, STEM_R = 3, STEM_TP = 32, STEM_NP = 8;
constexpr int STEM_PW = 2 * STEM_TP + STEM_K - 2;   // 69 input columns feed 32 stride-2 outputs

__global__ __launch_bounds__(256) void conv_stem_kernel(const float* __restrict__ img, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int out_ld, int H, int W, int Ho, int Wo, int tiles_per_row,
                                                        int cout, int relu) {
  __shared__ float sp[3][STEM_K][STEM_PW + 3];
  const int tile = blockIdx.x;
  const int orow = tile / tiles_per_row;           // b*Ho + yo
  const int xo0 = (tile - orow * tiles_per_row) * STEM_TP;
  const int yo = orow % Ho;
  const int b = orow / Ho;
  const int iy0 = 2 * yo - STEM_R, ix0 = 2 * xo0 - STEM_R;
  const float* src = img + (long long)b * 3 * H * W;
  for (int e = threadIdx.x; e < 3 * STEM_K * STEM_PW; e += 256) {
    const int ci = e / (STEM_K * STEM_PW);
    const int r = e - ci * (STEM_K * STEM_PW);
    const int ky = r / STEM_PW, xx = r - ky * STEM_PW;
    const int gy = iy0 + ky, gx = ix0 + xx;
    float v = 0.f;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = src[((long long)ci * H + gy) * W + gx];
    sp[ci][ky][xx] = v;
  }
  __syncthreads();
  const int g = threadIdx.x >> 6;                  // pixel group: 8 outputs
  for (int c = threadIdx.x & 63; c < cout; c += 64) {
    float acc[STEM_NP];
    const float b0 = bias ? bias[c] : 0.f;
#pragma unroll
    for (int p = 0; p < STEM_NP; ++p) acc[p] = b0;
#pragma unroll 1   // a fully unrolled body hoists all 147 weight loads: 512 VGPRs and scratch
    for (int ky = 0; ky < STEM_K; ++ky) {
#pragma unroll 1
      for (int ci = 0; ci < 3; ++ci) {
        float v[2 * STEM_NP + STEM_K - 2];
#pragma unroll
        for (int j = 0; j < 2 * STEM_NP + STEM_K - 2; ++j) v[j] = sp[ci][ky][2 * STEM_NP * g + j];
#pragma unroll
        for (int kx = 0; kx < STEM_K; ++kx) {
          const float w = wgt[((ky * STEM_K + kx) * 3 + ci) * cout + c];
#pragma unroll
          for (int p = 0; p < STEM_NP; ++p) acc[p] = fmaf(v[2 * p + kx], w, acc[p]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < STEM_NP; ++p) {
      const int xo = xo0 + g * STEM_NP + p;
      if (xo < Wo) {
        float r = acc[p];
        if (relu) r = (r < 0.f) ? 0.f : r;
        out[((long long)orow * Wo + xo) * out_ld + c] = r;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Stem on the matrix cores (round 6; cout = 32 or 64 — SmallEncoder / BasicEncoder).  The VALU kernel above runs the 33.6 GFLOP of a
// 16-image 436x1024 stem at ~35 TFLOP/s (0.85 ms); as an implicit GEMM M = pixels, N = cout, K = 147 (+1 zero row) on
// v_mfma_f32_32x32x2f32 the same launch is bound by the fp32 matrix pipe at ~0.21 ms.
// One workgroup = 8 x 32 output pixels x all channels; four waves, wave w owns output rows 2w, 2w + 1 (two 32-pixel M tiles) x NT
// 32-channel N tiles.  The 3 x 21 x 69 input patch (zero padding applied while staging) and the whole [148][cout] weight sit in LDS
// (17 + 37 KB at cout 64: two workgroups per CU).  K-step s covers k = 2s, 2s + 1 with k = (ky * 7 + kx) * 3 + ci, the order of the
// packed weight: lane (pixel l & 31, k-half l >> 5) reads ITS element of the im2col row straight from the patch —
// patch[ci][2 py + ky][2 px + kx], the per-k offset a compile-time constant selected by the lane half — so no im2col matrix exists
// anywhere.  The K loop is fully unrolled (74 steps x (2 + NT) ds_read_b32 + 2 NT MFMAs).
// ------------------------------------------------------------------------------------------------------------------
constexpr int SM_ROWS = 8, SM_COLS = 32;
constexpr int SM_PR = 2 * SM_ROWS + STEM_K - 2, SM_PC = 2 * SM_COLS + STEM_K - 2;   // 21 x 69
constexpr int SM_KTOT = 3 * STEM_K * STEM_K, SM_KSTEPS = (SM_KTOT + 1) / 2;       // 147 -> 74 steps of 2

__host__ __device__ constexpr int sm_koff(int k) {           // patch offset of im2col column k
  return (k % 3) * (SM_PR * SM_PC) + ((k / 3) / STEM_K) * SM_PC + (k / 3) % STEM_K;
}

bool g_stem_valu = false;      // pfk_debug_set_stem_valu(1): the VALU kernel for every width (A/B timing, tests)

template <int NT, int S>
struct StemSteps {
  static __device__ __forceinline__ void run(f32x16 (&acc)[2][NT], const float* s_in, const float* s_w, const int (&abase)[2], int hl, int px) {
    constexpr int k0 = 2 * S, k1 = (2 * S + 1 < SM_KTOT) ? 2 * S + 1 : SM_KTOT - 1;   // k = 147: the weight row is zero, any valid address
    const int o = hl ? sm_koff(k1) : sm_koff(k0);
    const float a0 = s_in[abase[0] + o], a1 = s_in[abase[1] + o];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float b = s_w[((2 * S) + hl) * (NT * 32) + nt * 32 + px];
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][nt], 0, 0, 0);
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][nt], 0, 0, 0);
    }
    if constexpr (S + 1 < SM_KSTEPS) StemSteps<NT, S + 1>::run(acc, s_in, s_w, abase, hl, px);
  }
};

// Persistent: the grid is two workgroups per CU, each stages the weight ONCE and walks the tile list with stride gridDim.x; the next
// tile's patch is requested into registers before the current tile's MFMA loop and parked in the other LDS buffer behind it, so the
// global-load latency of the staging hides under the matrix work (one __syncthreads per tile).
constexpr int SM_PATCH = 3 * SM_PR * SM_PC, SM_PT = (SM_PATCH + 255) / 256;     // 4347 floats, 17 per thread

template <int NT, typename TO>
__global__ __launch_bounds__(256, 2) void conv_stem_mfma_kernel(const float* __restrict__ img, const float* __restrict__ wgt,
                                                                 const float* __restrict__ bias, TO* __restrict__ out, int out_ld,
                                                                 int H, int W, int Ho, int Wo, int tiles_x, int tiles_y, int ntiles,
                                                                 int relu) {
  constexpr int COUT = NT * 32;
  __shared__ float s_in[2][SM_PATCH];
  __shared__ float s_w[2 * SM_KSTEPS * COUT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int px = lane & 31, hl = lane >> 5;
  // this thread's patch elements: e = tid + 256 j -> (ci, yy, xx), the same for every tile
  int poff[SM_PT];            // ci * H * W + yy * W + xx, relative to the patch origin (may be used with a negative origin)
  short pyy[SM_PT], pxx[SM_PT];
#pragma unroll
  for (int j = 0; j < SM_PT; ++j) {
    const int e = threadIdx.x + 256 * j;
    const int ci = e / (SM_PR * SM_PC);
    const int r = e - ci * (SM_PR * SM_PC);
    const int yy = r / SM_PC, xx = r - yy * SM_PC;
    pyy[j] = (short)yy; pxx[j] = (short)xx;
    poff[j] = (ci * H + yy) * W + xx;
  }
  float pv[SM_PT];
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x, t1 = tile / tiles_x;
    const int ty = t1 % tiles_y, b = t1 / tiles_y;
    const int iy0 = 2 * ty * SM_ROWS - STEM_R, ix0 = 2 * tx * SM_COLS - STEM_R;
    const float* src = img + (long long)b * 3 * H * W + (long long)iy0 * W + ix0;
#pragma unroll
    for (int j = 0; j < SM_PT; ++j) {
      const int gy = iy0 + pyy[j], gx = ix0 + pxx[j];
      pv[j] = 0.f;
      if (threadIdx.x + 256 * j < SM_PATCH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) pv[j] = src[poff[j]];
    }
  };
  auto park = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < SM_PT; ++j)
      if (threadIdx.x + 256 * j < SM_PATCH) dst[threadIdx.x + 256 * j] = pv[j];
  };

  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (int e = threadIdx.x; e < 2 * SM_KSTEPS * COUT / 4; e += 256) {        // [147][COUT] + one zero row, 16 bytes per thread
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (e * 4 < SM_KTOT * COUT) v = *reinterpret_cast<const f32x4*>(wgt + e * 4);
    *reinterpret_cast<f32x4*>(s_w + e * 4) = v;
  }
  if (tile < ntiles) park(s_in[0]);
  __syncthreads();
  float b0[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b0[nt] = bias ? bias[nt * 32 + px] : 0.f;
  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) abase[mt] = (2 * (2 * wid + mt)) * SM_PC + 2 * px;

  for (int buf = 0; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const int next = tile + gridDim.x;
    if (next < ntiles) fetch(next);
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    StemSteps<NT, 0>::run(acc, s_in[buf], s_w, abase, hl, px);
    // accumulator layout: register r of lane l = pixel (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the M tile, channel l & 31 of the N
    // tile: a store instruction writes two 128-byte (fp32) row segments
    const int tx = tile % tiles_x, t1 = tile / tiles_x;
    const int ty = t1 % tiles_y, b = t1 / tiles_y;
    const int yo0 = ty * SM_ROWS, xo0 = tx * SM_COLS;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = nt * 32 + px;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int yo = yo0 + 2 * wid + mt;
        if (yo >= Ho) continue;
        const long long row0 = ((long long)b * Ho + yo) * Wo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int xo = xo0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
          if (xo >= Wo) continue;
          float v = acc[mt][nt][r] + b0[nt];
          if (relu) v = (v < 0.f) ? 0.f : v;
          out[(row0 + xo) * out_ld + c] = (TO)v;
        }
      }
    }
    if (next < ntiles) park(s_in[buf ^ 1]);
    __syncthreads();      // the next patch is complete; every wave is done reading this one before the tile after next overwrites it
  }
}

template <typename TO>
int stem_launch(const float* img, const float* weight, const float* bias, TO* out, int out_ld, int B, int H, int W, int cout, int relu,
                hipStream_t st) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int tiles_x = (Wo + SM_COLS - 1) / SM_COLS, tiles_y = (Ho + SM_ROWS - 1) / SM_ROWS;
  const long long tiles = (long long)B * tiles_x * tiles_y;
  if (tiles > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  if ((long long)3 * H * W >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;     // 32-bit offsets inside one image
  const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);               // two workgroups per CU, persistent over the tile list
  if (cout == 64)
    hipLaunchKernelGGL((conv_stem_mfma_kernel<2, TO>), dim3(grid), dim3(256), 0, st, img, weight, bias, out, out_ld, H, W, Ho, Wo,
                       tiles_x, tiles_y, (int)tiles, relu);
  else if (cout == 32)
    hipLaunchKernelGGL((conv_stem_mfma_kernel<1, TO>), dim3(grid), dim3(256), 0, st, img, weight, bias, out, out_ld, H, W, Ho, Wo,
                       tiles_x, tiles_y, (int)tiles, relu);
  else
    return PFK_ERR_UNSUPPORTED;
  return pfk_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Instance-norm statistics over pixel-major x[B*HW][ld], channels [0, C), C % 4 == 0.
// ONE pass over x: per (image, chunk) partial sums of x and of x*x, both carried in double (an fp32 square is exact in double), reduced
// over the chunks in a fixed order (chunk_reduce_kernel); finish: mean, var_biased = E[x^2] - mean^2 (in double: the cancellation
// costs mean^2/var * 2^-53, nothing next to the fp32 rounding of the results), rstd = 1/sqrt(var + eps)  (F.instance_norm,
// extractor.py:136-140).  A thread owns 4 consecutive channels (float4) and every (256 / (C/4))-th row of the chunk.
// ------------------------------------------------------------------------------------------------------------------
constexpr int IN_CHUNKS_MAX = 256;   // partial-sum slots per image in the workspace
// chunks per image: enough blocks to fill the chip at small batches (the partial kernels are latency-bound per block), 32 once
// the batch alone provides them
static inline int in_chunks(int B) { int c = 1024 / (B > 0 ? B : 1); return c < 32 ? 32 : (c > IN_CHUNKS_MAX ? IN_CHUNKS_MAX : c); }

// Sums are carried in DOUBLE: a channel whose mean is large next to its spread (common after a biased convolution) loses the
// low digits of the variance when the sums carry fp32 accumulation error, and everything downstream of the norm —
// the normalised activations, their relu masks and, in training, the weight gradients of the layers in front of it — inherits
// that error (measured on BasicEncoder gradients: 9e-3 with fp32 sums, 1e-6 with double; MIOpen's fp32 kernels 9e-4).
typedef double f64x4s __attribute__((ext_vector_type(4)));

// four consecutive channels of a pixel-major row as fp32: fp32 rows as they are, bf16 rows widened exactly
__device__ __forceinline__ f32x4 ld4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4f(const __bf16* p) {
  typedef unsigned int u32x2e __attribute__((ext_vector_type(2)));
  const u32x2e w = *reinterpret_cast<const u32x2e*>(p);
  f32x4 v;
  v[0] = __builtin_bit_cast(float, w[0] << 16); v[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, w[1] << 16); v[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
  return v;
}

// part: two arrays [B][nch][C] one after the other (sum, then sum of squares; `astride` doubles apart)
// TX = __bf16: the statistics of a bf16 convolution output (what F.instance_norm sees under the reference's autocast switch)
template <typename TX>
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const TX* __restrict__ x, int ld, int C, int HW,
                                                               double* __restrict__ part, long long astride) {
  __shared__ f64x4s red[2][256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tpr = C >> 2;                   // threads per row
  const int rpi = 256 / tpr;                // rows per iteration
  const int t = threadIdx.x;
  const int c4 = (t % tpr) * 4, rr = t / tpr;
  const bool active = rr < rpi;
  const int nch = (int)gridDim.x;
  const int rows = (HW + nch - 1) / nch;
  const int r0 = chunk * rows, r1 = min(HW, r0 + rows);
  f64x4s s0 = {0., 0., 0., 0.}, q0 = {0., 0., 0., 0.};
  if (active) {
    const TX* base = x + (long long)b * HW * ld + c4;
    // two independent accumulator sets: the fp64 add / fma chain of one set would otherwise serialise the loop (a chunk has
    // only a few hundred rows per thread and, at batch 1, the grid is far from filling the chip)
    f64x4s s1 = {0., 0., 0., 0.}, q1 = {0., 0., 0., 0.};
    int r = r0 + rr;
    for (; r + rpi < r1; r += 2 * rpi) {
      const f32x4 v0 = ld4f(base + (long long)r * ld);
      const f32x4 v1 = ld4f(base + (long long)(r + rpi) * ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d0 = (double)v0[e], d1 = (double)v1[e];
        s0[e] += d0; q0[e] = fma(d0, d0, q0[e]);
        s1[e] += d1; q1[e] = fma(d1, d1, q1[e]);
      }
    }
    if (r < r1) {
      const f32x4 v = ld4f(base + (long long)r * ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)v[e];
        s0[e] += d; q0[e] = fma(d, d, q0[e]);
      }
    }
    s0 += s1; q0 += q1;
  }
  red[0][t] = s0;
  red[1][t] = q0;
  __syncthreads();
  for (int u = t; u < 2 * tpr; u += 256) {   // [0, tpr): sums; [tpr, 2 tpr): sums of squares
    const int which = u >= tpr, tt = u - which * tpr;
    f64x4s s = red[which][tt];
    for (int k = 1; k < rpi; ++k) s += red[which][tt + k * tpr];
    *reinterpret_cast<f64x4s*>(part + which * astride + ((long long)b * nch + chunk) * C + tt * 4) = s;
  }
}

// part[b][k][c], k < nch  ->  tot[b][c] = sum_k, for NA arrays laid out one after the other (stride `astride` doubles in part,
// `tstride` in tot).  One block of 1024 threads per (64 channels, image): thread = (channel, one of sixteen chunk slices), two
// accumulators each; slices are combined through LDS in a fixed order, so the result does not depend on the launch.  (A single
// thread per (b, c) walking all chunks is a chain of nch dependent adds — 66 us at 256 chunks, more than the pass that produced
// them; four slices: 17 us.)
// FINISH (the statistics pass): NA == 2 (sums, sums of squares) and the thread that holds both totals of a (b, c) writes mean / rstd
// itself — instnorm_finish_kernel's arithmetic, one launch less per norm (15 per feature-encoder forward; ~5 us each at batch 1).
template <bool FINISH = false>
__global__ __launch_bounds__(1024) void chunk_reduce_kernel(const double* __restrict__ part, long long astride, int NA, int nch, int C,
                                                            double* __restrict__ tot, long long tstride, int HW = 0, float eps = 0.f,
                                                            float* __restrict__ mean = nullptr, float* __restrict__ rstd = nullptr) {
  __shared__ double red[16][64];
  const int cl = threadIdx.x & 63, b = blockIdx.y, c = blockIdx.x * 64 + cl, sl = threadIdx.x >> 6;
  double fin[2] = {0., 0.};
  for (int a = 0; a < NA; ++a) {
    double s0 = 0., s1 = 0.;
    if (c < C) {
      const double* p = part + a * astride + (long long)b * nch * C + c;
      int k = sl;
      for (; k + 16 < nch; k += 32) { s0 += p[(long long)k * C]; s1 += p[(long long)(k + 16) * C]; }
      if (k < nch) s0 += p[(long long)k * C];
    }
    red[sl][cl] = s0 + s1;
    __syncthreads();
    if (sl == 0 && c < C) {
      double s = 0.;
#pragma unroll
      for (int k = 0; k < 16; k += 4) s += (red[k][cl] + red[k + 1][cl]) + (red[k + 2][cl] + red[k + 3][cl]);
      if constexpr (FINISH) fin[a & 1] = s;
      else tot[a * tstride + (long long)b * C + c] = s;
    }
    __syncthreads();
  }
  if constexpr (FINISH) {
    if (sl == 0 && c < C) {
      const double m = fin[0] / (double)HW;
      double var = fin[1] / (double)HW - m * m;
      var = var > 0. ? var : 0.;
      mean[(long long)b * C + c] = (float)m;
      rstd[(long long)b * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
}

__global__ void instnorm_finish_kernel(const double* __restrict__ sum, const double* __restrict__ sq, int HW, float eps,
                                       float* __restrict__ mean, float* __restrict__ rstd, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // b*C + c
  if (i >= total) return;
  const double m = sum[i] / (double)HW;
  double var = sq[i] / (double)HW - m * m;
  var = var > 0. ? var : 0.;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = (x - mean) * rstd ; relu? ; [y = residual + y ; relu?]   (float4 per thread)
// TR / TO = __bf16: the residual rows / the output as the K8b convolutions hold activations (ABI 7): the statistics and the
// normalisation stay fp32 on the fp32 convolution output, only what the next convolution reads is 16 bits wide.
__device__ __forceinline__ f32x4 na_load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 na_load4(const __bf16* p) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 w = *reinterpret_cast<const u32x2*>(p);
  f32x4 v;
  v[0] = __builtin_bit_cast(float, w[0] << 16); v[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, w[1] << 16); v[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
  return v;
}
__device__ __forceinline__ void na_store4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void na_store4(__bf16* p, const f32x4 v) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
  *reinterpret_cast<bf16x4*>(p) = h;
}

template <typename TR = float, typename TO = float, typename TX = float>
__global__ __launch_bounds__(256) void norm_apply_kernel(const TX* __restrict__ x, int x_ld, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const TR* __restrict__ residual,
                                                         int res_ld, TO* __restrict__ out, int out_ld, long long M, int HW,
                                                         int C, int relu, int relu_after) {
  const int tpr = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long p = idx / tpr;
  if (p >= M) return;
  const int c4 = (int)(idx - p * tpr) * 4;
  const long long sidx = (p / HW) * C + c4;
  const f32x4 m = *reinterpret_cast<const f32x4*>(mean + sidx);
  const f32x4 r = *reinterpret_cast<const f32x4*>(rstd + sidx);
  f32x4 v = ld4f(x + p * x_ld + c4);
  v = (v - m) * r;
  if (relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
  }
  if (residual != nullptr) {
    v = na_load4(residual + p * res_ld + c4) + v;
    if (relu_after) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
    }
  }
  na_store4(out + p * out_ld + c4, v);
}


// ------------------------------------------------------------------------------------------------------------------
// Backward of  y = relu?((x - mean) * rstd)  with per-(image, channel) statistics over HW pixels (instance norm; batch norm
// in training mode is the same thing with the whole batch as ONE "image", B = 1, HW = all pixels):
//     g   = dy * (y > 0)                      (relu mask, from x_hat = (x - mean) * rstd)
//     s1  = sum_p g,   s2 = sum_p g * x_hat   (per (image, channel); also d(beta), d(gamma) of an affine batch norm)
//     dx  = rstd * (g - s1 / HW - x_hat * s2 / HW)
// Partial sums per (image, chunk) reduced in a fixed order (deterministic), then one elementwise pass.
// ------------------------------------------------------------------------------------------------------------------
// Accumulation and the final subtraction run in DOUBLE: when the incoming gradient is dominated by its per-channel mean (the
// situation of an encoder fed by the correlation backward), g - mean(g) - x_hat * mean(g x_hat) cancels to a small residual and
// fp32 sums lose 2-3 digits there (measured: 1e-2 relative on encoder weight gradients, for MIOpen's kernels as well); torch's
// CPU kernels accumulate float tensors in double for the same reason.  MI355X's fp64 vector rate makes this free next to the
// memory traffic of the pass.
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void norm_bwd_partial_kernel(const float* __restrict__ x, int x_ld, const float* __restrict__ dy,
                                                               int dy_ld, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, int C, int HW, int relu,
                                                               double* __restrict__ part1, double* __restrict__ part2) {
  __shared__ f64x4 red1[256], red2[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tpr = C >> 2, rpi = 256 / tpr;
  const int t = threadIdx.x;
  const int c4 = (t % tpr) * 4, rr = t / tpr;
  const bool active = rr < rpi;
  const int nch = (int)gridDim.x;
  const int rows = (HW + nch - 1) / nch;
  const int r0 = chunk * rows, r1 = min(HW, r0 + rows);
  f64x4 a1 = {0., 0., 0., 0.}, a2 = {0., 0., 0., 0.};
  if (active) {
    const f32x4 m = *reinterpret_cast<const f32x4*>(mean + (long long)b * C + c4);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + (long long)b * C + c4);
    const float* xb = x + (long long)b * HW * x_ld + c4;
    const float* gb = dy + (long long)b * HW * dy_ld + c4;
    for (int r = r0 + rr; r < r1; r += rpi) {
      const f32x4 xh = (*reinterpret_cast<const f32x4*>(xb + (long long)r * x_ld) - m) * rs;
      f32x4 g = *reinterpret_cast<const f32x4*>(gb + (long long)r * dy_ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (relu) g[e] = xh[e] > 0.f ? g[e] : 0.f;
        a1[e] += (double)g[e];
        a2[e] += (double)g[e] * (double)xh[e];
      }
    }
  }
  red1[t] = a1; red2[t] = a2;
  __syncthreads();
  if (t < tpr) {
    f64x4 s1 = red1[t], s2 = red2[t];
    for (int k = 1; k < rpi; ++k) { s1 += red1[t + k * tpr]; s2 += red2[t + k * tpr]; }
    *reinterpret_cast<f64x4*>(part1 + ((long long)b * nch + chunk) * C + c4) = s1;
    *reinterpret_cast<f64x4*>(part2 + ((long long)b * nch + chunk) * C + c4) = s2;
  }
}

// float copies of the reduced sums for the caller (d beta / d gamma of an affine batch norm)
__global__ void norm_bwd_copy_kernel(const double* __restrict__ d1, const double* __restrict__ d2, float* __restrict__ s1,
                                     float* __restrict__ s2, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (s1) s1[i] = (float)d1[i];
  if (s2) s2[i] = (float)d2[i];
}

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, int x_ld, const float* __restrict__ dy,
                                                             int dy_ld, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const double* __restrict__ s1,
                                                             const double* __restrict__ s2, float* __restrict__ dx, int dx_ld,
                                                             long long M, int HW, int C, int relu) {
  const int tpr = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long p = idx / tpr;
  if (p >= M) return;
  const int c4 = (int)(idx - p * tpr) * 4;
  const long long sidx = (p / HW) * C + c4;
  const f32x4 m = *reinterpret_cast<const f32x4*>(mean + sidx), rs = *reinterpret_cast<const f32x4*>(rstd + sidx);
  const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + p * x_ld + c4) - m) * rs;
  const f32x4 g = *reinterpret_cast<const f32x4*>(dy + p * dy_ld + c4);
  const double inv = 1.0 / (double)HW;
  f32x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ge = (relu && !(xh[e] > 0.f)) ? 0.f : g[e];
    out[e] = (float)((double)rs[e] * ((double)ge - s1[sidx + e] * inv - (double)xh[e] * (s2[sidx + e] * inv)));
  }
  *reinterpret_cast<f32x4*>(dx + p * dx_ld + c4) = out;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight (and bias) gradient of the stem: dW[ky][kx][ci][c] = sum_p dY[p][c] * img[ci][2yo + ky - 3][2xo + kx - 3].
// Same tiling as the forward kernel (32 consecutive output pixels of one output row per tile, thread = channel x 8 pixels,
// the 3 x 7 x 69 input patch staged in LDS), but PERSISTENT: a block walks tiles bid, bid + G, ... and keeps its 147 (+1 bias)
// accumulators per thread in registers; at the end the four pixel-group waves are summed through LDS and the block's
// partial goes to part[block][148][64]; a second kernel adds the partials in block order (deterministic).
// ------------------------------------------------------------------------------------------------------------------
constexpr int STEM_NW = 3 * STEM_K * STEM_K;   // 147

__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(const float* __restrict__ img, const float* __restrict__ dy,
                                                              int dy_ld, float* __restrict__ part, int H, int W, int Ho, int Wo,
                                                              int tiles_per_row, long long tiles, int cout) {
  __shared__ float sp[3][STEM_K][STEM_PW + 3];
  __shared__ float sred[4][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const bool cok = lane < cout;
  float acc[STEM_NW + 1];
#pragma unroll
  for (int i = 0; i <= STEM_NW; ++i) acc[i] = 0.f;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int orow = (int)(tile / tiles_per_row);
    const int xo0 = (int)(tile - (long long)orow * tiles_per_row) * STEM_TP;
    const int yo = orow % Ho, b = orow / Ho;
    const int iy0 = 2 * yo - STEM_R, ix0 = 2 * xo0 - STEM_R;
    const float* src = img + (long long)b * 3 * H * W;
    __syncthreads();   // the previous tile's patch is no longer read
    for (int e = threadIdx.x; e < 3 * STEM_K * STEM_PW; e += 256) {
      const int ci = e / (STEM_K * STEM_PW);
      const int r = e - ci * (STEM_K * STEM_PW);
      const int ky = r / STEM_PW, xx = r - ky * STEM_PW;
      const int gy = iy0 + ky, gx = ix0 + xx;
      float v = 0.f;
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = src[((long long)ci * H + gy) * W + gx];
      sp[ci][ky][xx] = v;
    }
    float d[STEM_NP];
#pragma unroll
    for (int p = 0; p < STEM_NP; ++p) {
      const int xo = xo0 + g * STEM_NP + p;
      d[p] = (cok && xo < Wo) ? dy[((long long)orow * Wo + xo) * dy_ld + lane] : 0.f;
      acc[STEM_NW] += d[p];
    }
    __syncthreads();
#pragma unroll
    for (int ky = 0; ky < STEM_K; ++ky)
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        float v[2 * STEM_NP + STEM_K - 2];
#pragma unroll
        for (int j = 0; j < 2 * STEM_NP + STEM_K - 2; ++j) v[j] = sp[ci][ky][2 * STEM_NP * g + j];
#pragma unroll
        for (int kx = 0; kx < STEM_K; ++kx) {
          float t = acc[(ky * STEM_K + kx) * 3 + ci];
#pragma unroll
          for (int p = 0; p < STEM_NP; ++p) t = fmaf(d[p], v[2 * p + kx], t);
          acc[(ky * STEM_K + kx) * 3 + ci] = t;
        }
      }
  }
  // sum the four pixel-group waves (fixed order) and write this block's partial: part[block][148][64]
  float* mine = part + (long long)blockIdx.x * (STEM_NW + 1) * 64;
#pragma unroll 1
  for (int i = 0; i <= STEM_NW; ++i) {
    float v = 0.f;
    // acc[i] with a run-time i would spill the array: select it with a compile-time unrolled scan instead
#pragma unroll
    for (int k = 0; k <= STEM_NW; ++k) v = (k == i) ? acc[k] : v;
    __syncthreads();
    sred[g][lane] = v;
    __syncthreads();
    if (g == 0) mine[i * 64 + lane] = (sred[0][lane] + sred[1][lane]) + (sred[2][lane] + sred[3][lane]);
  }
}

__global__ __launch_bounds__(256) void conv_stem_wgrad_reduce_kernel(const float* __restrict__ part, int nblocks, int cout,
                                                                     float* __restrict__ dw, float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (k, c) over 148 x 64
  if (i >= (STEM_NW + 1) * 64) return;
  const int k = i >> 6, c = i & 63;
  if (c >= cout) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += part[(long long)b * (STEM_NW + 1) * 64 + i];
  if (k < STEM_NW) dw[k * cout + c] = s;
  else if (db != nullptr) db[c] = s;
}

}  // namespace

extern "C" {

int pfk_conv_stem_f32(const float* img, const float* weight, const float* bias, float* out, int out_ld, int B,
                      int H, int W, int cout, int relu, pfk_stream_t stream) {
  if (!img || !weight || !out || B <= 0 || H <= 0 || W <= 0 || cout <= 0 || out_ld < cout) return PFK_ERR_BAD_ARG;
  // cout 32 / 64 (the reference's two encoders) with a 16-byte aligned weight: the MFMA kernel; anything else: the VALU kernel
  if ((cout == 32 || cout == 64) && pfk_aligned16(weight) && !g_stem_valu)
    return stem_launch<float>(img, weight, bias, out, out_ld, B, H, W, cout, relu, static_cast<hipStream_t>(stream));
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int tpr = (Wo + STEM_TP - 1) / STEM_TP;
  const long long tiles = (long long)B * Ho * tpr;
  if (tiles > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)tiles), dim3(256), 0, static_cast<hipStream_t>(stream), img, weight,
                     bias, out, out_ld, H, W, Ho, Wo, tpr, cout, relu);
  return pfk_launch_status();
}

int pfk_conv_stem_b16(const float* img, const float* weight, const float* bias, void* out_bf16, int out_ld, int B, int H, int W,
                      int cout, int relu, pfk_stream_t stream) {
  if (!img || !weight || !out_bf16 || B <= 0 || H <= 0 || W <= 0 || cout <= 0 || out_ld < cout) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(weight) || (reinterpret_cast<uintptr_t>(out_bf16) & 1u)) return PFK_ERR_ALIGNMENT;
  return stem_launch<__bf16>(img, weight, bias, static_cast<__bf16*>(out_bf16), out_ld, B, H, W, cout, relu, static_cast<hipStream_t>(stream));
}

int pfk_debug_set_stem_valu(int on) {
  if (!pfk_debug_knobs_enabled()) return PFK_ERR_DISABLED;
  g_stem_valu = on != 0;
  return PFK_OK;
}

long long pfk_instnorm_workspace_bytes(int B, int C) { return (2LL * B * in_chunks(B) * C + 2LL * B * C) * (long long)sizeof(double); }

}  // extern "C"

template <typename TX>
static int instnorm_stats_launch(const TX* x, int ld, int B, int HW, int C, float eps, float* mean, float* rstd,
                                 void* workspace, long long workspace_bytes, pfk_stream_t stream) {
  if (!x || !mean || !rstd || !workspace || B <= 0 || HW <= 0 || C <= 0 || ld < C) return PFK_ERR_BAD_ARG;
  if ((C & 3) || (ld & 3) || C > 1024 || (reinterpret_cast<uintptr_t>(x) & (sizeof(TX) * 4 - 1)) || !pfk_aligned16(mean) || !pfk_aligned16(rstd) ||
      (reinterpret_cast<uintptr_t>(workspace) & 31u))
    return PFK_ERR_ALIGNMENT;
  if (workspace_bytes < pfk_instnorm_workspace_bytes(B, C)) return PFK_ERR_BAD_ARG;
  const int nch = in_chunks(B);
  double* part = static_cast<double*>(workspace);            // 2 x [B][nch][C]: sums, sums of squares
  const long long astride = (long long)B * nch * C;
  double* sum = part + 2 * astride;                          // [B][C]
  double* sq = sum + (size_t)B * C;                          // [B][C]
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)nch, (unsigned)B), rgrid((unsigned)((C + 63) / 64), (unsigned)B);
  hipLaunchKernelGGL(instnorm_partial_kernel<TX>, grid, dim3(256), 0, st, x, ld, C, HW, part, astride);
  (void)sq;
  hipLaunchKernelGGL(chunk_reduce_kernel<true>, rgrid, dim3(1024), 0, st, part, astride, 2, nch, C, sum, (long long)B * C, HW, eps, mean, rstd);
  return pfk_launch_status();
}

extern "C" {

int pfk_instnorm_stats_f32(const float* x, int ld, int B, int HW, int C, float eps, float* mean, float* rstd,
                           void* workspace, long long workspace_bytes, pfk_stream_t stream) {
  return instnorm_stats_launch<float>(x, ld, B, HW, C, eps, mean, rstd, workspace, workspace_bytes, stream);
}

int pfk_instnorm_stats_b16(const void* x_bf16, int ld, int B, int HW, int C, float eps, float* mean, float* rstd,
                           void* workspace, long long workspace_bytes, pfk_stream_t stream) {
  return instnorm_stats_launch<__bf16>(static_cast<const __bf16*>(x_bf16), ld, B, HW, C, eps, mean, rstd, workspace, workspace_bytes, stream);
}

long long pfk_norm_bwd_workspace_bytes(int B, int C) { return ((long long)B * in_chunks(B) * C * 2 + (long long)B * C * 2) * (long long)sizeof(double); }

int pfk_norm_bwd_f32(const float* x, int x_ld, const float* dy, int dy_ld, const float* mean, const float* rstd, float* dx,
                     int dx_ld, float* sum_g, float* sum_gxhat, int B, int HW, int C, int relu, void* workspace,
                     long long workspace_bytes, pfk_stream_t stream) {
  if (!x || !dy || !mean || !rstd || !dx || !workspace || B <= 0 || HW <= 0 || C <= 0) return PFK_ERR_BAD_ARG;
  if (x_ld < C || dy_ld < C || dx_ld < C) return PFK_ERR_BAD_ARG;
  if ((C & 3) || (x_ld & 3) || (dy_ld & 3) || (dx_ld & 3) || C > 1024 || !pfk_aligned16(x) || !pfk_aligned16(dy) ||
      !pfk_aligned16(dx) || !pfk_aligned16(mean) || !pfk_aligned16(rstd) || (reinterpret_cast<uintptr_t>(workspace) & 31u))
    return PFK_ERR_ALIGNMENT;
  if (workspace_bytes < pfk_norm_bwd_workspace_bytes(B, C)) return PFK_ERR_BAD_ARG;
  double* p1 = static_cast<double*>(workspace);
  const int nch = in_chunks(B);
  double* p2 = p1 + (size_t)B * nch * C;
  double* d1 = p2 + (size_t)B * nch * C;
  double* d2 = d1 + (size_t)B * C;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(norm_bwd_partial_kernel, dim3((unsigned)nch, (unsigned)B), dim3(256), 0, st, x, x_ld, dy, dy_ld, mean, rstd, C, HW,
                     relu, p1, p2);
  const int total = B * C;
  // p1 and p2 are adjacent [B][nch][C] arrays, d1 and d2 adjacent [B][C] arrays: one reduction launch for both
  hipLaunchKernelGGL(chunk_reduce_kernel<false>, dim3((unsigned)((C + 63) / 64), (unsigned)B), dim3(1024), 0, st, p1, (long long)B * nch * C, 2, nch, C,
                     d1, (long long)B * C);
  if (sum_g || sum_gxhat)
    hipLaunchKernelGGL(norm_bwd_copy_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d1, d2, sum_g, sum_gxhat, total);
  const long long M = (long long)B * HW;
  const long long blocks = (M * (C >> 2) + 255) / 256;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, x_ld, dy, dy_ld, mean, rstd, d1, d2, dx,
                     dx_ld, M, HW, C, relu);
  return pfk_launch_status();
}

constexpr int STEM_WG_BLOCKS = 512;
long long pfk_conv_stem_wgrad_workspace_bytes(void) { return (long long)STEM_WG_BLOCKS * (STEM_NW + 1) * 64 * (long long)sizeof(float); }

int pfk_conv_stem_wgrad_f32(const float* img, const float* dy, int dy_ld, float* dw, float* db, int B, int H, int W, int cout,
                            void* workspace, long long workspace_bytes, pfk_stream_t stream) {
  if (!img || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0 || cout <= 0 || dy_ld < cout) return PFK_ERR_BAD_ARG;
  if (cout > 64) return PFK_ERR_UNSUPPORTED;
  if (workspace_bytes < pfk_conv_stem_wgrad_workspace_bytes()) return PFK_ERR_BAD_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int tpr = (Wo + STEM_TP - 1) / STEM_TP;
  const long long tiles = (long long)B * Ho * tpr;
  const int nblocks = (int)(tiles < STEM_WG_BLOCKS ? tiles : STEM_WG_BLOCKS);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(conv_stem_wgrad_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, img, dy, dy_ld, part, H, W, Ho, Wo, tpr,
                     tiles, cout);
  hipLaunchKernelGGL(conv_stem_wgrad_reduce_kernel, dim3(((STEM_NW + 1) * 64 + 255) / 256), dim3(256), 0, st, part, nblocks, cout, dw, db);
  return pfk_launch_status();
}

int pfk_norm_apply_f32(const float* x, int x_ld, const float* mean, const float* rstd, const float* residual,
                       int residual_ld, float* out, int out_ld, int B, int HW, int C, int relu,
                       int relu_after_residual, pfk_stream_t stream) {
  if (!x || !mean || !rstd || !out || B <= 0 || HW <= 0 || C <= 0 || x_ld < C || out_ld < C) return PFK_ERR_BAD_ARG;
  if ((C & 3) || (x_ld & 3) || (out_ld & 3) || !pfk_aligned16(x) || !pfk_aligned16(out) || !pfk_aligned16(mean) ||
      !pfk_aligned16(rstd))
    return PFK_ERR_ALIGNMENT;
  if (residual && (residual_ld < C || (residual_ld & 3) || !pfk_aligned16(residual))) return PFK_ERR_ALIGNMENT;
  const long long M = (long long)B * HW;
  const long long threads = M * (C >> 2);
  const long long blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((norm_apply_kernel<float, float>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, x_ld,
                     mean, rstd, residual, residual_ld, out, out_ld, M, HW, C, relu, relu_after_residual);
  return pfk_launch_status();
}

int pfk_norm_apply_b16(const void* x, int x_bf16, int x_ld, const float* mean, const float* rstd, const void* residual_bf16,
                       int residual_ld, void* out_bf16, int out_ld, int B, int HW, int C, int relu,
                       int relu_after_residual, pfk_stream_t stream) {
  if (!x || !mean || !rstd || !out_bf16 || B <= 0 || HW <= 0 || C <= 0 || x_ld < C || out_ld < C) return PFK_ERR_BAD_ARG;
  if ((C & 3) || (x_ld & 3) || (out_ld & 3) || (reinterpret_cast<uintptr_t>(x) & (x_bf16 ? 7u : 15u)) ||
      (reinterpret_cast<uintptr_t>(out_bf16) & 7u) || !pfk_aligned16(mean) || !pfk_aligned16(rstd))
    return PFK_ERR_ALIGNMENT;
  if (residual_bf16 && (residual_ld < C || (residual_ld & 3) || (reinterpret_cast<uintptr_t>(residual_bf16) & 7u))) return PFK_ERR_ALIGNMENT;
  const long long M = (long long)B * HW;
  const long long threads = M * (C >> 2);
  const long long blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const __bf16* r16 = static_cast<const __bf16*>(residual_bf16);
  __bf16* o16 = static_cast<__bf16*>(out_bf16);
  if (x_bf16)
    hipLaunchKernelGGL((norm_apply_kernel<__bf16, __bf16, __bf16>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const __bf16*>(x), x_ld,
                       mean, rstd, r16, residual_ld, o16, out_ld, M, HW, C, relu, relu_after_residual);
  else
    hipLaunchKernelGGL((norm_apply_kernel<__bf16, __bf16, float>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float*>(x), x_ld,
                       mean, rstd, r16, residual_ld, o16, out_ld, M, HW, C, relu, relu_after_residual);
  return pfk_launch_status();
}

}  // extern "C"
