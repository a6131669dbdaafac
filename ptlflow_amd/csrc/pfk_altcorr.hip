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

// Feature-map element: float, or bf16 (pfk_altcorr_forward_bf16: bf16 maps, exact widening, fp32 products and accumulation —
// half the gather bytes of an L2-bound kernel).  load4: four consecutive channels as floats.
typedef unsigned short bf16_t;
typedef unsigned int ac_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 widen4(ac_u32x2 u) {
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) { return widen4(*reinterpret_cast<const ac_u32x2*>(p)); }

template <class T>
__global__ __launch_bounds__(256) void altcorr_fwd_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
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
    a[i] = (live && c < C) ? load4(f1 + p * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int t0 = 0; t0 < ntaps; t0 += 4) {
    const int t = t0 + g;
    const int iy = t / n, ix = t - iy * n;
    const int yy = y0 + iy, xx = x0 + ix;
    const bool ok = live && t < ntaps && (unsigned)yy < (unsigned)H2 && (unsigned)xx < (unsigned)W2;
    float acc = 0.f;
    if (ok) {
      const T* row = f2 + ((b * H2 + yy) * (long long)W2 + xx) * C;
#pragma unroll
      for (int i = 0; i < MAX_C4; ++i) {
        const int c = cl * 4 + i * 64;
        if (c < C) {
          const f32x4 v = load4(row + c);
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

// ------------------------------------------------------------------------------------------------------------------
// Window-sharing formulation on the fp32 matrix cores (round 3).  Neighbouring pixels look at nearly the same part of fmap2:
// an 8 x TPY patch of fmap1 pixels whose flow is locally smooth has all its (2r+2)^2 windows inside one bounding box of
// roughly (8 + 2r + 1 + spread) x (TPY + 2r + 1 + spread) points.  The block computes the box once,
//     S[pixel i][box point j] = <fmap1[i, :], fmap2[j, :]>          a [8 TPY x C] x [C x NP] GEMM, v_mfma_f32_32x32x2_f32,
// with the box rows staged through LDS exactly once per block (the per-pixel kernel above re-reads every fmap2 row once per pixel
// that touches it: ~100 KB of L2 gathers per pixel; here 8-16 KB), and every pixel then picks its own (2r+2)^2 taps out of S and
// combines them with the same bilinear formula.  The box is clipped to the map (points outside are zeros anyway); pixels with
// non-finite / absurd coordinates stay outside the box (all their taps are zero, as in the per-pixel kernel); a patch whose
// clipped box has more than WS_NP_MAX points (strongly divergent flow) is left to altcorr_fwd_overflow_kernel, launched right behind:
// the per-pixel algorithm on just those patches (both kernels take the decision with the same patch_box()).
// K step = 16 channels (64-byte LDS rows, 16-byte chunk index XOR (row >> 2) & 3: conflict-free ds_write_b128 / ds_read_b128),
// two LDS stages, registers prefetch the next step; four waves split the box points (column blocks of 32, interleaved).
// ------------------------------------------------------------------------------------------------------------------
constexpr int WS_NP_MAX = 512;                 // box points per block (16 column blocks of 32)
constexpr int WS_BK = 16;                      // channels per K step
constexpr int WS_ROWB = WS_BK * 4;             // 64-byte LDS rows
typedef unsigned int ws_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned WS_OOB = 0x80000000u;

// per-pixel algorithm of altcorr_fwd_kernel for one pixel per wave (wave-local: no workgroup barrier)
template <class T>
__device__ __forceinline__ void altcorr_pixel(const T* __restrict__ f1, const T* __restrict__ f2, float* __restrict__ out,
                                              bool live, long long p, long long b, int pix, float x, float y, long long hw, int H2,
                                              int W2, int C, int r, float* s_tap_w, int lane) {
  const int g = lane >> 4, cl = lane & 15;
  const int rd = 2 * r + 1, n = rd + 1, ntaps = n * n;
  const float fx = floorf(x), fy = floorf(y);
  const float dx = x - fx, dy = y - fy;
  const int x0 = (fabsf(fx) < 1.0e9f) ? (int)fx - r : -(1 << 30);
  const int y0 = (fabsf(fy) < 1.0e9f) ? (int)fy - r : -(1 << 30);
  f32x4 a[MAX_C4];
#pragma unroll
  for (int i = 0; i < MAX_C4; ++i) {
    const int c = cl * 4 + i * 64;
    a[i] = (live && c < C) ? load4(f1 + p * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int t0 = 0; t0 < ntaps; t0 += 4) {
    const int t = t0 + g;
    const int iy = t / n, ix = t - iy * n;
    const int yy = y0 + iy, xx = x0 + ix;
    const bool ok = live && t < ntaps && (unsigned)yy < (unsigned)H2 && (unsigned)xx < (unsigned)W2;
    float acc = 0.f;
    if (ok) {
      const T* row = f2 + ((b * H2 + yy) * (long long)W2 + xx) * C;
#pragma unroll
      for (int i = 0; i < MAX_C4; ++i) {
        const int c = cl * 4 + i * 64;
        if (c < C) {
          const f32x4 v = load4(row + c);
          acc = fmaf(a[i].x, v.x, fmaf(a[i].y, v.y, fmaf(a[i].z, v.z, fmaf(a[i].w, v.w, acc))));
        }
      }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (cl == 0 && t < ntaps) s_tap_w[t] = acc;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // s_tap_w is this wave's own: in-order LDS, no workgroup barrier
  if (live) {
    const float ex = 1.0f - dx, ey = 1.0f - dy;
    float* o = out + b * rd * rd * hw + pix;
    for (int k = lane; k < rd * rd; k += 64) {
      const int ox = k / rd, oy = k - ox * rd;
      const float s00 = s_tap_w[oy * n + ox], s01 = s_tap_w[oy * n + ox + 1];
      const float s10 = s_tap_w[(oy + 1) * n + ox], s11 = s_tap_w[(oy + 1) * n + ox + 1];
      o[k * hw] = s00 * ey * ex + s01 * ey * dx + s10 * dy * ex + s11 * dy * dx;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next pixel's taps overwrite them
}

// Coordinates of the 8 x TPY patch at (py0, px0) — lane l < 8 TPY is pixel (l >> 3, l & 7) — and the bounding box {x0, y0, w, h} of
// the windows of its finite pixels, clipped to the map (a window entirely outside contributes nothing).  One full wave; every
// lane returns the box.  The window-sharing kernel and the overflow kernel both decide with THIS function.
constexpr int WS_FAR = -(1 << 30);
template <int TPY>
__device__ __forceinline__ void patch_box(const float* __restrict__ coords, long long b, int py0, int px0, int H1, int W1, int H2,
                                          int W2, int r, int lane, float& x, float& y, int& x0, int& y0, int (&box)[4]) {
  constexpr int TP = 8 * TPY;
  const int n = 2 * r + 2;
  const int yy = py0 + (lane >> 3), xx = px0 + (lane & 7);
  const bool inside = lane < TP && yy < H1 && xx < W1;
  x = 0.f; y = 0.f; x0 = WS_FAR; y0 = WS_FAR;
  if (inside) {
    const long long p = b * ((long long)H1 * W1) + (long long)yy * W1 + xx;
    x = coords[p * 2 + 0];
    y = coords[p * 2 + 1];
    const float fx = floorf(x), fy = floorf(y);
    x0 = (fabsf(fx) < 1.0e9f) ? (int)fx - r : WS_FAR;
    y0 = (fabsf(fy) < 1.0e9f) ? (int)fy - r : WS_FAR;
  }
  const bool use = inside && x0 != WS_FAR && y0 != WS_FAR && x0 < W2 && y0 < H2 && x0 + n > 0 && y0 + n > 0;
  int lo_x = use ? max(x0, 0) : (1 << 30), hi_x = use ? min(x0 + n, W2) : -(1 << 30);
  int lo_y = use ? max(y0, 0) : (1 << 30), hi_y = use ? min(y0 + n, H2) : -(1 << 30);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo_x = min(lo_x, __shfl_xor(lo_x, off, 64)); hi_x = max(hi_x, __shfl_xor(hi_x, off, 64));
    lo_y = min(lo_y, __shfl_xor(lo_y, off, 64)); hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
  }
  const bool any = hi_x > lo_x && hi_y > lo_y;
  box[0] = any ? lo_x : 0; box[1] = any ? lo_y : 0; box[2] = any ? hi_x - lo_x : 0; box[3] = any ? hi_y - lo_y : 0;
}

// The patch takes the shared GEMM when its box fits the LDS budget.  (A cost rule on top — GEMM only while TP x NP <= 6 x the taps
// the per-pixel algorithm would compute — was measured: it trims the all-noise case, 400 -> 360 us, and costs smooth fields
// 72 -> 88 us at 110x256, because a rejected border patch then runs alone in the second launch.  Not kept.)
__device__ __forceinline__ bool patch_takes_gemm(const int (&box)[4]) { return (long long)box[2] * box[3] <= WS_NP_MAX; }

// ---- launch-wide gate (round 4) --------------------------------------------------------------------------------------------------
// The window-sharing GEMM pays when MANY patches take it.  When almost every box overflows (iid-noise coordinate fields), the few
// border patches that still fit ran their ~30 us GEMM blocks on an otherwise idle chip in FRONT of the per-pixel pass: 400 us
// against the per-pixel kernel's 333 (0.83x).  altcorr_plan_kernel counts the patches that would take the GEMM (8 patches per
// block — two per wave, one coordinate load deep — per-block counts: no atomics, no zero-fill, deterministic); both passes read
// the counts and, when fewer than 5/8 of the patches qualify, the GEMM pass leaves and the per-pixel pass takes EVERY patch.
// Measured (MI355X, 55x128, batch 8, C = 256, r = 4; gpurun_out/r4c_lookup.log, r4d): every patch qualifies 129 us vs 272
// per-pixel; ~3/4 qualify (smooth field of +-4 px per 8 px) 186 vs 276; half qualify 346 vs 313 — the two passes run one after
// the other and each leaves the chip half empty — so the gate sits between: 5/8.
constexpr int PLAN_PATCHES = 8;       // patches per plan block
template <int TPY>
__global__ __launch_bounds__(256) void altcorr_plan_kernel(const float* __restrict__ coords, int* __restrict__ counts, int B, int H1,
                                                           int W1, int H2, int W2, int r, int tiles_x, int tiles_y) {
  static_assert(TPY == 4, "two 8x4 patches per wave");
  __shared__ int s_cnt[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
  const int n = 2 * r + 2;
  const long long npatch = (long long)B * tiles_x * tiles_y;
  int mine = 0;
  {
    const long long tile = (long long)blockIdx.x * PLAN_PATCHES + wid * 2 + half;
    const bool tl = tile < npatch;
    const int txi = (int)(tile % tiles_x), tyi = (int)((tile / tiles_x) % tiles_y);
    const long long b = tile / ((long long)tiles_x * tiles_y);
    const int yy = tyi * TPY + (hl >> 3), xx = txi * 8 + (hl & 7);
    const bool inside = tl && yy < H1 && xx < W1;
    int x0 = WS_FAR, y0 = WS_FAR;
    if (inside) {
      const long long p = b * ((long long)H1 * W1) + (long long)yy * W1 + xx;
      const float fx = floorf(coords[p * 2 + 0]), fy = floorf(coords[p * 2 + 1]);
      x0 = (fabsf(fx) < 1.0e9f) ? (int)fx - r : WS_FAR;
      y0 = (fabsf(fy) < 1.0e9f) ? (int)fy - r : WS_FAR;
    }
    const bool use = inside && x0 != WS_FAR && y0 != WS_FAR && x0 < W2 && y0 < H2 && x0 + n > 0 && y0 + n > 0;   // as patch_box()
    int lo_x = use ? max(x0, 0) : (1 << 30), hi_x = use ? min(x0 + n, W2) : -(1 << 30);
    int lo_y = use ? max(y0, 0) : (1 << 30), hi_y = use ? min(y0 + n, H2) : -(1 << 30);
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {       // within the 32-lane half
      lo_x = min(lo_x, __shfl_xor(lo_x, off, 64)); hi_x = max(hi_x, __shfl_xor(hi_x, off, 64));
      lo_y = min(lo_y, __shfl_xor(lo_y, off, 64)); hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
    }
    const bool any = hi_x > lo_x && hi_y > lo_y;
    const int box[4] = {0, 0, any ? hi_x - lo_x : 0, any ? hi_y - lo_y : 0};
    if (tl && hl == 0 && patch_takes_gemm(box)) ++mine;
  }
  mine += __shfl_xor(mine, 32, 64);                 // the two halves' leaders
  if (lane == 0) s_cnt[wid] = mine;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// one wave: do enough patches take the GEMM for the window-sharing pass to run at all?  (counts == nullptr: always)
__device__ __forceinline__ bool gemm_pass_enabled(const int* __restrict__ counts, int nplan, long long npatch, int lane) {
  if (counts == nullptr) return true;
  int s = 0;
  for (int i = lane; i < nplan; i += 64) s += counts[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  return 8LL * s >= 5LL * npatch;
}

// The patches the window-sharing kernel skipped: the per-pixel algorithm, one pixel per wave, blocks in the per-pixel kernel's own
// row-major order (four consecutive pixels of an image row — they always lie in one patch); a block whose patch took the GEMM
// leaves after the box test (one per block, wave 0).
template <int TPY, class T>
__global__ __launch_bounds__(256) void altcorr_fwd_overflow_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                                   const float* __restrict__ coords, float* __restrict__ out,
                                                                   int H1, int W1, int H2, int W2, int C, int r, int groups_x,
                                                                   int force, const int* __restrict__ counts, int nplan,
                                                                   long long npatch) {
  __shared__ float s_tap[4][104];
  __shared__ float s_xy[4][2];
  __shared__ int s_over;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int gx = blockIdx.x % groups_x;                 // group of four pixels in its row
  const long long rowid = blockIdx.x / groups_x;        // b * H1 + y
  const long long b = rowid / H1;
  const int yy = (int)(rowid - b * H1);
  const int xg = gx * 4;
  const int px0 = xg & ~7, py0 = yy - yy % TPY;
  if (wid == 0) {
    float x, y;
    int x0, y0, box[4];
    patch_box<TPY>(coords, b, py0, px0, H1, W1, H2, W2, r, lane, x, y, x0, y0, box);
    const int q0 = (yy - py0) * 8 + (xg - px0);         // first of this block's pixels inside the patch
    if (lane >= q0 && lane < q0 + 4) { s_xy[lane - q0][0] = x; s_xy[lane - q0][1] = y; }
    const bool gemm_on = gemm_pass_enabled(counts, nplan, npatch, lane);
    if (lane == 0) s_over = force || !gemm_on || !patch_takes_gemm(box);
  }
  __syncthreads();
  if (!s_over) return;
  const long long hw = (long long)H1 * W1;
  const int xx = xg + wid;
  const bool live = xx < W1;
  const int pix = yy * W1 + xx;
  altcorr_pixel(f1, f2, out, live, b * hw + pix, b, pix, s_xy[wid][0], s_xy[wid][1], hw, H2, W2, C, r, s_tap[wid], lane);
}

template <int TPY, class T>
__global__ __launch_bounds__(256, 2) void altcorr_fwd_ws_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                                const float* __restrict__ coords, float* __restrict__ out,
                                                                int B, int H1, int W1, int H2, int W2, int C, int r,
                                                                int tiles_x, int tiles_y, const int* __restrict__ counts, int nplan) {
  constexpr int TP = 8 * TPY, MT = TP / 32;
  static_assert(TP == 32 || TP == 64, "patches of 8 x 4 or 8 x 8 pixels");
  constexpr int STAGE = (TP + WS_NP_MAX) * WS_ROWB;        // bytes per stage: A rows then B rows
  static_assert(2 * STAGE >= 32 * WS_NP_MAX * 4, "the S buffer of one 32-pixel half aliases the two stages");
  extern __shared__ __attribute__((aligned(16))) char ws_smem[];   // [2][STAGE]
  __shared__ float s_x[TP], s_y[TP];        // target coordinates
  __shared__ int s_x0[TP], s_y0[TP];        // window origin (floor - r), or the far sentinel
  __shared__ int s_box[5];                  // bx0, by0, bw, bh, takes-the-GEMM flag

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int tile = blockIdx.x;
  const int txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y;
  const long long b = tile / (tiles_x * tiles_y);
  const int px0 = txi * 8, py0 = tyi * TPY;
  const long long hw = (long long)H1 * W1;
  const int rd = 2 * r + 1;
  constexpr int FAR = WS_FAR;

  // ---- per-pixel coordinates and the bounding box of the patch's windows (wave 0) -------------------------------------------------
  if (wid == 0) {
    float x, y;
    int x0, y0, box[4];
    patch_box<TPY>(coords, b, py0, px0, H1, W1, H2, W2, r, lane, x, y, x0, y0, box);
    if (lane < TP) { s_x[lane] = x; s_y[lane] = y; s_x0[lane] = x0; s_y0[lane] = y0; }
    if (lane < 4) s_box[lane] = box[lane];
    const bool gemm_on = gemm_pass_enabled(counts, nplan, (long long)B * tiles_x * tiles_y, lane);
    if (lane == 4) s_box[4] = gemm_on && patch_takes_gemm(box);
  }
  __syncthreads();
  const int bx0 = s_box[0], by0 = s_box[1], bw = s_box[2], bh = s_box[3];
  const long long np_ll = (long long)bw * bh;
  if (!s_box[4]) return;              // strongly divergent flow: altcorr_fwd_overflow_kernel (launched behind this one) takes the patch
  const int NP = (int)np_ll;
  const int NCB = (NP + 31) >> 5;                       // column blocks of 32 box points

  f32x16 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][q][e] = 0.f;

  if (NP > 0) {
    // ---- staging plan: thread t owns 16-byte chunk (t & 3) of A row t >> 2 (TP == 64) and of B rows (t >> 2) + 64 i -------------
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(f1), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(f2), 0, 0x7fffffff, 0x00020000);
    constexpr int ES = (int)sizeof(T);                      // bytes per feature-map element
    const int ch = t & 3, r0 = t >> 2;
    const unsigned sw = (unsigned)(((ch ^ ((r0 >> 2) & 3)) << 4));      // swizzled chunk byte offset (rows r0 + 64 i share the key)
    unsigned aoff = WS_OOB;
    if (r0 < TP) {
      const int yy = py0 + (r0 >> 3), xx = px0 + (r0 & 7);
      if (yy < H1 && xx < W1) aoff = (unsigned)(((b * hw + (long long)yy * W1 + xx) * C + ch * 4) * ES);
    }
    unsigned boff[WS_NP_MAX / 64];
#pragma unroll
    for (int i = 0; i < WS_NP_MAX / 64; ++i) {
      const int j = r0 + 64 * i;
      boff[i] = WS_OOB;
      if (j < NP) {
        const int jy = j / bw, jx = j - jy * bw;           // inside the clipped box => inside the map
        boff[i] = (unsigned)((((b * H2 + by0 + jy) * (long long)W2 + bx0 + jx) * C + ch * 4) * ES);
      }
    }
    const int nrow_b = (NCB * 32 + 63) >> 6;               // B passes of 64 rows that hold live column blocks
    ws_u32x4 ra, rb[WS_NP_MAX / 64];
    auto load = [&](int k0) {
      const bool cok = k0 + ch * 4 < C;                    // C is a multiple of 4
      // four channels per thread: one 16-byte (fp32) or 8-byte (bf16, widened on the way into LDS: the GEMM stays fp32) load
      if constexpr (ES == 4) {
        ra = __builtin_amdgcn_raw_buffer_load_b128(rs1, (cok && r0 < TP) ? aoff : WS_OOB, k0 * 4, 0);
#pragma unroll
        for (int i = 0; i < WS_NP_MAX / 64; ++i)
          if (i < nrow_b) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs2, cok ? boff[i] : WS_OOB, k0 * 4, 0);
      } else {
        ra = __builtin_bit_cast(ws_u32x4, widen4(__builtin_amdgcn_raw_buffer_load_b64(rs1, (cok && r0 < TP) ? aoff : WS_OOB, k0 * 2, 0)));
#pragma unroll
        for (int i = 0; i < WS_NP_MAX / 64; ++i)
          if (i < nrow_b) rb[i] = __builtin_bit_cast(ws_u32x4, widen4(__builtin_amdgcn_raw_buffer_load_b64(rs2, cok ? boff[i] : WS_OOB, k0 * 2, 0)));
      }
    };
    auto store = [&](char* stage) {
      if (r0 < TP) *reinterpret_cast<ws_u32x4*>(stage + r0 * WS_ROWB + sw) = ra;
#pragma unroll
      for (int i = 0; i < WS_NP_MAX / 64; ++i)
        if (i < nrow_b) *reinterpret_cast<ws_u32x4*>(stage + (TP + r0 + 64 * i) * WS_ROWB + sw) = rb[i];
    };
    // fragment addressing: lane l feeds row l & 31 and the k half l >> 5; per 8 channels one ds_read_b128 (chunk 2 q + (l >> 5))
    const int frow = lane & 31, hl = lane >> 5;
    const int key = (frow >> 2) & 3;                       // rows 32 m + frow share it
    const int ko0 = ((0 + hl) ^ key) << 4, ko1 = ((2 + hl) ^ key) << 4;

    const int nsteps = (C + WS_BK - 1) / WS_BK;
    char* st0 = ws_smem;
    char* st1 = ws_smem + STAGE;
    load(0);
    store(st0);
    __syncthreads();
    for (int k = 0; k < nsteps; ++k) {
      char* cur = (k & 1) ? st1 : st0;
      char* nxt = (k & 1) ? st0 : st1;
      if (k + 1 < nsteps) load((k + 1) * WS_BK);
      f32x4 fa[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        fa[mt][0] = *reinterpret_cast<const f32x4*>(cur + (mt * 32 + frow) * WS_ROWB + ko0);
        fa[mt][1] = *reinterpret_cast<const f32x4*>(cur + (mt * 32 + frow) * WS_ROWB + ko1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cb = wid + 4 * q;                        // wave-uniform
        if (cb < NCB) {
          const char* brow = cur + (TP + cb * 32 + frow) * WS_ROWB;
          const f32x4 fb0 = *reinterpret_cast<const f32x4*>(brow + ko0);
          const f32x4 fb1 = *reinterpret_cast<const f32x4*>(brow + ko1);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][0][e], fb0[e], acc[mt][q], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][1][e], fb1[e], acc[mt][q], 0, 0, 0);
          }
        }
      }
      if (k + 1 < nsteps) store(nxt);
      __syncthreads();
    }
  }

  // ---- S of one 32-pixel half through LDS, then every pixel's (2r+1)^2 cells from its own taps ------------------------------------
  float* S = reinterpret_cast<float*>(ws_smem);            // [32][SLD]
  const int SLD = NCB * 32 + 1;                            // odd row stride: the tap gathers below spread over the banks
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (NP > 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cb = wid + 4 * q;
        if (cb < NCB) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            S[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * SLD + cb * 32 + (lane & 31)] = acc[mt][q][e];
        }
      }
    }
    __syncthreads();
    for (int idx = t; idx < 32 * rd * rd; idx += 256) {
      const int kc = idx >> 5, i = idx & 31;               // pixel fastest: a row of the patch is one 32-byte run of an output plane
      const int q = mt * 32 + i;
      const int yy = py0 + (q >> 3), xx = px0 + (q & 7);
      if (yy >= H1 || xx >= W1) continue;
      const int ox = kc / rd, oy = kc - ox * rd;
      const float x = s_x[q], y = s_y[q];
      const float dx = x - floorf(x), dy = y - floorf(y), ex = 1.0f - dx, ey = 1.0f - dy;
      const int wx = s_x0[q] - bx0 + ox, wy = s_y0[q] - by0 + oy;       // tap (oy, ox) in box coordinates (garbage for FAR: filtered)
      const bool far = s_x0[q] == FAR || s_y0[q] == FAR;
      const bool cx0 = !far && (unsigned)wx < (unsigned)bw, cx1 = !far && (unsigned)(wx + 1) < (unsigned)bw;
      const bool cy0 = !far && (unsigned)wy < (unsigned)bh, cy1 = !far && (unsigned)(wy + 1) < (unsigned)bh;
      const float* Si = S + i * SLD;
      const float s00 = (cy0 && cx0) ? Si[wy * bw + wx] : 0.f, s01 = (cy0 && cx1) ? Si[wy * bw + wx + 1] : 0.f;
      const float s10 = (cy1 && cx0) ? Si[(wy + 1) * bw + wx] : 0.f, s11 = (cy1 && cx1) ? Si[(wy + 1) * bw + wx + 1] : 0.f;
      out[(b * rd * rd + kc) * hw + (long long)yy * W1 + xx] = s00 * ey * ex + s01 * ey * dx + s10 * dy * ex + s11 * dy * dx;
    }
    __syncthreads();
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

int g_altcorr_mode = 0;   // tuning / test knob (pfk_debug_set_altcorr): 0 heuristic, 1 per-pixel kernel, 2 / 3 window-sharing 8x4 / 8x8, 4 = the overflow kernel alone on every patch (timing)

namespace {
// forward for float or bf16 feature maps (coords / out fp32 either way)
template <class T>
int altcorr_forward_launch(const T* fmap1, const T* fmap2, const float* coords, float* out, int B, int H1,
                           int W1, int H2, int W2, int C, int radius, void* workspace, pfk_stream_t stream) {
  if (!fmap1 || !fmap2 || !coords || !out) return PFK_ERR_BAD_ARG;
  if (B <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0) return PFK_ERR_BAD_ARG;
  if (radius < 1 || radius > 4) return PFK_ERR_UNSUPPORTED;
  if ((C & 3) || !pfk_aligned16(fmap1) || !pfk_aligned16(fmap2)) return PFK_ERR_ALIGNMENT;
  if (C > 64 * MAX_C4) return PFK_ERR_UNSUPPORTED;
  const long long M = (long long)B * H1 * W1;
  const long long blocks = (M + 3) / 4;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // Window-sharing MFMA kernel on 8 x 4 patches once the patch grid fills the chip (a block lives ~30 us), the per-pixel kernel below that and for maps whose byte offsets do not fit the 32-bit buffer addressing of the staging
  // loads.  MI355X, C = 256, r = 4 (scripts/lookup_bench.py, gpurun_out/r3_altcorr.log): 55x128 batch 8 with a smooth flow field
  // 270 -> 129 us (2.1x; 22 TFLOP/s of useful window work), 110x256 (1/4 resolution) 141 -> 73 us; a field with +-4 px of
  // low-frequency variation per 8 px 271 -> 186 us; iid noise of sigma 6 px on every pixel (no two windows share anything: every box
  // overflows and the patch is handed to altcorr_fwd_overflow_kernel) 333 -> 400 us (the overflow kernel alone, in the per-pixel
  // kernel's row-major block order, 333; the rest is the border patches' GEMMs in front of it).  8 x 8 patches never beat 8 x 4.
  const int tx = (W1 + 7) / 8;
  const long long t8 = (long long)B * ((H1 + 7) / 8) * tx, t4 = (long long)B * ((H1 + 3) / 4) * tx;
  const bool fits32 = (long long)B * H1 * W1 * C * (long long)sizeof(T) < 0x7fffffffLL && (long long)B * H2 * W2 * C * (long long)sizeof(T) < 0x7fffffffLL;
  const int mode = g_altcorr_mode;
  if (fits32 && mode != 1 && (mode >= 2 || t4 >= 256)) {
    const bool big = mode == 3;
    const int gxn = (W1 + 3) / 4;
    const long long ogrid = (long long)B * H1 * gxn;
    if (ogrid > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
    // launch-wide gate (8 x 4 patches, when the caller lent a workspace): per-block counts of the patches that take the GEMM
    int* counts = (!big && mode == 0) ? static_cast<int*>(workspace) : nullptr;
    const int nplan = (int)((t4 + PLAN_PATCHES - 1) / PLAN_PATCHES);
    if (counts)
      hipLaunchKernelGGL((altcorr_plan_kernel<4>), dim3((unsigned)nplan), dim3(256), 0, st, coords, counts, B, H1, W1, H2, W2, radius, tx,
                         (H1 + 3) / 4);
    if (big) {
      constexpr size_t smem = 2 * (64 + WS_NP_MAX) * WS_ROWB;
      static pfk_device_once once;
      once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(altcorr_fwd_ws_kernel<8, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
      hipLaunchKernelGGL((altcorr_fwd_ws_kernel<8, T>), dim3((unsigned)t8), dim3(256), smem, st, fmap1, fmap2, coords, out, B, H1, W1, H2, W2, C,
                         radius, tx, (H1 + 7) / 8, (const int*)nullptr, 0);
      hipLaunchKernelGGL((altcorr_fwd_overflow_kernel<8, T>), dim3((unsigned)ogrid), dim3(256), 0, st, fmap1, fmap2, coords, out, H1, W1, H2, W2,
                         C, radius, gxn, 0, (const int*)nullptr, 0, t8);
    } else {
      constexpr size_t smem = 2 * (32 + WS_NP_MAX) * WS_ROWB;
      static pfk_device_once once;
      once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(altcorr_fwd_ws_kernel<4, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
      if (mode != 4)
        hipLaunchKernelGGL((altcorr_fwd_ws_kernel<4, T>), dim3((unsigned)t4), dim3(256), smem, st, fmap1, fmap2, coords, out, B, H1, W1, H2, W2,
                           C, radius, tx, (H1 + 3) / 4, (const int*)counts, nplan);
      hipLaunchKernelGGL((altcorr_fwd_overflow_kernel<4, T>), dim3((unsigned)ogrid), dim3(256), 0, st, fmap1, fmap2, coords, out, H1, W1, H2, W2,
                         C, radius, gxn, mode == 4, (const int*)counts, nplan, t4);
    }
    return pfk_launch_status();
  }
  hipLaunchKernelGGL((altcorr_fwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, fmap1,
                     fmap2, coords, out, M, H1, W1, H2, W2, C, radius);
  return pfk_launch_status();
}
}  // namespace

extern "C" {

int pfk_debug_set_altcorr(int mode) { if (!pfk_debug_knobs_enabled()) return PFK_ERR_DISABLED; g_altcorr_mode = mode; return PFK_OK; }

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


long long pfk_altcorr_workspace_bytes(int B, int H1, int W1) {
  if (B <= 0 || H1 <= 0 || W1 <= 0) return 0;
  const long long t4 = (long long)B * ((H1 + 3) / 4) * ((W1 + 7) / 8);
  return ((t4 + PLAN_PATCHES - 1) / PLAN_PATCHES) * (long long)sizeof(int);
}

int pfk_altcorr_forward_f32(const float* fmap1, const float* fmap2, const float* coords, float* out, int B, int H1,
                            int W1, int H2, int W2, int C, int radius, void* workspace, pfk_stream_t stream) {
  return altcorr_forward_launch<float>(fmap1, fmap2, coords, out, B, H1, W1, H2, W2, C, radius, workspace, stream);
}

int pfk_altcorr_forward_bf16(const void* fmap1_bf16, const void* fmap2_bf16, const float* coords, float* out, int B, int H1,
                             int W1, int H2, int W2, int C, int radius, void* workspace, pfk_stream_t stream) {
  return altcorr_forward_launch<bf16_t>(static_cast<const bf16_t*>(fmap1_bf16), static_cast<const bf16_t*>(fmap2_bf16), coords, out,
                                        B, H1, W1, H2, W2, C, radius, workspace, stream);
}

}  // extern "C"
