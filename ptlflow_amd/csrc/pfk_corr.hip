// K2 (pyramid pooling) and K3 (radius-r bilinear pyramid lookup) for gfx950.
//
// This file is compiled with -ffp-contract=off: the coordinate arithmetic must round after
// every operation exactly like the reference's chain of separate torch ops
// (ptlflow/models/raft/utils.py:71-72 then grid_sample's un-normalise), otherwise floor() of the
// sample position — the tap index — differs from the reference for ~7 % of integer coordinates
// (SURVEY.md finding 3).  The only fused operations are the three explicit fmaf() of the
// bilinear accumulation, which reproduce torch's CPU grid_sample bit for bit.
#include "pfk_common.h"
#include "pfk_lookup.h"

namespace {

// ------------------------------------------------------------------------------------------
// K3 lookup.  One workgroup per PIX consecutive source pixels, one 64-lane wave per pyramid level.
//
// All (2r+1)^2 samples of a level share (up to fp32 rounding of the round trip) one fractional
// offset, so they touch a (2r+2)^2 patch of that pixel's [h_l][w_l] correlation map.  The wave
// stages a 12x12 patch (one extra ring absorbs the +-1 index wobble of the round trip, so the
// staged window always contains every tap the reference would read) into LDS with row-contiguous
// loads — 12 consecutive floats per row, 3 loads per lane — zero-filling everything outside
// the map (that *is* grid_sample's zero padding), then each lane produces samples k = lane,
// lane+64 from LDS with its own per-tap floor()/weights.  Algorithmic traffic: (2r+2)^2 reads +
// (2r+1)^2 writes per pixel per level (20.4 MB / iteration at 55x128, L=4, r=4).
// ------------------------------------------------------------------------------------------

struct LookupArgs {
  const void* lv[PFK_MAX_LEVELS];   // float or __bf16 maps (template parameter T of the kernel)
  int lh[PFK_MAX_LEVELS];
  int lw[PFK_MAX_LEVELS];
  int L, r, B, h, w;
  const float* coords;
  void* out;
  int out_ld, out_bf16;
};

__device__ __forceinline__ void store_sample(const LookupArgs& a, size_t idx, float t) {
  if (a.out_bf16) static_cast<__bf16*>(a.out)[idx] = (__bf16)t;
  else static_cast<float*>(a.out)[idx] = t;
}

// Blocked volume layout (round 5): a level's [H][W] map per source pixel is stored as ceil(H/4) x ceil(W/8) tiles of 4 rows x 8
// columns, 32 consecutive elements each — one 128-byte line per fp32 tile.  The 12 x 12 window a lookup stages then touches
// (12 + dy) / 4 x (12 + dx) / 8 ~ 3.75 x 2.4 = 9 lines instead of the 12 rows x 1.34 = 16 lines of the row-major map (every row
// of the window lies in its own line there): the HBM fetch per lookup shrinks accordingly.  Element (y, x):
__device__ __forceinline__ int blocked_index(int y, int x, int tw) {
  return ((((y >> 2) * tw + (x >> 3)) << 5) | ((y & 3) << 3) | (x & 7));
}
__host__ __device__ __forceinline__ int blocked_tiles_h(int H) { return (H + 3) >> 2; }
__host__ __device__ __forceinline__ int blocked_tiles_w(int W) { return (W + 7) >> 3; }

// PIX consecutive source pixels per workgroup: the kernel is a chain of dependent latencies (coords -> patch
// loads -> LDS -> outputs), so each wave keeps PIX independent patches in flight instead of one — 4x fewer,
// 4x fatter workgroups, one resident round on the chip at 55x128.
// Every wave works on its own LDS region (its level's patches and tap tables), so the only synchronisation needed between
// staging and sampling is within the wave: LDS operations of a wave complete in order, `s_waitcnt lgkmcnt(0)` (also a
// compiler barrier) is enough — no workgroup barrier couples the four levels' very different amounts of work.

// (pixels per workgroup: template parameter PIX, 4 by default — lookup_launch)

// R = radius (compile-time: the (2R+1)^2 sample enumeration divides by constants), PIX pixels per workgroup.
// SHFL (measurement variant, pfk_debug_set_lookup_pix(14)): the staged patch stays in the registers it was loaded into and
// every tap is fetched with cross-lane reads (ds_bpermute) instead of LDS stores + loads — the alternative DESIGN.md section 3
// weighs against the LDS staging; same arithmetic, same bits.  profiles/r04_e_lookup_shuffle.md has the numbers.
// BLK: the levels are in the blocked 4 x 8 layout (pfk_corr_lookup_blocked_*); only the address of a staged element changes.
template <int PIX, int R, typename T, bool SHFL = false, bool BLK = false>
__global__ __launch_bounds__(256) void lookup_kernel(const LookupArgs a) {
  constexpr int n = 2 * R + 1, nn = n * n;
  // PAIR (bf16 maps in the blocked layout): the window is fetched as 12 rows x 7 aligned 32-bit words — two horizontally adjacent
  // bf16 elements of one 4 x 8 tile row — from the even column at or left of the window's first one: 84 word loads per window
  // (two per lane) instead of 144 16-bit loads (three per lane; measured 69 us against 52 for the same lookup on fp32 maps at batch 8,
  // gpurun_out/r6l_lookup.log: the sub-dword loads, not the bytes, set the pace).  The staged patch is 12 x 14, the tap tables carry
  // the window's odd/even column offset; same elements, same arithmetic, same bits.
  constexpr bool PAIR = BLK && !SHFL && sizeof(T) == 2;
  constexpr int PW = PAIR ? PATCH + 2 : PATCH, PLD = PAIR ? PATCH_LD + 2 : PATCH_LD;
  __shared__ float s_patch[4][PIX][PATCH * PLD];
  __shared__ float s_wx[4][PIX][12], s_wy[4][PIX][12];
  __shared__ int s_rx[4][PIX][12], s_ry[4][PIX][12];   // per window index: patch column / row (x 13) of the north-west tap

  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  // 32-bit pixel arithmetic (the host guarantees B*h*w < 2^31): the block index is wave-uniform, so everything derived from
  // it lives on the CU's single scalar unit — 64-bit divisions and multiplications there were the kernel's bottleneck
  const unsigned M = (unsigned)a.B * (unsigned)a.h * (unsigned)a.w;
  const unsigned p0 = blockIdx.x * (unsigned)PIX;
  const unsigned N = (unsigned)(a.h * a.w);

  float cx0[PIX], cy0[PIX];
  {
    unsigned b = p0 / N, pix = p0 - b * N;      // one division per block; the other pixels step from it
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
    if (active) {
      const int Hl = a.lh[l], Wl = a.lw[l];
      const float inv = 1.0f / (float)(1 << l);   // coords / 2**l is exact (corr.py:45)
      const int twl = blocked_tiles_w(Wl);
      const unsigned mapsz = BLK ? (unsigned)(blocked_tiles_h(Hl) * twl * 32) : (unsigned)(Hl * Wl);   // elements per source pixel
      const T* vol0 = static_cast<const T*>(a.lv[l]) + (size_t)p0 * mapsz;   // one 32x32->64 multiply per level
      float v[PIX][3];
      // tap tables of all PIX pixels in one pass (two for PIX = 8): entry e = q * n + i evaluates window index i of pixel q on
      // both axes (two IEEE divisions per lane; one pass over 36 lanes instead of PIX passes over 9)
      for (int e = lane; e < PIX * n; e += 64) {
        const int q = e / n, i = e - q * n;
        float cxq = cx0[0], cyq = cy0[0];
#pragma unroll
        for (int qq = 1; qq < PIX; ++qq) { cxq = (q == qq) ? cx0[qq] : cxq; cyq = (q == qq) ? cy0[qq] : cyq; }
        const float cx = cxq * inv, cy = cyq * inv;
        const float xb = floorf(cx) - (float)(R + 1);
        const float yb = floorf(cy) - (float)(R + 1);
        const float off = (float)(i - R);
        const float ix = roundtrip(cx + off, (float)(Wl - 1), (float)(Wl - 1) * 0.5f);
        const float iy = roundtrip(cy + off, (float)(Hl - 1), (float)(Hl - 1) * 0.5f);
        const float x0 = floorf(ix), y0 = floorf(iy);
        const float dxf = x0 - xb, dyf = y0 - yb;   // position of the tap inside the staged patch (absurd / NaN -> 0)
        s_rx[wid][q][i] = ((dxf >= 0.f && dxf <= (float)(PATCH - 2)) ? (int)dxf : 0) + (PAIR ? (safe_base(xb) & 1) : 0);
        s_ry[wid][q][i] = ((dyf >= 0.f && dyf <= (float)(PATCH - 2)) ? (int)dyf : 0) * PLD;
        s_wx[wid][q][i] = ix - x0;
        s_wy[wid][q][i] = iy - y0;
      }
#pragma unroll
      for (int q = 0; q < PIX; ++q) {
        const float cx = cx0[q] * inv, cy = cy0[q] * inv;
        const float xb = floorf(cx) - (float)(R + 1);
        const float yb = floorf(cy) - (float)(R + 1);
        const int xbi = safe_base(xb), ybi = safe_base(yb);
        const bool pl = (p0 + q) < M;
        const T* vol = vol0 + (size_t)q * mapsz;
        if constexpr (PAIR) {
          const int xbe = xbi & ~1;               // even column at or left of the window (two's complement: also for negative bases)
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {        // all loads of all PIX patches are issued before any is used
            const int e = lane + 64 * e2;
            const int yy = e / (PW / 2), xp = e - yy * (PW / 2);
            const int gy = ybi + yy, gx = xbe + 2 * xp;
            unsigned wv = 0u;
            if (pl && e < PATCH * (PW / 2) && (unsigned)gy < (unsigned)Hl && (unsigned)gx < (unsigned)Wl)
              wv = *reinterpret_cast<const unsigned*>(vol + blocked_index(gy, gx, twl));     // gx even: both halves in one tile row
            if (gx + 1 >= Wl) wv &= 0xffffu;      // the odd partner lies outside the map: zero padding, whatever the tile holds there
            v[q][e2] = __builtin_bit_cast(float, wv);
          }
        } else {
#pragma unroll
        for (int e3 = 0; e3 < 3; ++e3) {          // all loads of all PIX patches are issued before any is used
          const int e = lane + 64 * e3;
          const int yy = e / PATCH, xx = e - yy * PATCH;
          const int gy = ybi + yy, gx = xbi + xx;
          float t = 0.f;
          if (pl && e < PATCH * PATCH && (unsigned)gy < (unsigned)Hl && (unsigned)gx < (unsigned)Wl)
            t = (float)vol[BLK ? blocked_index(gy, gx, twl) : gy * Wl + gx];   // 32-bit offset inside one map; bf16 volume: exact widening (grid_sample is fp32 under autocast)
          v[q][e3] = t;
        }
        }
      }
      if constexpr (SHFL) {
        // element e of a patch lives in lane e & 63, register e >> 6: a tap is three cross-lane reads (one per register) and
        // a select, per pixel (the register file cannot be indexed by a per-lane pixel number, so the flat sample list of
        // the LDS path is not available: 2 passes per pixel, 47 idle lanes in the second)
        wave_lds_sync();      // the tap tables
        const size_t outl = (size_t)p0 * (unsigned)a.out_ld + l * nn;
#pragma unroll
        for (int q = 0; q < PIX; ++q) {
#pragma unroll
          for (int kk = 0; kk < (nn + 63) / 64; ++kk) {
            const int k = lane + 64 * kk;
            const bool valid = k < nn && p0 + q < M;
            const int kc = k < nn ? k : 0;
            const int i = kc / n, j = kc - i * n;
            const float wx = s_wx[wid][q][i], wy = s_wy[wid][q][j];
            const int e00 = (s_ry[wid][q][j] / PATCH_LD) * PATCH + s_rx[wid][q][i];
            float tap[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int e = e00 + (c & 1) + (c >> 1) * PATCH;
              const float r0 = __shfl(v[q][0], e & 63, 64), r1 = __shfl(v[q][1], e & 63, 64), r2 = __shfl(v[q][2], e & 63, 64);
              tap[c] = (e >> 6) == 0 ? r0 : ((e >> 6) == 1 ? r1 : r2);
            }
            const float ex = 1.0f - wx, sy = 1.0f - wy;
            const float w_nw = sy * ex, w_ne = sy * wx, w_sw = wy * ex, w_se = wy * wx;
            float t = tap[0] * w_nw;
            t = fmaf(tap[1], w_ne, t);
            t = fmaf(tap[2], w_sw, t);
            t = fmaf(tap[3], w_se, t);
            if (valid) store_sample(a, outl + q * a.out_ld + k, t);
          }
        }
      } else {
      if constexpr (PAIR) {
#pragma unroll
        for (int q = 0; q < PIX; ++q)
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            const int e = lane + 64 * e2;
            if (e < PATCH * (PW / 2)) {
              const int yy = e / (PW / 2), xp = e - yy * (PW / 2);
              const unsigned wv = __builtin_bit_cast(unsigned, v[q][e2]);
              s_patch[wid][q][yy * PLD + 2 * xp] = __builtin_bit_cast(float, wv << 16);            // exact widening of the two halves
              s_patch[wid][q][yy * PLD + 2 * xp + 1] = __builtin_bit_cast(float, wv & 0xffff0000u);
            }
          }
      } else {
#pragma unroll
      for (int q = 0; q < PIX; ++q)
#pragma unroll
        for (int e3 = 0; e3 < 3; ++e3) {
          const int e = lane + 64 * e3;
          if (e < PATCH * PATCH) {
            const int yy = e / PATCH, xx = e - yy * PATCH;
            s_patch[wid][q][yy * PLD + xx] = v[q][e3];
          }
        }
      }
      }
    }
    wave_lds_sync();
    if (active && !SHFL) {
      // the PIX * (2R+1)^2 samples of this level as one flat list over the lanes (324 = 5.06 wave-iterations at R = 4
      // instead of 4 x 2 with 47 idle lanes in every second one)
      const size_t outl = (size_t)p0 * (unsigned)a.out_ld + l * nn;
      for (int idx = lane; idx < PIX * nn; idx += 64) {
        const int q = idx / nn, k = idx - q * nn;
        if (p0 + q >= M) continue;
        const int i = k / n, j = k - i * n;
        const float wx = s_wx[wid][q][i], wy = s_wy[wid][q][j];
        const float* pq = &s_patch[wid][q][s_ry[wid][q][j] + s_rx[wid][q][i]];
        const float nw = pq[0], ne = pq[1], sw = pq[PLD], se = pq[PLD + 1];
        const float ex = 1.0f - wx, sy = 1.0f - wy;
        const float w_nw = sy * ex, w_ne = sy * wx, w_sw = wy * ex, w_se = wy * wx;
        float t = nw * w_nw;
        t = fmaf(ne, w_ne, t);
        t = fmaf(sw, w_sw, t);
        t = fmaf(se, w_se, t);
        store_sample(a, outl + q * a.out_ld + k, t);
      }
    }
    wave_lds_sync();   // the next round restages this wave's region
  }
}

// ------------------------------------------------------------------------------------------
// K2 pooling: one thread per output element, HBM-bound (reads 4 B x 4, writes 4 B).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool2x2_kernel(const T* __restrict__ in,
                                                      T* __restrict__ out, long long total,
                                                      int H, int W, int Ho, int Wo) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int xo = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int yo = (int)(t % Ho);
    const long long m = t / Ho;
    const T* src = in + (m * H + 2 * yo) * (long long)W + 2 * xo;
    const float a00 = (float)src[0], a01 = (float)src[1], a10 = (float)src[W], a11 = (float)src[W + 1];
    out[idx] = (T)((((a00 + a01) + a10) + a11) * 0.25f);   // bf16: fp32 accumulate, one rounding (torch's opmath)
  }
}

// 2x2 / stride-2 average of a pixel-major feature map [B][H][W][ld] -> [B][H/2][W/2][out_ld], C channels (C % 4 == 0),
// same summation order as above.  This IS F.interpolate(scale_factor=0.5, mode="bilinear", align_corners=False): every
// output samples the centre of a 2x2 cell, four weights of exactly 0.25 (sea_raft/corr.py:81-83) — bit-identical to it.
__global__ __launch_bounds__(256) void fmap_pool2x2_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out,
                                                           int out_ld, long long total4, int C4, int H, int W, int Ho, int Wo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int c = (int)(idx % C4) * 4;
  long long t = idx / C4;
  const int xo = (int)(t % Wo); t /= Wo;
  const int yo = (int)(t % Ho);
  const long long b = t / Ho;
  const float* src = in + ((b * H + 2 * yo) * (long long)W + 2 * xo) * in_ld + c;
  const f32x4 a00 = *reinterpret_cast<const f32x4*>(src), a01 = *reinterpret_cast<const f32x4*>(src + in_ld);
  const f32x4 a10 = *reinterpret_cast<const f32x4*>(src + (long long)W * in_ld), a11 = *reinterpret_cast<const f32x4*>(src + (long long)(W + 1) * in_ld);
  *reinterpret_cast<f32x4*>(out + ((b * Ho + yo) * (long long)Wo + xo) * out_ld + c) = (((a00 + a01) + a10) + a11) * 0.25f;
}

// K2 on the blocked layout: in = M maps of (H, W) as 4 x 8 tiles, out = the (H/2, W/2) maps as 4 x 8 tiles.  One thread per
// element of the OUTPUT STORAGE (pad elements of edge tiles are written as zero, so the storage is fully defined); the 2 x 2
// cell of an output element lies inside ONE input tile (rows 2yo & 3 in {0, 2}, columns 2xo & 7 even).  Same summation order.
template <typename T>
__global__ __launch_bounds__(256) void pool2x2_blocked_kernel(const T* __restrict__ in, T* __restrict__ out, long long total,
                                                              int TH, int TW, int Ho, int Wo, int THo, int TWo) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int e = (int)(idx & 31);
    const long long t = idx >> 5;
    const int tx = (int)(t % TWo);
    const long long t2 = t / TWo;
    const int ty = (int)(t2 % THo);
    const long long m = t2 / THo;
    const int yo = ty * 4 + (e >> 3), xo = tx * 8 + (e & 7);
    float v = 0.f;
    if (yo < Ho && xo < Wo) {
      const T* src = in + m * ((long long)TH * TW * 32) + blocked_index(2 * yo, 2 * xo, TW);
      const float a00 = (float)src[0], a01 = (float)src[1], a10 = (float)src[8], a11 = (float)src[9];
      v = (((a00 + a01) + a10) + a11) * 0.25f;
    }
    out[idx] = (T)v;
  }
}

// Rows of a pixel-major feature map [B][H*W][in_ld] permuted into the blocked order: out [B][TH*TW*32][out_ld], row
// ((ty*TW + tx)*32 + r*8 + c) = pixel (4ty + r, 8tx + c), zero rows where that pixel lies outside the map.  K1 run against this
// map writes the volume's rows directly in the blocked layout — a GEMM does not care which order its B rows come in.
__global__ __launch_bounds__(256) void fmap_to_blocked_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out,
                                                              int out_ld, long long total4, int C4, int H, int W, int TH, int TW) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int c = (int)(idx % C4) * 4;
  long long t = idx / C4;
  const int e = (int)(t & 31); t >>= 5;
  const int tx = (int)(t % TW); t /= TW;
  const int ty = (int)(t % TH);
  const long long b = t / TH;
  const int y = ty * 4 + (e >> 3), x = tx * 8 + (e & 7);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (y < H && x < W) v = *reinterpret_cast<const f32x4*>(in + ((b * H + y) * (long long)W + x) * in_ld + c);
  *reinterpret_cast<f32x4*>(out + (((b * TH + ty) * (long long)TW + tx) * 32 + e) * out_ld + c) = v;
}

int g_lookup_pix = 4;   // pixels per workgroup: 4 or 8; 14 = 4 with cross-lane tap reads (pfk_debug_set_lookup_pix; tuning knob)

template <int PX, typename T, bool SHFL = false, bool BLK = false>
int lookup_launch_pix(const LookupArgs& a, int radius, long long M, hipStream_t st) {
  const dim3 grid((unsigned)((M + PX - 1) / PX)), block(256);
  switch (radius) {
    case 1: hipLaunchKernelGGL((lookup_kernel<PX, 1, T, SHFL, BLK>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((lookup_kernel<PX, 2, T, SHFL, BLK>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((lookup_kernel<PX, 3, T, SHFL, BLK>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((lookup_kernel<PX, 4, T, SHFL, BLK>), grid, block, 0, st, a); break;
  }
  return pfk_launch_status();
}

template <typename T, bool BLK = false>
int lookup_launch(const pfk_lookup_desc* d, pfk_stream_t stream) {
  if (!d || !d->coords || !d->out) return PFK_ERR_BAD_ARG;
  if (d->num_levels < 1 || d->num_levels > PFK_MAX_LEVELS) return PFK_ERR_BAD_ARG;
  if (d->radius < 1 || d->radius > 4) return PFK_ERR_UNSUPPORTED;
  if (d->B <= 0 || d->h <= 0 || d->w <= 0) return PFK_ERR_BAD_ARG;
  const int n = 2 * d->radius + 1;
  if (d->out_ld < d->num_levels * n * n) return PFK_ERR_BAD_ARG;
  LookupArgs a{};
  for (int l = 0; l < d->num_levels; ++l) {
    if (!d->levels[l] || d->lvl_h[l] <= 0 || d->lvl_w[l] <= 0) return PFK_ERR_BAD_ARG;
    a.lv[l] = d->levels[l]; a.lh[l] = d->lvl_h[l]; a.lw[l] = d->lvl_w[l];
  }
  a.L = d->num_levels; a.r = d->radius; a.B = d->B; a.h = d->h; a.w = d->w;
  a.coords = d->coords; a.out = d->out; a.out_ld = d->out_ld; a.out_bf16 = d->out_bf16;
  const long long M = (long long)d->B * d->h * d->w;
  if (M >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;   // the kernel's pixel arithmetic is 32-bit
  hipStream_t st = static_cast<hipStream_t>(stream);
  if constexpr (BLK) {
    for (int l = 0; l < d->num_levels; ++l)     // 32-bit element offsets inside one source pixel's map
      if ((long long)blocked_tiles_h(d->lvl_h[l]) * blocked_tiles_w(d->lvl_w[l]) * 32 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
    if constexpr (sizeof(T) == 2)               // bf16 maps are read as aligned 32-bit pairs
      for (int l = 0; l < d->num_levels; ++l)
        if (reinterpret_cast<uintptr_t>(d->levels[l]) & 3u) return PFK_ERR_ALIGNMENT;
    // 8 pixels per workgroup from 28 160 source pixels up (a pyramid far beyond the 256 MB Infinity Cache: more windows in
    // flight per wave pay; 55x128 batch 8, iid field 60.7 -> 55.9 us, smooth 55.2 -> 54.7), 4 below (batch 1: 10.1 vs 12.2 us;
    // gpurun_out/r5c_lookup.log).  g_lookup_pix: 4 = this rule, 8 = always 8, 104 = always 4 (tuning knob).
    const bool eight = g_lookup_pix == 8 || (g_lookup_pix == 4 && M >= 28160);
    return eight ? lookup_launch_pix<8, T, false, true>(a, d->radius, M, st) : lookup_launch_pix<4, T, false, true>(a, d->radius, M, st);
  }
  if (g_lookup_pix == 14) return lookup_launch_pix<4, T, true>(a, d->radius, M, st);
  return g_lookup_pix == 8 ? lookup_launch_pix<8, T>(a, d->radius, M, st) : lookup_launch_pix<4, T>(a, d->radius, M, st);
}

template <typename T>
int pool_blocked_launch(const T* in, T* out, int64_t M, int H, int W, pfk_stream_t stream) {
  if (!in || !out || M <= 0 || H <= 0 || W <= 0) return PFK_ERR_BAD_ARG;
  const int Ho = H / 2, Wo = W / 2;
  const int THo = blocked_tiles_h(Ho), TWo = blocked_tiles_w(Wo);
  const long long total = (long long)M * THo * TWo * 32;
  if (total == 0) return PFK_OK;  // a 1-pixel level pools to nothing
  long long blocks = (total + 255) / 256;
  if (blocks > 256LL * 32) blocks = 256LL * 32;
  hipLaunchKernelGGL(pool2x2_blocked_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, total,
                     blocked_tiles_h(H), blocked_tiles_w(W), Ho, Wo, THo, TWo);
  return pfk_launch_status();
}

template <typename T>
int pool_launch(const T* in, T* out, int64_t M, int H, int W, pfk_stream_t stream) {
  if (!in || !out || M <= 0 || H <= 0 || W <= 0) return PFK_ERR_BAD_ARG;
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)M * Ho * Wo;
  if (total == 0) return PFK_OK;  // a 1-pixel level pools to nothing
  long long blocks = (total + 255) / 256;
  if (blocks > 256LL * 32) blocks = 256LL * 32;  // grid-stride beyond 32 blocks per CU
  hipLaunchKernelGGL(pool2x2_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, total, H,
                     W, Ho, Wo);
  return pfk_launch_status();
}

}  // namespace

extern "C" {

int pfk_debug_set_lookup_pix(int pix) { if (!pfk_debug_knobs_enabled()) return PFK_ERR_DISABLED; g_lookup_pix = (pix == 8 || pix == 14 || pix == 104) ? pix : 4; return PFK_OK; }

int pfk_corr_lookup_f32(const pfk_lookup_desc* d, pfk_stream_t stream) { return lookup_launch<float>(d, stream); }

// levels[] point to bf16 maps (the pyramid of pfk_corr_volume_bf16 / pfk_corr_pool2x2_bf16); coordinates, weights,
// accumulation and the output stay fp32 — F.grid_sample is on autocast's fp32 list, so that is what the reference runs.
int pfk_corr_lookup_bf16(const pfk_lookup_desc* d, pfk_stream_t stream) { return lookup_launch<__bf16>(d, stream); }

int pfk_corr_lookup_blocked_f32(const pfk_lookup_desc* d, pfk_stream_t stream) { return lookup_launch<float, true>(d, stream); }

int pfk_corr_lookup_blocked_bf16(const pfk_lookup_desc* d, pfk_stream_t stream) { return lookup_launch<__bf16, true>(d, stream); }

int pfk_corr_pool2x2_blocked_f32(const float* in, float* out, int64_t M, int H, int W, pfk_stream_t stream) {
  return pool_blocked_launch<float>(in, out, M, H, W, stream);
}

int pfk_corr_pool2x2_blocked_bf16(const void* in, void* out, int64_t M, int H, int W, pfk_stream_t stream) {
  return pool_blocked_launch<__bf16>(static_cast<const __bf16*>(in), static_cast<__bf16*>(out), M, H, W, stream);
}

int64_t pfk_blocked_map_elems(int H, int W) { return (int64_t)blocked_tiles_h(H) * blocked_tiles_w(W) * 32; }

int pfk_fmap_to_blocked_f32(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, pfk_stream_t stream) {
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || in_ld < C || out_ld < C) return PFK_ERR_BAD_ARG;
  if ((C & 3) || (in_ld & 3) || (out_ld & 3) || !pfk_aligned16(in) || !pfk_aligned16(out)) return PFK_ERR_ALIGNMENT;
  const int TH = blocked_tiles_h(H), TW = blocked_tiles_w(W);
  const long long total4 = (long long)B * TH * TW * 32 * (C / 4);
  if ((total4 + 255) / 256 > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fmap_to_blocked_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     in_ld, out, out_ld, total4, C / 4, H, W, TH, TW);
  return pfk_launch_status();
}

int pfk_corr_pool2x2_f32(const float* in, float* out, int64_t M, int H, int W, pfk_stream_t stream) {
  return pool_launch<float>(in, out, M, H, W, stream);
}

int pfk_corr_pool2x2_bf16(const void* in, void* out, int64_t M, int H, int W, pfk_stream_t stream) {
  return pool_launch<__bf16>(static_cast<const __bf16*>(in), static_cast<__bf16*>(out), M, H, W, stream);
}

int pfk_fmap_pool2x2_f32(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, pfk_stream_t stream) {
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || in_ld < C || out_ld < C) return PFK_ERR_BAD_ARG;
  if ((C & 3) || (in_ld & 3) || (out_ld & 3) || !pfk_aligned16(in) || !pfk_aligned16(out)) return PFK_ERR_ALIGNMENT;
  const int Ho = H / 2, Wo = W / 2;
  const long long total4 = (long long)B * Ho * Wo * (C / 4);
  if (total4 == 0) return PFK_OK;
  if ((total4 + 255) / 256 > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fmap_pool2x2_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     in_ld, out, out_ld, total4, C / 4, H, W, Ho, Wo);
  return pfk_launch_status();
}

}  // extern "C"
