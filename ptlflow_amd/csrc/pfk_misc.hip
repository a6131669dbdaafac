// Small direct (VALU) kernels around the MFMA convolutions: the 2-channel 7x7 flow convolution,
// the 2-output flow head fused with the coordinate update, convex upsampling, layout changes.
// None of them is GEMM-shaped enough to pay for a matrix-core tile; they are bandwidth/latency
// bound and written for coalesced 64-lane access.
#include "pfk_common.h"

namespace {

__device__ __forceinline__ f32x4 load4_widen(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4_widen(const __bf16* p) {      // four bf16 (8 bytes) -> fp32, exact
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 w = *reinterpret_cast<const u32x2*>(p);
  f32x4 v;
  v.x = __builtin_bit_cast(float, w[0] << 16); v.y = __builtin_bit_cast(float, w[0] & 0xffff0000u);
  v.z = __builtin_bit_cast(float, w[1] << 16); v.w = __builtin_bit_cast(float, w[1] & 0xffff0000u);
  return v;
}


// out[p][co] = relu?( bias[co] + sum_{tap} in[p+off][0]*w[tap][0][co] + in[p+off][1]*w[tap][1][co] )
// raft/update.py:100,107 (convf1).  Thread per (pixel, co); the flow taps are wave-broadcast
// loads, the weights are read coalesced along co.
__global__ __launch_bounds__(256) void conv_cin2_kernel(const float* __restrict__ in, int in_ld,
                                                        const float* __restrict__ wgt,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ out, int out_ld,
                                                        int out_coff, long long M, int H, int W,
                                                        int k, int cout, int relu) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * cout) return;
  const long long p = idx / cout;
  const int co = (int)(idx - p * cout);
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const int half = k >> 1;
  float acc = bias ? bias[co] : 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int yy = y + ky - half;
    if ((unsigned)yy >= (unsigned)H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int xx = x + kx - half;
      if ((unsigned)xx >= (unsigned)W) continue;
      const float* src = in + (p + (long long)(ky - half) * W + (kx - half)) * in_ld;
      const float* w = wgt + (long long)(ky * k + kx) * 2 * cout;
      acc = fmaf(src[0], w[co], acc);
      acc = fmaf(src[1], w[cout + co], acc);
    }
  }
  if (relu) acc = (acc < 0.f) ? 0.f : acc;  // NaN-propagating like torch.relu
  out[p * out_ld + out_coff + co] = acc;
}

// Tiled version for the common odd k <= 7: one block = 16 consecutive pixels of one image row x up to 128
// output channels (thread = one channel x 8 pixels).  The (k x (16+k-1)) two-channel flow patch is staged once
// in LDS with the zero padding applied there; per ky a thread pulls its 8+k-1 patch columns into registers
// (LDS broadcast reads) and reuses them across kx, so the inner loop is pure FMA.  Same FMA order as the
// simple kernel (tap-major, x then y channel; padded taps add exactly 0), so results are bit-identical to it.
template <int K, typename TO = float>
__global__ __launch_bounds__(256) void conv_cin2_tiled_kernel(const float* __restrict__ in, int in_ld,
                                                              const float* __restrict__ wgt,
                                                              const float* __restrict__ bias,
                                                              TO* __restrict__ out, int out_ld,
                                                              int out_coff, int H, int W, int tiles_per_row,
                                                              int cout, int relu) {
  constexpr int TP = 16, R = K / 2, PW = TP + 2 * R, NP = 8;
  __shared__ float sf[2][K][PW];
  const int tile = blockIdx.x;
  const int row = tile / tiles_per_row;           // b*H + y
  const int x0 = (tile - row * tiles_per_row) * TP;
  const int y = row % H;
  const long long rowbase = (long long)(row - y) * W;   // b*H*W
  for (int e = threadIdx.x; e < K * PW; e += 256) {
    const int ky = e / PW, xx = e - ky * PW;
    const int gy = y + ky - R, gx = x0 + xx - R;
    float fx = 0.f, fy = 0.f;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
      const float* src = in + (rowbase + (long long)gy * W + gx) * in_ld;
      fx = src[0];
      fy = src[1];
    }
    sf[0][ky][xx] = fx;
    sf[1][ky][xx] = fy;
  }
  __syncthreads();
  const int g = threadIdx.x >> 7;
  const int xs = x0 + g * NP;
  for (int c = threadIdx.x & 127; c < cout; c += 128) {
    float acc[NP];
    const float b0 = bias ? bias[c] : 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = b0;
#pragma unroll 1   // unrolling ky as well hoists all K*K*2 weight loads (326 VGPRs at K = 7: one wave per SIMD)
    for (int ky = 0; ky < K; ++ky) {
      float fx[NP + K - 1], fy[NP + K - 1];
#pragma unroll
      for (int j = 0; j < NP + K - 1; ++j) {
        fx[j] = sf[0][ky][g * NP + j];
        fy[j] = sf[1][ky][g * NP + j];
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float* w = wgt + (long long)(ky * K + kx) * 2 * cout;
        const float w0 = w[c], w1 = w[cout + c];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          acc[p] = fmaf(fx[p + kx], w0, acc[p]);
          acc[p] = fmaf(fy[p + kx], w1, acc[p]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (xs + p < W) {
        float v = acc[p];
        if (relu) v = (v < 0.f) ? 0.f : v;
        out[(rowbase + (long long)y * W + xs + p) * out_ld + out_coff + c] = (TO)v;      // bf16 output: round to nearest even
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same 7x7 convolution on the matrix cores (round 6; cout = 64 or 128 — SmallMotionEncoder / BasicMotionEncoder's convf1,
// raft/update.py:80,98).  The tiled VALU kernel runs the 1.4 GFLOP of a batch-8 launch in 31.5 us (33 us with bf16 rows out) — per
// iteration, on every path (fp32: 1.0 ms of a 101 ms forward; K8b: 1.05 ms of 27).  As an implicit GEMM M = pixels, N = cout, K = 98 on
// v_mfma_f32_32x32x2f32 a K-step is ONE tap: k = 2 tap + c, so the two halves of a wave read the x / y flow component of the same
// patch cell.  A workgroup = 4 rows x 32 columns of output pixels (wave w: row w, one 32-pixel M tile x NT 32-channel N tiles); the
// 2 x 10 x 38 flow patch (zero padding applied while staging) and the whole [98][cout] weight sit in LDS (3 + 50 KB at cout 128:
// two workgroups per CU); persistent over the tile list (the weight is staged once per workgroup).  The accumulators START at the
// bias and the MFMA adds k = 0, 1, 2, ... in order — an fmaf chain in the tiled kernel's order (tap-major, x then y): same bits.
// ------------------------------------------------------------------------------------------------------------------
constexpr int C2M_ROWS = 4, C2M_COLS = 32, C2M_K = 7, C2M_R = 3;
constexpr int C2M_PR = C2M_ROWS + C2M_K - 1, C2M_PC = C2M_COLS + C2M_K - 1;     // 10 x 38 patch cells
constexpr int C2M_PATCH = C2M_PR * C2M_PC;                                        // per flow component
constexpr int C2M_STEPS = C2M_K * C2M_K;                                          // 49 K-steps of 2

template <int NT, int S>
struct Cin2Steps {
  static __device__ __forceinline__ void run(f32x16 (&acc)[NT], const float* s_in, const float* s_w, int abase, int hl, int px) {
    constexpr int ky = S / C2M_K, kx = S % C2M_K;
    const float a = s_in[abase + ky * C2M_PC + kx];              // abase already holds this lane half's component plane
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s_w[(2 * S + hl) * (NT * 32) + nt * 32 + px], acc[nt], 0, 0, 0);
    if constexpr (S + 1 < C2M_STEPS) Cin2Steps<NT, S + 1>::run(acc, s_in, s_w, abase, hl, px);
  }
};

template <int NT, typename TO>
__global__ __launch_bounds__(256, 2) void conv_cin2_mfma_kernel(const float* __restrict__ in, int in_ld, const float* __restrict__ wgt,
                                                                 const float* __restrict__ bias, TO* __restrict__ out, int out_ld,
                                                                 int out_coff, int H, int W, int tiles_x, int tiles_y, int ntiles,
                                                                 int relu) {
  constexpr int COUT = NT * 32;
  __shared__ float s_in[2 * C2M_PATCH];
  __shared__ float s_w[2 * C2M_STEPS * COUT];
  for (int e = threadIdx.x; e < 2 * C2M_STEPS * COUT / 4; e += 256)
    *reinterpret_cast<f32x4*>(s_w + e * 4) = *reinterpret_cast<const f32x4*>(wgt + e * 4);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int px = lane & 31, hl = lane >> 5;
  const int abase = hl * C2M_PATCH + wid * C2M_PC + px;
  float b0[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b0[nt] = bias ? bias[nt * 32 + px] : 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % tiles_x, t1 = tile / tiles_x;
    const int ty = t1 % tiles_y, b = t1 / tiles_y;
    const int y0 = ty * C2M_ROWS, x0 = tx * C2M_COLS;
    const long long img = (long long)b * H * W;
    __syncthreads();                     // the previous tile's reads of s_in are done (first pass: orders nothing, harmless)
    for (int e = threadIdx.x; e < C2M_PATCH; e += 256) {
      const int yy = e / C2M_PC, xx = e - yy * C2M_PC;
      const int gy = y0 + yy - C2M_R, gx = x0 + xx - C2M_R;
      float fx = 0.f, fy = 0.f;
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
        const float* src = in + (img + (long long)gy * W + gx) * in_ld;
        fx = src[0];
        fy = src[1];
      }
      s_in[e] = fx;
      s_in[C2M_PATCH + e] = fy;
    }
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = b0[nt];
    Cin2Steps<NT, 0>::run(acc, s_in, s_w, abase, hl, px);
    const int y = y0 + wid;
    if (y < H) {
      const long long row0 = img + (long long)y * W;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = nt * 32 + px;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
          if (x >= W) continue;
          float v = acc[nt][r];
          if (relu) v = (v < 0.f) ? 0.f : v;
          out[(row0 + x) * out_ld + out_coff + c] = (TO)v;
        }
      }
    }
  }
}

int g_cin2_valu = 0;           // pfk_debug_set_cin2_valu: 0 = by size, 1 = always the tiled VALU kernel, 2 = the MFMA kernel at every size

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one reduce-scatter step: of 2 N values a lane keeps the half selected by its bit OFF and adds the partner's copy of it
template <int N, int OFF>
__device__ __forceinline__ void rs_step(float* s, int lane) {
  const bool up = (lane & OFF) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float lo = s[i], hi = s[N + i];
    s[i] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, OFF, 64);
  }
}

// FlowHead.conv2 (3x3, cin -> 2; raft/update.py:10,14) + RAFT loop bookkeeping (raft.py:174,178).
// One wave per PIX consecutive pixels of one image row: lanes split the input channels (float4 each, coalesced 1 KiB rows), the
// wave walks the 3 x (PIX + 2) input rows its pixels touch ONCE each and feeds every row to the (up to three) pixels that use it
// (3.75 row loads per pixel at PIX = 8 instead of 9), one row band — 6 weight float4s — at a time.  Per pixel the taps still
// arrive in (ky, kx) order and the channel FMAs in the same chain, so the per-lane sums are those of the tap-by-tap formulation.
// Tail: the 2 PIX per-lane sums are reduced over the 64 lanes by a reduce-scatter butterfly (offsets 32, 16, ... exactly the
// pairing order of `wave_sum`, so every total has wave_sum's bits; 17 instead of 96 lane exchanges at PIX = 8) that leaves total
// j in the lanes whose upper bits spell j — and THOSE lanes apply coords1 += delta and flow = coords1 - coords0 for their
// (pixel, component) in parallel: one global round trip per wave instead of PIX dependent ones on lane 0 (which was most of this
// kernel's time: it is a latency chain, not a bandwidth problem).
template <int PIX, typename TI = float>
__global__ __launch_bounds__(256) void flow_delta_kernel(
    const TI* __restrict__ in, int in_ld, int cin, const float* __restrict__ wgt,
    const float* __restrict__ bias, const float* __restrict__ coords0, float* coords1,
    float* delta_out, float* flow_out, int flow_ld, __bf16* flow_b16, int flow_b16_ld, long long rows, int H, int W, int tpr) {
  static_assert(PIX == 4 || PIX == 8, "two reduce-scatter depths are written out below");
  const int lane = threadIdx.x & 63;
  // wave-uniform on purpose (readfirstlane): the row / column tests below must be scalar branches, not per-lane ones whose phi
  // copies of the 2 PIX accumulators doubled the kernel's registers
  const long long w = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long rowid = w / tpr;                  // b * H + y
  if (rowid >= rows) return;
  const int x0 = (int)(w - rowid * tpr) * PIX;
  const int y = (int)(rowid % H);
  float s[2 * PIX];                                 // s[o * PIX + q]: component o of pixel q
#pragma unroll
  for (int j = 0; j < 2 * PIX; ++j) s[j] = 0.f;
  for (int c = lane * 4; c < cin; c += 256) {
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {                // one row band at a time: 6 weight float4s + PIX + 2 rows in registers
      // zero padding without branches: an out-of-range row / column is read from a clamped (valid) address and replaced by zeros
      // — the reference's padded convolution adds exactly these 0 * w terms; branches here cost phi copies of all accumulators
      const int yy = y + ky - 1;
      const bool yvalid = (unsigned)yy < (unsigned)H;
      f32x4 wa[3], wb[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        wa[kx] = *reinterpret_cast<const f32x4*>(wgt + (long long)((ky * 3 + kx) * 2) * cin + c);
        wb[kx] = *reinterpret_cast<const f32x4*>(wgt + (long long)((ky * 3 + kx) * 2 + 1) * cin + c);
      }
      const TI* rowp = in + ((rowid + (yvalid ? ky - 1 : 0)) * W) * in_ld + c;
#pragma unroll
      for (int xi = 0; xi < PIX + 2; ++xi) {
        const int xx = x0 - 1 + xi;
        const bool valid = yvalid && (unsigned)xx < (unsigned)W;          // wave-uniform
        const int xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        f32x4 v = load4_widen(rowp + (long long)xc * in_ld);
        if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 2; kx >= 0; --kx) {            // pixel q = xi - kx sees this row as its tap kx (ascending kx per pixel as xi grows)
          const int q = xi - kx;
          if (q < 0 || q >= PIX) continue;
          const f32x4 a = wa[kx], b = wb[kx];
          s[q] = fmaf(v.x, a.x, fmaf(v.y, a.y, fmaf(v.z, a.z, fmaf(v.w, a.w, s[q]))));
          s[PIX + q] = fmaf(v.x, b.x, fmaf(v.y, b.y, fmaf(v.z, b.z, fmaf(v.w, b.w, s[PIX + q]))));
        }
      }
    }
  }
  // reduce-scatter: at offset `off` a lane keeps the half of its values selected by its bit `off` and adds the partner's copy
  // of that half (own + partner's: the operands of wave_sum's step at the same offset)
  rs_step<PIX, 32>(s, lane);                        // 2 PIX -> PIX values per lane
  rs_step<PIX / 2, 16>(s, lane);
  rs_step<PIX / 4, 8>(s, lane);
  if constexpr (PIX == 8) rs_step<1, 4>(s, lane);
  float tot = s[0];
#pragma unroll
  for (int off = (PIX == 8 ? 2 : 4); off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
  // lane l now holds total j = l >> (PIX == 8 ? 2 : 3) (bit 5 of the lane = most significant bit of j = the component)
  constexpr int REP = PIX == 8 ? 4 : 8;             // lanes per total
  const int j = lane / REP, o = j / PIX, q = j - o * PIX;
  if ((lane & (REP - 1)) == 0 && x0 + q < W) {
    const long long hw = (long long)H * W;
    const long long bimg = rowid / H;
    const long long p = rowid * W + x0 + q, pix = p - bimg * hw;
    const long long ic = (bimg * 2 + o) * hw + pix;
    const float d = tot + (bias ? bias[o] : 0.f);
    const float c1 = __fadd_rn(coords1[ic], d);
    coords1[ic] = c1;
    if (delta_out) delta_out[ic] = d;
    const float fl = __fsub_rn(c1, coords0[ic]);
    if (flow_out) flow_out[p * flow_ld + o] = fl;
    if (flow_b16) flow_b16[p * flow_b16_ld + o] = (__bf16)fl;       // the bf16 copy the K8b convolutions read (hx's 16-bit twin)
  }
}

__global__ __launch_bounds__(256) void flow_from_coords_kernel(const float* __restrict__ c0,
                                                               const float* __restrict__ c1,
                                                               float* __restrict__ flow_out,
                                                               int flow_ld, long long M,
                                                               long long hw) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const long long b = p / hw, pix = p - b * hw;
  const long long ix = (b * 2 + 0) * hw + pix, iy = (b * 2 + 1) * hw + pix;
  flow_out[p * flow_ld + 0] = c1[ix] - c0[ix];
  flow_out[p * flow_ld + 1] = c1[iy] - c0[iy];
}

// upflow8 (raft/utils.py:94-96; raft_small has no mask head): 8 * bilinear(flow, size = 8x, align_corners = True) with
// flow = coords1 - coords0, NCHW [B][2][H][W] -> [B][2][8H][8W].  Index arithmetic as torch's upsample_bilinear2d: scale =
// (in - 1) / (out - 1) in fp32, src = scale * dst, i0 = trunc(src), lambda1 = src - i0, i1 = i0 + (i0 < in - 1).  One thread
// per output pixel (x fastest: coalesced stores; the 2 x 2 x 2 reads hit a tiny L2-resident input); write-bound.
__global__ __launch_bounds__(256) void upflow8_kernel(const float* __restrict__ c0, const float* __restrict__ c1,
                                                      float* __restrict__ out, int B, int H, int W) {
  const int Ho = 8 * H, Wo = 8 * W;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per = (long long)Ho * Wo;
  if (idx >= per * B) return;
  const int b = (int)(idx / per);
  const int r = (int)(idx - (long long)b * per);
  const int yo = r / Wo, xo = r - yo * Wo;
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const float fy = sy * (float)yo, fx = sx * (float)xo;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const long long hw = (long long)H * W;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const float* p0 = c0 + ((long long)b * 2 + ch) * hw;
    const float* p1 = c1 + ((long long)b * 2 + ch) * hw;
    const float v00 = p1[y0 * W + x0] - p0[y0 * W + x0], v01 = p1[y0 * W + x1] - p0[y0 * W + x1];
    const float v10 = p1[y1 * W + x0] - p0[y1 * W + x0], v11 = p1[y1 * W + x1] - p0[y1 * W + x1];
    const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    out[((long long)b * 2 + ch) * per + r] = 8.f * v;
  }
}

// RAFT.upsample_flow (raft.py:112-123).  One wave per coarse pixel, lane = sy*8 + sx: the nine
// mask reads are 256-byte coalesced rows, the 3x3 flow neighbourhood is broadcast.
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow, int flow_ld,
                                                              const float* __restrict__ mask,
                                                              int mask_ld, float* __restrict__ out,
                                                              long long M, int H, int W) {
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
  float ox = 0.f, oy = 0.f;
  const float* fx = flow + (b * 2 + 0) * hw;
  const float* fy = flow + (b * 2 + 1) * hw;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    float vx = 0.f, vy = 0.f;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
      if (flow_ld > 0) {   // pixel-major: flow[p][0..1] (the update engine's hx slice)
        const float* f = flow + (b * hw + (long long)yy * W + xx) * flow_ld;
        vx = 8.0f * f[0];
        vy = 8.0f * f[1];
      } else {             // NCHW
        vx = 8.0f * fx[yy * W + xx];
        vy = 8.0f * fy[yy * W + xx];
      }
    }
    const float wk = m[k] * inv;
    ox += wk * vx;
    oy += wk * vy;
  }
  const int sy = lane >> 3, sx = lane & 7;
  const long long HW8 = hw * 64;
  const long long o = (long long)(8 * y + sy) * (8 * W) + 8 * x + sx;
  out[(b * 2 + 0) * HW8 + o] = ox;
  out[(b * 2 + 1) * HW8 + o] = oy;
}

// The same arithmetic, four coarse pixels per wave: lane = pixel-in-group * 16 + float4 index inside the 64 mask floats of a tap
// (sy = idx >> 1, sx = 4 * (idx & 1) .. +3), so a tap is ONE 1 KiB load per wave (9 per 4 pixels instead of 36 of 256 bytes) and an
// output row piece of the group is 128 contiguous bytes.  Every output element runs the scalar kernel's operation sequence
// (max, exp, sum, one reciprocal, taps in order): the two kernels produce the same bits.  Needs 16-byte aligned mask rows.
// TM = __bf16: the mask as the K8b mask head writes it (2-byte logits, widened exactly; half the bytes of this HBM-bound kernel).
template <typename TM = float>
__global__ __launch_bounds__(256) void convex_upsample4_kernel(const float* __restrict__ flow, int flow_ld,
                                                               const TM* __restrict__ mask,
                                                               int mask_ld, float* __restrict__ out,
                                                               long long M, int H, int W) {
  const int lane = threadIdx.x & 63;
  const long long p = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  if (p >= M) return;
  const int idx = lane & 15;
  const long long hw = (long long)H * W;
  const long long b = p / hw;
  const int pix = (int)(p - b * hw);
  const int y = pix / W, x = pix - y * W;
  const TM* mrow = mask + p * mask_ld + idx * 4;
  f32x4 m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = load4_widen(mrow + k * 64);
  float vx[9], vy[9];
  const float* fx = flow + (b * 2 + 0) * hw;
  const float* fy = flow + (b * 2 + 1) * hw;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    vx[k] = 0.f; vy[k] = 0.f;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
      if (flow_ld > 0) {
        const float* f = flow + (b * hw + (long long)yy * W + xx) * flow_ld;
        vx[k] = 8.0f * f[0];
        vy[k] = 8.0f * f[1];
      } else {
        vx[k] = 8.0f * fx[yy * W + xx];
        vy[k] = 8.0f * fy[yy * W + xx];
      }
    }
  }
  f32x4 ox, oy;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) mx = fmaxf(mx, m[k][c]);
    float e[9], sum = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { e[k] = expf(m[k][c] - mx); sum += e[k]; }
    const float inv = 1.0f / sum;
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float wk = e[k] * inv;
      ax += wk * vx[k];
      ay += wk * vy[k];
    }
    ox[c] = ax;
    oy[c] = ay;
  }
  const int sy = idx >> 1, sx = (idx & 1) * 4;
  const long long HW8 = hw * 64;
  const long long o = (long long)(8 * y + sy) * (8 * W) + 8 * x + sx;
  *reinterpret_cast<f32x4*>(out + (b * 2 + 0) * HW8 + o) = ox;
  *reinterpret_cast<f32x4*>(out + (b * 2 + 1) * HW8 + o) = oy;
}

int launch_convex_upsample(const float* flow, int flow_ld, const float* mask, int mask_ld, float* out, long long M, int H, int W,
                           hipStream_t st) {
  if (pfk_aligned16(mask) && pfk_aligned16(out) && (mask_ld & 3) == 0) {
    const long long blocks = (M + 15) / 16;
    if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(convex_upsample4_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, flow, flow_ld, mask, mask_ld, out, M, H, W);
  } else {
    const long long blocks = (M + 3) / 4;
    if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)blocks), dim3(256), 0, st, flow, flow_ld, mask, mask_ld, out, M, H, W);
  }
  return pfk_launch_status();
}

// 32x32 LDS tile transposes between NCHW ([C][HW] per image) and pixel-major ([HW][ld]).
__global__ __launch_bounds__(256) void nchw_to_pm_kernel(const float* __restrict__ in,
                                                         float* __restrict__ out, int out_ld,
                                                         int out_coff, int C, int HW) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = in + (long long)b * C * HW;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    tile[i][tx] = (c < C && p < HW) ? src[(long long)c * HW + p] : 0.f;
  }
  __syncthreads();
  float* dst = out + (long long)b * HW * out_ld + out_coff;
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    if (p < HW && c < C) dst[(long long)p * out_ld + c] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void pm_to_nchw_kernel(const float* __restrict__ in, int in_ld,
                                                         int in_coff, float* __restrict__ out,
                                                         int C, int HW, int out_ld) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = in + (long long)b * HW * in_ld + in_coff;
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    tile[i][tx] = (p < HW && c < C) ? src[(long long)p * in_ld + c] : 0.f;
  }
  __syncthreads();
  float* dst = out + (long long)b * C * out_ld;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    if (c < C && p < HW) dst[(long long)c * out_ld + p] = tile[tx][i];
  }
}

// Warm start (RAFT's forward_interpolate, ptlflow/utils/external/raft.py:155-185): push every pixel along its flow, keep
// the landing points strictly inside the image, give every grid point the flow of its nearest landing point.  The
// reference does this per sample on the host (scipy griddata "nearest" = a cKDTree query in float64); here it is a
// brute-force float64 arg-min — N^2 = 49.6 M distance evaluations at 55x128, well under a millisecond — with the
// scattered points streamed through LDS (broadcast reads) and a strict "<" so exact ties go to the lowest source
// index.  This file is compiled with -ffp-contract=off: d2 = dx*dx + dy*dy rounds after every operation as numpy does.
__global__ __launch_bounds__(256) void forward_interp_kernel(const float* __restrict__ flow, float* __restrict__ out,
                                                             int H, int W) {
  __shared__ double sx[256], sy[256];
  const int N = H * W;
  const int b = blockIdx.y;
  const float* fx = flow + (long long)b * 2 * N;
  const float* fy = fx + N;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const double qx = (double)(q % W), qy = (double)(q / W);
  double best = 1.0e300;
  int arg = -1;
  for (int t0 = 0; t0 < N; t0 += 256) {
    const int j = t0 + threadIdx.x;
    double x1 = 1.0e200, y1 = 1.0e200;      // invalid / out-of-range sources sit infinitely far away
    if (j < N) {
      const double px = (double)(j % W) + (double)fx[j];
      const double py = (double)(j / W) + (double)fy[j];
      if (px > 0.0 && px < (double)W && py > 0.0 && py < (double)H) { x1 = px; y1 = py; }
    }
    sx[threadIdx.x] = x1;
    sy[threadIdx.x] = y1;
    __syncthreads();
    const int lim = min(256, N - t0);
    for (int k = 0; k < lim; ++k) {
      const double dx = sx[k] - qx, dy = sy[k] - qy;
      const double d2 = dx * dx + dy * dy;
      if (d2 < best) { best = d2; arg = t0 + k; }
    }
    __syncthreads();
  }
  if (q < N) {
    const bool hit = arg >= 0 && best < 1.0e299;
    out[((long long)b * 2 + 0) * N + q] = hit ? fx[arg] : 0.f;
    out[((long long)b * 2 + 1) * N + q] = hit ? fy[arg] : 0.f;
  }
}

// In-place softmax over the rows of a [rows][ld] matrix (cols used): GMA's attention map, gma_utils.py:75-76
// (`sim.softmax(dim=-1)` over N = h*w target pixels, once per forward).  One 256-thread block per row, three passes over the
// row (max, sum of exp, normalise); the row (28 KB at 55x128) stays in L2 between passes.  expf / division as torch's kernel.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long long ld, int cols) {
  __shared__ float red[4];
  float* row = x + (long long)blockIdx.x * ld;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  float m = -INFINITY;
  for (int c = t; c < cols; c += 256) m = fmaxf(m, row[c]);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
  if (lane == 0) red[wid] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = t; c < cols; c += 256) sum += expf(row[c] - m);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s, 64);
  if (lane == 0) red[wid] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  for (int c = t; c < cols; c += 256) row[c] = expf(row[c] - m) / sum;
}

}  // namespace

extern "C" {

int pfk_abi_version(void) { return PFK_ABI_VERSION; }

const char* pfk_status_string(int status) {
  switch (status) {
    case PFK_OK: return "ok";
    case PFK_ERR_BAD_ARG: return "bad argument";
    case PFK_ERR_ALIGNMENT: return "pointer/stride alignment";
    case PFK_ERR_UNSUPPORTED: return "unsupported shape";
    case PFK_ERR_LAUNCH: return "kernel launch failed";
    case PFK_ERR_DISABLED: return "debug knob disabled (set PFK_DEBUG_KNOBS=1 in the process environment to use pfk_debug_set_*)";
    default: return "unknown status";
  }
}

}  // extern "C"

template <typename TO>
static int conv_cin2_launch(const float* in, int in_ld, const float* weight, const float* bias,
                            TO* out, int out_ld, int out_coff, int B, int H, int W, int k, int cout,
                            int relu, pfk_stream_t stream) {
  if (!in || !weight || !out || B <= 0 || H <= 0 || W <= 0 || cout <= 0) return PFK_ERR_BAD_ARG;
  if (k <= 0 || !(k & 1) || in_ld < 2 || out_ld < out_coff + cout) return PFK_ERR_BAD_ARG;
  const long long M = (long long)B * H * W;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the MFMA kernel from one tile per CU up (batch 8 of 55x128: 33.1 -> 26.6 us, bf16 rows 31.0 -> 24.2); below that its per-workgroup
  // weight staging (50 KB) is not amortised (batch 1, 56 tiles: 12.3 us against 7.4 for the VALU kernel) — gpurun_out/r6r_cin2.log.
  // g_cin2_valu: 1 = always the VALU kernel, 2 = the MFMA kernel at every size (tests).  Same bits either way.
  const int c2_tx = (W + C2M_COLS - 1) / C2M_COLS, c2_ty = (H + C2M_ROWS - 1) / C2M_ROWS;
  if (k == 7 && (cout == 64 || cout == 128) && pfk_aligned16(weight) && g_cin2_valu != 1 &&
      ((long long)B * c2_tx * c2_ty >= 256 || g_cin2_valu == 2)) {
    const int tiles_x = c2_tx, tiles_y = c2_ty;
    const long long tiles = (long long)B * tiles_x * tiles_y;
    if (tiles > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);       // two workgroups per CU, persistent over the tile list
    if (cout == 128)
      hipLaunchKernelGGL((conv_cin2_mfma_kernel<4, TO>), dim3(grid), dim3(256), 0, st, in, in_ld, weight, bias, out, out_ld, out_coff, H, W,
                         tiles_x, tiles_y, (int)tiles, relu);
    else
      hipLaunchKernelGGL((conv_cin2_mfma_kernel<2, TO>), dim3(grid), dim3(256), 0, st, in, in_ld, weight, bias, out, out_ld, out_coff, H, W,
                         tiles_x, tiles_y, (int)tiles, relu);
    return pfk_launch_status();
  }
  if (k == 3 || k == 5 || k == 7) {
    const int tpr = (W + 15) / 16;
    const long long tiles = (long long)B * H * tpr;
    if (tiles > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
    dim3 grid((unsigned)tiles), block(256);
    if (k == 7) hipLaunchKernelGGL((conv_cin2_tiled_kernel<7, TO>), grid, block, 0, st, in, in_ld, weight, bias, out, out_ld, out_coff, H, W, tpr, cout, relu);
    else if (k == 5) hipLaunchKernelGGL((conv_cin2_tiled_kernel<5, TO>), grid, block, 0, st, in, in_ld, weight, bias, out, out_ld, out_coff, H, W, tpr, cout, relu);
    else hipLaunchKernelGGL((conv_cin2_tiled_kernel<3, TO>), grid, block, 0, st, in, in_ld, weight, bias, out, out_ld, out_coff, H, W, tpr, cout, relu);
    return pfk_launch_status();
  }
  if constexpr (sizeof(TO) == 4) {
    const long long blocks = (M * cout + 255) / 256;
    if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(conv_cin2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, in_ld, weight, bias, out, out_ld,
                       out_coff, M, H, W, k, cout, relu);
    return pfk_launch_status();
  } else {
    return PFK_ERR_UNSUPPORTED;      // 16-bit output: the tiled kernel's sizes only (k = 3, 5, 7)
  }
}

template <typename TI>
static int flow_delta_launch(const TI* in, int in_ld, int cin, const float* weight,
                             const float* bias, const float* coords0, float* coords1, float* delta_out,
                             float* flow_out, int flow_ld, __bf16* flow_b16, int flow_b16_ld, int B, int H, int W, pfk_stream_t stream);

extern "C" {

int pfk_conv_cin2_f32(const float* in, int in_ld, const float* weight, const float* bias,
                      float* out, int out_ld, int out_coff, int B, int H, int W, int k, int cout,
                      int relu, pfk_stream_t stream) {
  return conv_cin2_launch<float>(in, in_ld, weight, bias, out, out_ld, out_coff, B, H, W, k, cout, relu, stream);
}

int pfk_debug_set_cin2_valu(int on) {
  if (!pfk_debug_knobs_enabled()) return PFK_ERR_DISABLED;
  g_cin2_valu = (on == 1 || on == 2) ? on : 0;
  return PFK_OK;
}

int pfk_conv_cin2_b16(const float* in, int in_ld, const float* weight, const float* bias,
                      void* out_bf16, int out_ld, int out_coff, int B, int H, int W, int k, int cout,
                      int relu, pfk_stream_t stream) {
  return conv_cin2_launch<__bf16>(in, in_ld, weight, bias, static_cast<__bf16*>(out_bf16), out_ld, out_coff, B, H, W, k, cout, relu, stream);
}

}  // extern "C"

template <typename TI>
static int flow_delta_launch(const TI* in, int in_ld, int cin, const float* weight,
                             const float* bias, const float* coords0, float* coords1, float* delta_out,
                             float* flow_out, int flow_ld, __bf16* flow_b16, int flow_b16_ld, int B, int H, int W, pfk_stream_t stream) {
  if (!in || !weight || !coords0 || !coords1 || B <= 0 || H <= 0 || W <= 0) return PFK_ERR_BAD_ARG;
  if (cin <= 0 || in_ld < cin || (flow_out && flow_ld < 2) || (flow_b16 && flow_b16_ld < 2)) return PFK_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(in) & (sizeof(TI) * 4 - 1)) || !pfk_aligned16(weight) || (in_ld & 3) || (cin & 3))
    return PFK_ERR_ALIGNMENT;
  // 8 pixels per wave (fewest row loads) once that still gives every SIMD a few waves; 4 per wave below (batch 1: 7040 pixels)
  const long long rows = (long long)B * H;
  const int pixw = rows * ((W + 7) / 8) >= 4096 ? 8 : 4;
  const int tpr = (W + pixw - 1) / pixw;             // waves per image row
  const long long blocks = (rows * tpr + 3) / 4;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (pixw == 8)
    hipLaunchKernelGGL((flow_delta_kernel<8, TI>), dim3((unsigned)blocks), dim3(256), 0, st, in, in_ld, cin, weight, bias, coords0,
                       coords1, delta_out, flow_out, flow_ld, flow_b16, flow_b16_ld, rows, H, W, tpr);
  else
    hipLaunchKernelGGL((flow_delta_kernel<4, TI>), dim3((unsigned)blocks), dim3(256), 0, st, in, in_ld, cin, weight, bias, coords0,
                       coords1, delta_out, flow_out, flow_ld, flow_b16, flow_b16_ld, rows, H, W, tpr);
  return pfk_launch_status();
}

extern "C" {

int pfk_flow_delta_f32(const float* in, int in_ld, int cin, const float* weight,
                       const float* bias, const float* coords0, float* coords1, float* delta_out,
                       float* flow_out, int flow_ld, int B, int H, int W, pfk_stream_t stream) {
  return flow_delta_launch<float>(in, in_ld, cin, weight, bias, coords0, coords1, delta_out, flow_out, flow_ld, nullptr, 0, B, H, W, stream);
}

int pfk_flow_delta_b16(const void* in_bf16, int in_ld, int cin, const float* weight,
                       const float* bias, const float* coords0, float* coords1, float* delta_out,
                       float* flow_out, int flow_ld, void* flow_out_b16, int flow_b16_ld, int B, int H, int W, pfk_stream_t stream) {
  return flow_delta_launch<__bf16>(static_cast<const __bf16*>(in_bf16), in_ld, cin, weight, bias, coords0, coords1, delta_out, flow_out,
                                   flow_ld, static_cast<__bf16*>(flow_out_b16), flow_b16_ld, B, H, W, stream);
}

int pfk_flow_from_coords_f32(const float* coords0, const float* coords1, float* flow_out,
                             int flow_ld, int B, int H, int W, pfk_stream_t stream) {
  if (!coords0 || !coords1 || !flow_out || flow_ld < 2 || B <= 0 || H <= 0 || W <= 0)
    return PFK_ERR_BAD_ARG;
  const long long M = (long long)B * H * W;
  hipLaunchKernelGGL(flow_from_coords_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), coords0, coords1, flow_out, flow_ld, M,
                     (long long)H * W);
  return pfk_launch_status();
}

int pfk_convex_upsample_f32(const float* flow, const float* mask, int mask_ld, float* out, int B,
                            int H, int W, pfk_stream_t stream) {
  if (!flow || !mask || !out || mask_ld < 576 || B <= 0 || H <= 0 || W <= 0) return PFK_ERR_BAD_ARG;
  return launch_convex_upsample(flow, 0, mask, mask_ld, out, (long long)B * H * W, H, W, static_cast<hipStream_t>(stream));
}

int pfk_upflow8_f32(const float* coords0, const float* coords1, float* out, int B, int H, int W, pfk_stream_t stream) {
  if (!coords0 || !coords1 || !out || B <= 0 || H <= 0 || W <= 0) return PFK_ERR_BAD_ARG;
  const long long n = (long long)B * 64 * H * W;
  const long long blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(upflow8_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), coords0, coords1, out,
                     B, H, W);
  return pfk_launch_status();
}

int pfk_convex_upsample_pm_b16(const float* flow_pm, int flow_ld, const void* mask_bf16, int mask_ld, float* out,
                               int B, int H, int W, pfk_stream_t stream) {
  if (!flow_pm || !mask_bf16 || !out || B <= 0 || H <= 0 || W <= 0 || mask_ld < 576 || flow_ld < 2) return PFK_ERR_BAD_ARG;
  if ((reinterpret_cast<uintptr_t>(mask_bf16) & 7u) || !pfk_aligned16(out) || (mask_ld & 3)) return PFK_ERR_ALIGNMENT;
  const long long M = (long long)B * H * W;
  const long long blocks = (M + 15) / 16;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(convex_upsample4_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), flow_pm, flow_ld,
                     static_cast<const __bf16*>(mask_bf16), mask_ld, out, M, H, W);
  return pfk_launch_status();
}

int pfk_convex_upsample_pm_f32(const float* flow_pm, int flow_ld, const float* mask, int mask_ld, float* out,
                               int B, int H, int W, pfk_stream_t stream) {
  if (!flow_pm || !mask || !out || flow_ld < 2 || mask_ld < 576 || B <= 0 || H <= 0 || W <= 0) return PFK_ERR_BAD_ARG;
  return launch_convex_upsample(flow_pm, flow_ld, mask, mask_ld, out, (long long)B * H * W, H, W, static_cast<hipStream_t>(stream));
}

int pfk_nchw_to_pm_f32(const float* in, float* out, int out_ld, int out_coff, int B, int C, int H,
                       int W, pfk_stream_t stream) {
  if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || out_ld < out_coff + C)
    return PFK_ERR_BAD_ARG;
  const int HW = H * W;
  dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(nchw_to_pm_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     out, out_ld, out_coff, C, HW);
  return pfk_launch_status();
}

int pfk_pm_to_nchw_f32(const float* in, int in_ld, int in_coff, float* out, int B, int C, int H,
                       int W, pfk_stream_t stream) {
  if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || in_ld < in_coff + C)
    return PFK_ERR_BAD_ARG;
  const int HW = H * W;
  dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(pm_to_nchw_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     in_ld, in_coff, out, C, HW, HW);
  return pfk_launch_status();
}

int pfk_pm_to_cm_f32(const float* in, int in_ld, float* out, int out_ld, int B, int C, int N,
                     pfk_stream_t stream) {
  if (!in || !out || B <= 0 || C <= 0 || N <= 0 || in_ld < C || out_ld < N) return PFK_ERR_BAD_ARG;
  dim3 grid((N + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(pm_to_nchw_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     in_ld, 0, out, C, N, out_ld);
  return pfk_launch_status();
}

int pfk_softmax_rows_f32(float* x, long long rows, int cols, long long ld, pfk_stream_t stream) {
  if (!x || rows <= 0 || cols <= 0 || ld < cols || rows > 0x7fffffffLL) return PFK_ERR_BAD_ARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, cols);
  return pfk_launch_status();
}

int pfk_forward_interpolate_f32(const float* flow, float* out, int B, int H, int W, pfk_stream_t stream) {
  if (!flow || !out || B <= 0 || H <= 0 || W <= 0) return PFK_ERR_BAD_ARG;
  const long long N = (long long)H * W;
  if (N > 0x3fffffffLL) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(forward_interp_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), flow, out, H, W);
  return pfk_launch_status();
}

}  // extern "C"
