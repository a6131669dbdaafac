// Weight gradient of the implicit-GEMM convolution: the transposed product
//
//   dW[co][k] = sum_p dY[p][co] * A[p][k],     A[p][k = (source, tap, channel)] = src_s[p + tap][c]  (zero outside the image)
//
// on the fp32 matrix cores, straight from the pixel-major tensors of the forward pass — no im2col, no transposes.  The
// reduction index is the PIXEL: a K-step is 32 consecutive pixels, staged as they lie in memory (dY: 32 rows x 128 output
// channels, A: 32 rows x 32 input channels of one (source, tap, chunk), with the tap's zero padding done by out-of-range
// buffer offsets exactly as in the forward kernel).  v_mfma_f32_32x32x2_f32 wants A[i = co][k = pixel] and B[k = pixel][j = c]:
// lane l reads LDS element [pixel 2j + (l >> 5)][l & 31] with ds_read_b32 — each half-wave reads 32 consecutive floats of
// one row, conflict-free — one read pair per MFMA.
//
// Grid: (cout tiles) x (groups of K chunks) x (pixel splits); a tile is 32*CB output channels x 32*(4/CB) K columns, CB in
// {4, 2, 1} picked per launch to pad cout the least.  The output has few tiles and a very long reduction (zr of RAFT: 2 x 60
// tiles, 22 816+ pixels), so the pixel range is cut into `splits` slices that write partial tiles to a workspace; a second kernel
// adds the slices in a fixed order (deterministic, no atomics) into the packed [cout][ktot] layout of the forward weight.
// (A 128 x 128 tile with 64x64 wave tiles — half the LDS reads per MFMA — measured 5-30 % SLOWER on every training shape:
// the loop is not LDS-bound, fewer resident waves per SIMD hurt more.)
#include "pfk_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

struct WgradArgs {
  const float* src0; const float* src1; const float* src2;
  int ld0, ld1, ld2, ch0, ch1, ch2, nsrc;
  const float* dy; int dy_ld; int cout;
  int H, W, kh, kw;
  int Ho, Wo, stride;     // output grid (dY rows = B*Ho*Wo = M); output (yo, xo) reads input (yo*stride + dy, xo*stride + dx)
  long long M;
  float* part;            // [splits][cout][ktot]  (== the output when splits == 1)
  int ktot, chunks, tiles_m;   // ktot / chunks include the bias chunk when with_bias
  int with_bias;               // one extra 32-column chunk whose column 0 is sum_p dY[p][co] (A = a column of ones): the bias gradient
  long long px_per_split; // multiple of 32
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// One (source, tap, 32-channel) entry of the forward kernel's K enumeration.
struct ChunkRef { const float* src; int ld, cch, c0, dy, dx, bias, live; };

__device__ __forceinline__ ChunkRef chunk_ref(const WgradArgs& a, int chunk) {
  ChunkRef c;
  const int taps = a.kh * a.kw;
  int r = chunk, cps = (a.ch0 + 31) >> 5;
  c.src = a.src0; c.ld = a.ld0; c.cch = a.ch0;
  if (a.nsrc > 1 && r >= taps * cps) {
    r -= taps * cps; cps = (a.ch1 + 31) >> 5; c.src = a.src1; c.ld = a.ld1; c.cch = a.ch1;
    if (a.nsrc > 2 && r >= taps * cps) { r -= taps * cps; cps = (a.ch2 + 31) >> 5; c.src = a.src2; c.ld = a.ld2; c.cch = a.ch2; }
  }
  c.live = chunk < a.chunks;
  c.bias = a.with_bias && chunk == a.chunks - 1;
  const bool plain = c.live && !c.bias;
  const int tap = plain ? r / cps : 0;
  c.c0 = plain ? (r - tap * cps) * 32 : 0;
  c.dy = plain ? tap / a.kw - (a.kh >> 1) : 0;
  c.dx = plain ? tap % a.kw - (a.kw >> 1) : 0;
  return c;
}

// CB = 32-row output-channel blocks per tile (4, 2 or 1); the block's four waves cover CB co-blocks x NCH = 4 / CB consecutive K
// chunks, one 32x32 accumulator each, so the tile is (32 CB) output channels x (32 NCH) K columns.  The host picks CB to waste
// the fewest MFMAs on channel padding (cout = 64: CB 2, not half-empty 128-row tiles; 96 and 192: three exact 32- / 64-row tiles).
template <int CB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int NCH = 4 / CB, CO = 32 * CB, KC = 32 * NCH;
  __shared__ __attribute__((aligned(16))) float sY[2][32][CO];
  __shared__ __attribute__((aligned(16))) float sX[2][32][KC];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int groups = (a.chunks + NCH - 1) / NCH;
  const int grp = blockIdx.x % groups;
  const int tile_m = blockIdx.x / groups;
  const int split = blockIdx.y;
  const int co0 = tile_m * CO;

  const __amdgpu_buffer_rsrc_t rsy = rsrc_of(a.dy);
  const long long p_begin = (long long)split * a.px_per_split;
  const long long p_end = min(a.M, p_begin + a.px_per_split);
  const int steps = p_end > p_begin ? (int)((p_end - p_begin + 31) >> 5) : 0;

  // this thread stages row (t >> 3) of every step: float4 q of each of the NCH A chunks, float4 q + 8*i of the dY tile
  const int row = t >> 3, q = t & 7;
  long long p = p_begin + row;                     // output pixel (= dY row); its input pixel is (b, y*stride, x*stride)
  int x = (int)(p % a.Wo), y = (int)((p / a.Wo) % a.Ho), bimg = (int)(p / ((long long)a.Wo * a.Ho));
  const int sd = a.stride;
  __amdgpu_buffer_rsrc_t rsx[NCH];
  int tap_off[NCH], cdy[NCH], cdx[NCH], cld[NCH];
  bool c_ok[NCH], is_bias[NCH], yok[CB];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const ChunkRef c = chunk_ref(a, grp * NCH + i);
    rsx[i] = rsrc_of(c.src);
    tap_off[i] = (c.dy * a.W + c.dx) * c.ld + c.c0 + q * 4;
    cdy[i] = c.dy; cdx[i] = c.dx; cld[i] = c.ld;
    c_ok[i] = c.live && !c.bias && c.c0 + q * 4 < c.cch;
    is_bias[i] = c.bias;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) yok[i] = co0 + q * 4 + 32 * i < a.cout;

  u32x4 rx[NCH], ry[CB];
  auto load = [&](void) {
    const bool in = p < p_end;
    const int yi = y * sd, xi = x * sd;
    const int pin = (bimg * a.H + yi) * a.W + xi;   // input pixel index (== p for stride 1)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const bool ok = in && c_ok[i] && (unsigned)(yi + cdy[i]) < (unsigned)a.H && (unsigned)(xi + cdx[i]) < (unsigned)a.W;
      if (is_bias[i]) rx[i] = u32x4{(in && q == 0) ? 0x3f800000u : 0u, 0u, 0u, 0u};    // A = [1 0 0 ...] for every live pixel
      else rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsx[i], ok ? (unsigned)(pin * cld[i] + tap_off[i]) * 4u : OOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i)
      ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rsy, (in && yok[i]) ? (unsigned)((int)p * a.dy_ld + co0 + q * 4 + 32 * i) * 4u : OOB, 0, 0);
    p += 32;
    x += 32;
    while (x >= a.Wo) { x -= a.Wo; if (++y == a.Ho) { y = 0; ++bimg; } }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(&sX[buf][row][q * 4 + 32 * i]) = rx[i];
#pragma unroll
    for (int i = 0; i < CB; ++i) *reinterpret_cast<u32x4*>(&sY[buf][row][q * 4 + 32 * i]) = ry[i];
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  if (steps > 0) {
    load();
    store(0);
  }
  __syncthreads();
  const int l31 = lane & 31, hl = lane >> 5;
  const int wco = (wid % CB) * 32, wkc = (wid / CB) * 32;      // this wave's co block / chunk inside the tile
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < steps;
    if (more) load();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float av = sY[buf][2 * j + hl][wco + l31];
      const float bv = sX[buf][2 * j + hl][wkc + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (more) store(buf ^ 1);
    __syncthreads();
  }

  // D: column j = lane & 31 (input channel), rows i = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (output channel)
  const int chunk = grp * NCH + wid / CB;
  if (chunk >= a.chunks) return;
  float* out = a.part + (long long)split * a.cout * a.ktot;
  const int kcol = chunk * 32 + l31;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = co0 + wco + (e & 3) + 8 * (e >> 2) + 4 * hl;
    if (co < a.cout) out[(long long)co * a.ktot + kcol] = acc[e];
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           long long n, int splits) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(part + i);
  for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(part + (long long)k * n + i);
  *reinterpret_cast<f32x4*>(out + i) = s;
}

// The same slice reduction, written straight into PyTorch's layouts: dw [cout][cin][kh][kw] (cin = the sources' REAL channels
// one after the other) and db [cout] — the inverse of the host's weight packing (ptlflow_amd/packing.py), so a training step
// needs no per-call permute / copy kernels between the weight-gradient launch and autograd's accumulation.
struct UnpackArgs {
  const float* part; long long n; int splits;
  int ktot, cout_real, taps, nseg, cin_total, with_bias, accumulate;
  int first[3], nreal[3], cpad[3];
  float* dw; float* db;
};

__global__ __launch_bounds__(256) void wgrad_reduce_unpack_kernel(const UnpackArgs u) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)u.cout_real * u.ktot) return;
  const int co = (int)(idx / u.ktot), k = (int)(idx - (long long)co * u.ktot);
  float s = u.part[idx];
  for (int j = 1; j < u.splits; ++j) s += u.part[(long long)j * u.n + idx];      // same fixed order as wgrad_reduce_kernel
  int kk = k;
  if (u.with_bias && k >= u.ktot - 32) {
    if (k == u.ktot - 32 && u.db) u.db[co] = u.accumulate ? u.db[co] + s : s;
    return;
  }
  int sg = 0;
  while (sg < u.nseg - 1 && kk >= u.taps * u.cpad[sg]) { kk -= u.taps * u.cpad[sg]; ++sg; }
  const int tap = kk / u.cpad[sg], c = kk - tap * u.cpad[sg];
  if (c < u.nreal[sg]) {
    float* dst = u.dw + ((long long)co * u.cin_total + u.first[sg] + c) * u.taps + tap;
    *dst = u.accumulate ? *dst + s : s;     // (one thread per element: the running sum's order is the order of the calls)
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Gate arithmetic of one (Sep)ConvGRU pass for the training path (raft/update.py:24-32, 58-73), float4 per thread over
// pixel-major [M][C] tensors.  Forward keeps what the backward needs (z, r, q); the derivative kernels produce the
// pre-activation gradients the dgrad / wgrad convolutions consume.
//   f1: z = s(a_z), r = s(a_r), rh = r*h               f2: q = tanh(a_q), h' = (1-z)*h + z*q
//   b1: da_q = dh'*z*(1-q^2), da_z = dh'*(q-h)*z*(1-z), dh = dh'*(1-z)
//   b2: da_r = d(rh)*h*r*(1-r), dh += d(rh)*r
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 sigmoid4(f32x4 v) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = 1.0f / (1.0f + expf(-v[e]));
  return o;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__global__ __launch_bounds__(256) void gru_f1_kernel(const float* __restrict__ azr, const float* __restrict__ h, int h_ld,
                                                     float* __restrict__ z, float* __restrict__ r, float* __restrict__ rh,
                                                     long long M, int C) {
  const int tpr = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, p = idx / tpr;
  if (p >= M) return;
  const int c = (int)(idx - p * tpr) * 4;
  const f32x4 zz = sigmoid4(ld4(azr + p * 2 * C + c)), rr = sigmoid4(ld4(azr + p * 2 * C + C + c));
  st4(z + p * C + c, zz);
  st4(r + p * C + c, rr);
  st4(rh + p * C + c, rr * ld4(h + p * h_ld + c));
}

__global__ __launch_bounds__(256) void gru_f2_kernel(const float* __restrict__ aq, const float* __restrict__ z,
                                                     const float* __restrict__ h, int h_ld, float* __restrict__ q,
                                                     float* __restrict__ hn, long long M, int C) {
  const int tpr = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, p = idx / tpr;
  if (p >= M) return;
  const int c = (int)(idx - p * tpr) * 4;
  const f32x4 a = ld4(aq + p * C + c), zz = ld4(z + p * C + c), hh = ld4(h + p * h_ld + c);
  f32x4 qq;
#pragma unroll
  for (int e = 0; e < 4; ++e) qq[e] = tanhf(a[e]);
  st4(q + p * C + c, qq);
  st4(hn + p * C + c, (1.0f - zz) * hh + zz * qq);
}

__global__ __launch_bounds__(256) void gru_b1_kernel(const float* __restrict__ dhn, int dhn_ld, const float* __restrict__ z,
                                                     const float* __restrict__ q, const float* __restrict__ h, int h_ld,
                                                     float* __restrict__ daq, float* __restrict__ dazr, float* __restrict__ dh,
                                                     long long M, int C) {
  const int tpr = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, p = idx / tpr;
  if (p >= M) return;
  const int c = (int)(idx - p * tpr) * 4;
  const f32x4 g = ld4(dhn + p * dhn_ld + c), zz = ld4(z + p * C + c), qq = ld4(q + p * C + c), hh = ld4(h + p * h_ld + c);
  st4(daq + p * C + c, g * zz * (1.0f - qq * qq));
  st4(dazr + p * 2 * C + c, g * (qq - hh) * zz * (1.0f - zz));
  st4(dh + p * C + c, g * (1.0f - zz));
}

__global__ __launch_bounds__(256) void gru_b2_kernel(const float* __restrict__ drh, const float* __restrict__ h, int h_ld,
                                                     const float* __restrict__ r, float* __restrict__ dazr, float* __restrict__ dh,
                                                     long long M, int C) {
  const int tpr = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, p = idx / tpr;
  if (p >= M) return;
  const int c = (int)(idx - p * tpr) * 4;
  const f32x4 g = ld4(drh + p * C + c), hh = ld4(h + p * h_ld + c), rr = ld4(r + p * C + c);
  st4(dazr + p * 2 * C + C + c, g * hh * rr * (1.0f - rr));
  st4(dh + p * C + c, ld4(dh + p * C + c) + g * rr);
}

int g_wgrad_variant = 0;   // 0: tile height by padding waste (pick_cb); 1 / 2 / 4: forced CB — pfk_debug_set_wgrad

// 32-row output-channel blocks per tile: the tallest tile among {4, 2, 1} blocks that pads cout the least
int pick_cb(int cout) {
  if (g_wgrad_variant == 1 || g_wgrad_variant == 2 || g_wgrad_variant == 4) return g_wgrad_variant;
  int best = 4, waste = (cout + 127) / 128 * 128 - cout;
  const int w2 = (cout + 63) / 64 * 64 - cout, w1 = (cout + 31) / 32 * 32 - cout;
  if (w2 < waste) { best = 2; waste = w2; }
  if (w1 < waste) best = 1;
  return best;
}

// blocks of the main launch for a tile height: (cout tiles) x (K chunk groups)
long long wgrad_tiles(int cout, int chunks, int cb) { return (long long)((cout + 32 * cb - 1) / (32 * cb)) * ((chunks + 4 / cb - 1) / (4 / cb)); }

int pick_splits(long long tiles, long long M, long long target = 1024) {
  long long s = target / (tiles > 0 ? tiles : 1);
  const long long max_s = M / 512 > 0 ? M / 512 : 1;     // >= 16 K-steps per slice
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

inline int ktot_of(const pfk_conv_desc* d) {
  int k = 0;
  for (int s = 0; s < d->num_src; ++s) k += d->kh * d->kw * ((d->src[s].channels + 31) / 32 * 32);
  return k;
}

}  // namespace

extern "C" {

int pfk_debug_set_wgrad(int variant) { if (!pfk_debug_knobs_enabled()) return PFK_ERR_DISABLED; g_wgrad_variant = variant; return PFK_OK; }

long long pfk_conv_wgrad_workspace_bytes(const pfk_conv_desc* d, int with_bias) {
  if (!d || d->num_src < 1 || d->num_src > 3 || d->cout <= 0) return 0;
  const int ktot = ktot_of(d) + (with_bias ? 32 : 0);
  const int sd = d->stride > 1 ? d->stride : 1;
  const long long M = (long long)d->B * ((d->H - 1) / sd + 1) * ((d->W - 1) / sd + 1);
  int splits = 1;
  for (int cb = 1; cb <= 4; cb *= 2) {        // any tile height may run (pfk_debug_set_wgrad): size for the largest split count
    const int sc = pick_splits(wgrad_tiles(d->cout, ktot / 32, cb), M);
    splits = sc > splits ? sc : splits;
  }
  return splits > 1 ? (long long)splits * d->cout * ktot * (long long)sizeof(float) : 0;
}

// Validate, fill the kernel arguments (everything but `part`) and pick the split count.
static int wgrad_plan(const pfk_conv_desc* d, const float* dy, int dy_ld, int with_bias, WgradArgs& a, int& cb, int& groups, int& splits) {
  if (!d || !dy || d->num_src < 1 || d->num_src > 3) return PFK_ERR_BAD_ARG;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cout <= 0 || dy_ld < d->cout) return PFK_ERR_BAD_ARG;
  if (d->kh <= 0 || d->kw <= 0 || !(d->kh & 1) || !(d->kw & 1) || d->stride < 0) return PFK_ERR_BAD_ARG;
  if ((d->cout & 3) || (dy_ld & 3) || !pfk_aligned16(dy)) return PFK_ERR_ALIGNMENT;
  const pfk_conv_src* s = d->src;
  const int sd = d->stride > 1 ? d->stride : 1;
  const int Ho = (d->H - 1) / sd + 1, Wo = (d->W - 1) / sd + 1;
  const long long M = (long long)d->B * Ho * Wo;                 // dY rows: the OUTPUT grid
  const long long Min = (long long)d->B * d->H * d->W;           // source rows
  for (int i = 0; i < d->num_src; ++i) {
    if (!s[i].ptr || s[i].channels <= 0 || s[i].ld < s[i].channels) return PFK_ERR_BAD_ARG;
    if (!pfk_aligned16(s[i].ptr) || (s[i].ld & 3) || (s[i].channels & 3)) return PFK_ERR_ALIGNMENT;
    if (Min * s[i].ld * 4 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  }
  if (M * dy_ld * 4 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  a = WgradArgs{};
  a.src0 = s[0].ptr; a.ld0 = s[0].ld; a.ch0 = s[0].channels;
  if (d->num_src > 1) { a.src1 = s[1].ptr; a.ld1 = s[1].ld; a.ch1 = s[1].channels; }
  if (d->num_src > 2) { a.src2 = s[2].ptr; a.ld2 = s[2].ld; a.ch2 = s[2].channels; }
  a.nsrc = d->num_src;
  a.dy = dy; a.dy_ld = dy_ld; a.cout = d->cout;
  a.H = d->H; a.W = d->W; a.kh = d->kh; a.kw = d->kw; a.M = M;
  a.Ho = Ho; a.Wo = Wo; a.stride = sd;
  a.with_bias = with_bias ? 1 : 0;
  a.ktot = ktot_of(d) + (with_bias ? 32 : 0);
  a.chunks = a.ktot / 32;
  cb = pick_cb(d->cout);
  a.tiles_m = (d->cout + 32 * cb - 1) / (32 * cb);
  groups = (a.chunks + 4 / cb - 1) / (4 / cb);
  splits = pick_splits((long long)a.tiles_m * groups, M);
  a.px_per_split = ((M + splits - 1) / splits + 31) / 32 * 32;
  return PFK_OK;
}

static void wgrad_run(const WgradArgs& a, int cb, int groups, int splits, hipStream_t st) {
  const dim3 grid((unsigned)(a.tiles_m * groups), (unsigned)splits);
  if (cb == 4) hipLaunchKernelGGL(conv_wgrad_kernel<4>, grid, dim3(256), 0, st, a);
  else if (cb == 2) hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, st, a);
}

int pfk_conv_wgrad_f32(const pfk_conv_desc* d, const float* dy, int dy_ld, float* dw_packed, int with_bias, void* workspace,
                       long long workspace_bytes, pfk_stream_t stream) {
  if (!dw_packed) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(dw_packed)) return PFK_ERR_ALIGNMENT;
  WgradArgs a;
  int cb, groups, splits;
  const int rc = wgrad_plan(d, dy, dy_ld, with_bias, a, cb, groups, splits);
  if (rc != PFK_OK) return rc;
  const long long n = (long long)d->cout * a.ktot;
  if (splits > 1) {
    if (!workspace || workspace_bytes < splits * n * (long long)sizeof(float) || !pfk_aligned16(workspace)) return PFK_ERR_BAD_ARG;
    a.part = static_cast<float*>(workspace);
  } else {
    a.part = dw_packed;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  wgrad_run(a, cb, groups, splits, st);
  if (splits > 1)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st,
                       static_cast<const float*>(workspace), dw_packed, n, splits);
  return pfk_launch_status();
}

long long pfk_conv_wgrad_unpacked_workspace_bytes(const pfk_conv_desc* d, int with_bias) {
  if (!d || d->num_src < 1 || d->num_src > 3 || d->cout <= 0) return 0;
  const long long packed = pfk_conv_wgrad_workspace_bytes(d, with_bias);
  const long long one = (long long)d->cout * (ktot_of(d) + (with_bias ? 32 : 0)) * (long long)sizeof(float);
  return packed > one ? packed : one;       // the slices always go through the workspace here, even a single one
}

int pfk_conv_wgrad_unpacked_f32(const pfk_conv_desc* d, const int* real_channels, const float* dy, int dy_ld, int cout_real,
                                float* dw, float* db, int accumulate, void* workspace, long long workspace_bytes, pfk_stream_t stream) {
  if (!dw || !real_channels || !workspace || cout_real <= 0 || !d || cout_real > d->cout) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(workspace)) return PFK_ERR_ALIGNMENT;
  const int with_bias = db != nullptr;
  WgradArgs a;
  int cb, groups, splits;
  const int rc = wgrad_plan(d, dy, dy_ld, with_bias, a, cb, groups, splits);
  if (rc != PFK_OK) return rc;
  const long long n = (long long)d->cout * a.ktot;
  if (workspace_bytes < splits * n * (long long)sizeof(float)) return PFK_ERR_BAD_ARG;
  a.part = static_cast<float*>(workspace);
  UnpackArgs u{};
  u.part = a.part; u.n = n; u.splits = splits; u.ktot = a.ktot; u.cout_real = cout_real; u.taps = d->kh * d->kw;
  u.nseg = d->num_src; u.with_bias = with_bias; u.dw = dw; u.db = db; u.accumulate = accumulate != 0;
  int first = 0;
  for (int i = 0; i < d->num_src; ++i) {
    if (real_channels[i] <= 0 || real_channels[i] > d->src[i].channels) return PFK_ERR_BAD_ARG;
    u.first[i] = first; u.nreal[i] = real_channels[i]; u.cpad[i] = (d->src[i].channels + 31) / 32 * 32;
    first += real_channels[i];
  }
  u.cin_total = first;
  hipStream_t st = static_cast<hipStream_t>(stream);
  wgrad_run(a, cb, groups, splits, st);
  const long long total = (long long)cout_real * a.ktot;
  hipLaunchKernelGGL(wgrad_reduce_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, u);
  return pfk_launch_status();
}

static int gru_grid(long long M, int C, dim3* grid) {
  if (M <= 0 || C <= 0 || (C & 3)) return PFK_ERR_BAD_ARG;
  const long long blocks = (M * (C >> 2) + 255) / 256;
  if (blocks > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  *grid = dim3((unsigned)blocks);
  return PFK_OK;
}

int pfk_gru_gates_zr_f32(const float* a_zr, const float* h, int h_ld, float* z, float* r, float* rh, long long M, int C,
                         pfk_stream_t stream) {
  dim3 grid;
  if (!a_zr || !h || !z || !r || !rh || (h_ld & 3) || h_ld < C) return PFK_ERR_BAD_ARG;
  const int rc = gru_grid(M, C, &grid);
  if (rc != PFK_OK) return rc;
  hipLaunchKernelGGL(gru_f1_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a_zr, h, h_ld, z, r, rh, M, C);
  return pfk_launch_status();
}

int pfk_gru_gates_q_f32(const float* a_q, const float* z, const float* h, int h_ld, float* q, float* h_new, long long M, int C,
                        pfk_stream_t stream) {
  dim3 grid;
  if (!a_q || !z || !h || !q || !h_new || (h_ld & 3) || h_ld < C) return PFK_ERR_BAD_ARG;
  const int rc = gru_grid(M, C, &grid);
  if (rc != PFK_OK) return rc;
  hipLaunchKernelGGL(gru_f2_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a_q, z, h, h_ld, q, h_new, M, C);
  return pfk_launch_status();
}

int pfk_gru_backward_q_f32(const float* dh_new, int dh_new_ld, const float* z, const float* q, const float* h, int h_ld,
                           float* da_q, float* da_zr, float* dh, long long M, int C, pfk_stream_t stream) {
  dim3 grid;
  if (!dh_new || !z || !q || !h || !da_q || !da_zr || !dh || (h_ld & 3) || (dh_new_ld & 3)) return PFK_ERR_BAD_ARG;
  const int rc = gru_grid(M, C, &grid);
  if (rc != PFK_OK) return rc;
  hipLaunchKernelGGL(gru_b1_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), dh_new, dh_new_ld, z, q, h, h_ld, da_q,
                     da_zr, dh, M, C);
  return pfk_launch_status();
}

int pfk_gru_backward_zr_f32(const float* d_rh, const float* h, int h_ld, const float* r, float* da_zr, float* dh, long long M,
                            int C, pfk_stream_t stream) {
  dim3 grid;
  if (!d_rh || !h || !r || !da_zr || !dh || (h_ld & 3)) return PFK_ERR_BAD_ARG;
  const int rc = gru_grid(M, C, &grid);
  if (rc != PFK_OK) return rc;
  hipLaunchKernelGGL(gru_b2_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), d_rh, h, h_ld, r, da_zr, dh, M, C);
  return pfk_launch_status();
}

}  // extern "C"
