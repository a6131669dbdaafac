// K1 / K4-K6: implicit-GEMM convolution and all-pairs correlation on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate, 64 cycles / SIMD).
//
//   out[p][n] = epilogue( bias[n] + sum_k A[p][k] * Wt[n][k] )
//
// A is never materialised: row p of the K-step (source s, tap (dy,dx), channel chunk c0) is
// src_s[p + dy*W + dx][c0 .. c0+31], or zeros when the tap falls outside the image (zero
// padding) — a K-step is 32 channels = one 128-byte row segment per pixel, so a wave's staging
// load is 8 pixels x 128 B, fully coalesced.  Both operands are "K-contiguous" (NT GEMM), are
// staged global -> VGPR -> LDS as float4 with the next K-step's loads in flight while the
// current one is on the matrix pipe, and live in LDS as [rows][36] floats: the 4-float pad
// makes the per-lane ds_read_b128 fragment reads (row = lane & 31, 4 consecutive k) hit 16
// distinct 16-byte slots per 16-lane group, i.e. conflict-free (MI355X_MICROARCH.md, LDS).
//
// Fragment use: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31] for one
// 32x32x2 MFMA.  One ds_read_b128 gives a lane 4 consecutive k for its row; sub-step s of the
// 4 MFMAs that follow consumes component s, i.e. k = {kk*8 + s (lanes 0-31), kk*8 + 4 + s
// (lanes 32-63)}.  A and B use the same k permutation, so the sum over k is unchanged.
// D layout: col = l & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(l >> 5).
//
// Epilogues fuse what the reference does in separate torch ops (raft/update.py:58-73 etc.):
// bias, relu, scale, sigmoid gates + r*h, tanh + GRU blend, and write straight into channel
// slices of pixel-major buffers (no torch.cat).
#include "pfk_common.h"

namespace {

constexpr int BK = 32;       // channels per K-step
constexpr int LDS_LD = 36;   // padded LDS row length (floats)

struct GemmArgs {
  const float* src0; const float* src1; const float* src2;
  int ld0, ld1, ld2;
  int ch0, ch1, ch2;
  int nsrc;
  int H, W;            // image dims for tap bounds (M = B*H*W rows, batch folded into M)
  int kh, kw;
  const float* weight; // [b_rows][ktot]
  const float* bias;
  int b_rows;          // valid rows of weight (= cout)
  int ktot;
  int relu;
  float scale;
  float* out; int out_ld; int out_coff;
  float* h; int h_ld;
  float* aux_z; float* aux_rh;
  int ch_hidden;       // Ch for the GRU epilogues
  long long M;
  long long a_bs, b_bs, o_bs;  // per-blockIdx.y strides (batched correlation), floats
  int tiles_n;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const GemmArgs a) {
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int A_PT = BM / 32;  // staging rows per thread (256 threads cover 32 rows x 8 float4)
  constexpr int B_PT = BN / 32;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                          // [2][BM][LDS_LD]
  float* sB = smem + 2 * BM * LDS_LD;        // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm0 = (wid / WAVES_N) * WM;
  const int wn0 = (wid % WAVES_N) * WN;

  const int bid = pfk_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n;
  const int tile_m = bid / a.tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;
  const long long batch = blockIdx.y;

  const float* src0 = a.src0 + batch * a.a_bs;
  const float* wgt = a.weight + batch * a.b_bs;

  // ---- per-thread staging coordinates -------------------------------------------------------
  const int c4 = (tid & 7) * 4;  // float offset inside the 32-float K chunk
  const int r0 = tid >> 3;       // 0..31
  long long prow[A_PT];
  int py[A_PT], px[A_PT];
  bool pok[A_PT];
#pragma unroll
  for (int i = 0; i < A_PT; ++i) {
    const long long p = m0 + r0 + 32 * i;
    pok[i] = p < a.M;
    prow[i] = p;
    px[i] = (int)(p % a.W);
    py[i] = (int)((p / a.W) % a.H);
  }
  const float* wrow[B_PT];
  bool wok[B_PT];
#pragma unroll
  for (int i = 0; i < B_PT; ++i) {
    const int n = n0 + r0 + 32 * i;
    wok[i] = n < a.b_rows;
    wrow[i] = wgt + (long long)(wok[i] ? n : 0) * a.ktot + c4;
  }

  // ---- K-step iterator: source -> tap (ky, kx) -> 32-channel chunk ----------------------------
  int seg = 0, ky = 0, kx = 0, c0 = 0, kofs = 0;
  const int ph = a.kh >> 1, pw = a.kw >> 1;
  int total_steps = 0;
  {
    const int taps = a.kh * a.kw;
    total_steps += taps * ((a.ch0 + BK - 1) / BK);
    if (a.nsrc > 1) total_steps += taps * ((a.ch1 + BK - 1) / BK);
    if (a.nsrc > 2) total_steps += taps * ((a.ch2 + BK - 1) / BK);
  }

  f32x4 ra[A_PT], rb[B_PT];

  auto load_step = [&]() {
    const float* sp = seg == 0 ? src0 : (seg == 1 ? a.src1 : a.src2);
    const int sld = seg == 0 ? a.ld0 : (seg == 1 ? a.ld1 : a.ld2);
    const int sch = seg == 0 ? a.ch0 : (seg == 1 ? a.ch1 : a.ch2);
    const int dy = ky - ph, dx = kx - pw;
    const bool cok = (c0 + c4) < sch;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const bool ok = pok[i] && cok && (unsigned)(py[i] + dy) < (unsigned)a.H &&
                      (unsigned)(px[i] + dx) < (unsigned)a.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const float* g = sp + (prow[i] + (long long)dy * a.W + dx) * sld + c0 + c4;
        v = *reinterpret_cast<const f32x4*>(g);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (wok[i]) v = *reinterpret_cast<const f32x4*>(wrow[i] + kofs);
      rb[i] = v;
    }
  };
  auto advance = [&]() {
    const int sch = seg == 0 ? a.ch0 : (seg == 1 ? a.ch1 : a.ch2);
    kofs += BK;
    c0 += BK;
    if (c0 >= sch) {
      c0 = 0;
      if (++kx == a.kw) {
        kx = 0;
        if (++ky == a.kh) { ky = 0; ++seg; }
      }
    }
  };
  auto store_lds = [&](int buf) {
    float* dA = sA + buf * BM * LDS_LD;
    float* dB = sB + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < A_PT; ++i)
      *reinterpret_cast<f32x4*>(dA + (r0 + 32 * i) * LDS_LD + c4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PT; ++i)
      *reinterpret_cast<f32x4*>(dB + (r0 + 32 * i) * LDS_LD + c4) = rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  load_step();
  advance();
  store_lds(0);
  __syncthreads();

  const int frow = lane & 31;
  const int fk = (lane >> 5) * 4;

  for (int step = 0; step < total_steps; ++step) {
    const int buf = step & 1;
    const bool more = (step + 1) < total_steps;
    if (more) { load_step(); advance(); }

    const float* cA = sA + buf * BM * LDS_LD + (wm0 + frow) * LDS_LD + fk;
    const float* cB = sB + buf * BN * LDS_LD + (wn0 + frow) * LDS_LD + fk;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 fa[MT], fb[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        fa[mt] = *reinterpret_cast<const f32x4*>(cA + mt * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        fb[nt] = *reinterpret_cast<const f32x4*>(cB + nt * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][s], fb[nt][s], acc[mt][nt], 0, 0, 0);
    }

    if (more) store_lds(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------
  const int col_l = lane & 31;
  const int row_l = (lane >> 5) * 4;
  float* outp = a.out + batch * a.o_bs;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + wn0 + nt * 32 + col_l;
    const bool nok = n < a.b_rows;
    const float bias = (a.bias != nullptr && nok) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long p = m0 + wm0 + mt * 32 + (r & 3) + 8 * (r >> 2) + row_l;
        if (!nok || p >= a.M) continue;
        float v = acc[mt][nt][r] + bias;
        if constexpr (EPI == PFK_EPI_LINEAR) {
          if (a.relu) v = (v < 0.f) ? 0.f : v;  // NaN-propagating like torch.relu (fmaxf would drop NaN)
          v *= a.scale;
          outp[p * a.out_ld + a.out_coff + n] = v;
        } else if constexpr (EPI == PFK_EPI_GRU_ZR) {
          const int ch = a.ch_hidden;
          const float g = sigmoid_f(v);
          if (n < ch) {
            a.aux_z[p * ch + n] = g;
          } else {
            const int c = n - ch;
            a.aux_rh[p * ch + c] = g * a.h[p * a.h_ld + c];
          }
        } else {  // PFK_EPI_GRU_Q
          const int ch = a.ch_hidden;
          const float q = tanhf(v);
          const float z = a.aux_z[p * ch + n];
          const float hv = a.h[p * a.h_ld + n];
          // (1 - z) * h + z * q, each product rounded (no contraction), as update.py:64,71
          const float t0 = __fmul_rn(__fsub_rn(1.0f, z), hv);
          const float t1 = __fmul_rn(z, q);
          a.h[p * a.h_ld + n] = __fadd_rn(t0, t1);
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmArgs& a, int epi, int batches, hipStream_t st) {
  GemmArgs g = a;
  const long long tiles_m = (a.M + BM - 1) / BM;
  g.tiles_n = (a.b_rows + BN - 1) / BN;
  const long long nblk = tiles_m * g.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  const size_t smem = 2 * (BM + BN) * LDS_LD * sizeof(float);
  dim3 grid((unsigned)nblk, (unsigned)batches), block(256);
  switch (epi) {
    case PFK_EPI_LINEAR:
      hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PFK_EPI_LINEAR>), grid, block, smem, st, g);
      break;
    case PFK_EPI_GRU_ZR:
      hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PFK_EPI_GRU_ZR>), grid, block, smem, st, g);
      break;
    case PFK_EPI_GRU_Q:
      hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PFK_EPI_GRU_Q>), grid, block, smem, st, g);
      break;
    default:
      return PFK_ERR_BAD_ARG;
  }
  return pfk_launch_status();
}

int g_force_tile = -1;  // debug/tuning knob, see pfk_debug_set_tile

int launch(const GemmArgs& a, int epi, int batches, hipStream_t st) {
  int cfg;
  if (g_force_tile >= 0) cfg = g_force_tile;
  else if (a.b_rows >= 2048 && a.M >= 2048) cfg = 2;  // big square-ish GEMM (correlation volume)
  else if (a.b_rows > 64) cfg = 1;
  else cfg = 0;
  switch (cfg) {
    case 0: return launch_cfg<64, 64, 32, 32>(a, epi, batches, st);
    case 1: return launch_cfg<64, 128, 32, 64>(a, epi, batches, st);
    case 2: return launch_cfg<128, 128, 64, 64>(a, epi, batches, st);
    case 3: return launch_cfg<128, 64, 64, 32>(a, epi, batches, st);
    default: return PFK_ERR_BAD_ARG;
  }
}

inline int round_up32(int c) { return (c + 31) & ~31; }

}  // namespace

extern "C" {

void pfk_debug_set_tile(int cfg) { g_force_tile = cfg; }

int pfk_conv_ktot(const pfk_conv_desc* d) {
  if (!d || d->num_src < 1 || d->num_src > 3) return PFK_ERR_BAD_ARG;
  int k = 0;
  for (int s = 0; s < d->num_src; ++s) k += d->kh * d->kw * round_up32(d->src[s].channels);
  return k;
}

int pfk_conv2d_f32(const pfk_conv_desc* d, pfk_stream_t stream) {
  if (!d || d->num_src < 1 || d->num_src > 3 || !d->weight) return PFK_ERR_BAD_ARG;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cout <= 0) return PFK_ERR_BAD_ARG;
  if (d->kh <= 0 || d->kw <= 0 || !(d->kh & 1) || !(d->kw & 1)) return PFK_ERR_BAD_ARG;
  GemmArgs a{};
  const pfk_conv_src* s = d->src;
  for (int i = 0; i < d->num_src; ++i) {
    if (!s[i].ptr || s[i].channels <= 0 || s[i].ld < s[i].channels) return PFK_ERR_BAD_ARG;
    if (!pfk_aligned16(s[i].ptr) || (s[i].ld & 3) || (s[i].channels & 3)) return PFK_ERR_ALIGNMENT;
  }
  if (!pfk_aligned16(d->weight)) return PFK_ERR_ALIGNMENT;
  a.src0 = s[0].ptr; a.ld0 = s[0].ld; a.ch0 = s[0].channels;
  if (d->num_src > 1) { a.src1 = s[1].ptr; a.ld1 = s[1].ld; a.ch1 = s[1].channels; }
  if (d->num_src > 2) { a.src2 = s[2].ptr; a.ld2 = s[2].ld; a.ch2 = s[2].channels; }
  a.nsrc = d->num_src;
  a.H = d->H; a.W = d->W; a.kh = d->kh; a.kw = d->kw;
  a.weight = d->weight; a.bias = d->bias; a.b_rows = d->cout;
  a.ktot = pfk_conv_ktot(d);
  a.relu = d->relu; a.scale = d->scale;
  a.M = (long long)d->B * d->H * d->W;
  switch (d->epilogue) {
    case PFK_EPI_LINEAR:
      if (!d->out || d->out_ld < d->out_coff + d->cout) return PFK_ERR_BAD_ARG;
      a.out = d->out; a.out_ld = d->out_ld; a.out_coff = d->out_coff;
      break;
    case PFK_EPI_GRU_ZR:
      if (!d->h || !d->aux_z || !d->aux_rh || (d->cout & 63)) return PFK_ERR_BAD_ARG;
      a.h = d->h; a.h_ld = d->h_ld; a.aux_z = d->aux_z; a.aux_rh = d->aux_rh;
      a.ch_hidden = d->cout / 2;
      break;
    case PFK_EPI_GRU_Q:
      if (!d->h || !d->aux_z || (d->cout & 31)) return PFK_ERR_BAD_ARG;
      a.h = d->h; a.h_ld = d->h_ld; a.aux_z = d->aux_z;
      a.ch_hidden = d->cout;
      break;
    default:
      return PFK_ERR_BAD_ARG;
  }
  return launch(a, d->epilogue, 1, static_cast<hipStream_t>(stream));
}

int pfk_corr_volume_f32(const float* f1, int ld1, const float* f2, int ld2, float* out, int B,
                        int N1, int N2, int D, float scale, pfk_stream_t stream) {
  if (!f1 || !f2 || !out || B <= 0 || N1 <= 0 || N2 <= 0 || D <= 0) return PFK_ERR_BAD_ARG;
  if (ld1 < D || ld2 < D) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(f1) || !pfk_aligned16(f2) || (ld1 & 3) || (ld2 & 3) || (D & 3))
    return PFK_ERR_ALIGNMENT;
  if (ld2 != round_up32(D)) return PFK_ERR_UNSUPPORTED;  // f2 rows are read as packed weight rows
  GemmArgs a{};
  a.src0 = f1; a.ld0 = ld1; a.ch0 = D; a.nsrc = 1;
  a.H = 1; a.W = N1; a.kh = 1; a.kw = 1;
  a.weight = f2; a.bias = nullptr; a.b_rows = N2; a.ktot = ld2;
  a.relu = 0; a.scale = scale;
  a.out = out; a.out_ld = N2; a.out_coff = 0;
  a.M = N1;
  a.a_bs = (long long)N1 * ld1; a.b_bs = (long long)N2 * ld2; a.o_bs = (long long)N1 * N2;
  return launch(a, PFK_EPI_LINEAR, B, static_cast<hipStream_t>(stream));
}

}  // extern "C"
