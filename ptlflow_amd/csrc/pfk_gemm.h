// Pieces shared by the implicit-GEMM translation units (pfk_gemm.hip: fp32 MFMA; pfk_gemm_bf.hip: split-bf16 MFMA):
// the kernel argument block, the buffer-descriptor helper and the fused epilogues.
#pragma once
#include "pfk_common.h"

namespace pfkg {

struct GemmArgs {
  const float* src0; const float* src1; const float* src2;
  int ld0, ld1, ld2;
  int ch0, ch1, ch2;
  int nsrc;
  int H, W;            // input image dims for tap bounds (sources have B*H*W rows)
  int Ho, Wo, stride;  // output dims (M = B*Ho*Wo rows, batch folded into M); output (yo, xo) reads input (yo*stride + dy, xo*stride + dx)
  int relu2;           // LINEAR: relu again after the residual add (the residual blocks of raft/extractor.py)
  int kh, kw;
  const float* weight; // [b_rows][ktot]
  const float* bias;
  int b_rows;          // valid rows of weight (= cout)
  int ktot;
  int relu;
  float scale;
  float* out; int out_ld; int out_coff;
  int vec_flags;       // host-computed (gemm_vec_flags): bit 0 out, bit 1 residual, bit 2 bias may be accessed as aligned float4
  int out_bf16;        // LINEAR: `out` points to bf16 elements (out_ld / out_coff / o_bs in elements): v is rounded to nearest even
  const float* residual; int residual_ld;   // LINEAR: out = residual[p][n] + v (after relu/scale)
  float* h; int h_ld;
  float* aux_z; float* aux_rh;
  int ch_hidden;       // Ch for the GRU epilogues
  long long M;
  long long a_bs, b_bs, o_bs;  // per-blockIdx.y strides (batched correlation), floats
  int tiles_n;
  int supertile;           // > 0: walk the tile grid in supertile x supertile blocks (see tile_of) instead of row-major
  float* sk_ws;            // stream-K: one 64x64 fp32 partial per block
  unsigned* sk_flags;      // stream-K: one ready flag per block (all zero between launches)
  int sk_steps;            // K-steps per tile (host-computed)
  long long sk_tiles;      // output tiles
  int sk_groups;           // stream-K: 1, or 8 = one contiguous tile range per XCD (blockIdx.x % 8), stream-K inside each
  // persistent pipelined kernel (conv_gemm_pp_kernel): tile decode without runtime divisions
  unsigned wo_mul, ho_mul, tn_mul;   // magic multipliers for / Wo, / Ho, / tiles_n (fastdiv_u32)
  int wo_sh, ho_sh, tn_sh;
  int pp_tiles_m;          // row tiles per batch element
  int pp_tiles_pb;         // tiles per batch element (pp_tiles_m * tiles_n)
  int pp_batches;          // batch elements folded into the tile index (correlation volume)
  int pp_whole;            // 1: blocks get whole tiles only (no partial-tile hand-off, no workspace)
  const void* wbf;         // split-bf16 path: weight planes [nsplit][cout][ktot] bf16, same k order as `weight`
  long long wbf_plane_bytes;
  int m_tile_base;         // split-bf16 path: a launch may cover a range of row tiles only (launch_bf's tail split); first row tile, in units of BM
  // fused mask head conv2 + softmax + convex upsampling (mask_upsample_kernel): flow read pixel-major, 8x output NCHW
  const float* mu_flow; int mu_flow_ld; float* mu_out;
  int active_tiles_n;      // > 0: only column tiles [0, active_tiles_n) of 64 output channels are computed (pfk_conv_desc.cout_active); the
                           // schedule is the full launch's.  stream-K kernel only — tile grids just launch fewer column tiles
  int sk_split_tiles;      // > 0 (pfk_conv_desc.cout_split / 64, tiles_n == 2 * sk_split_tiles): the stream-K tile order interleaves the
                           // column tiles of the two output groups — position 2i is tile i, position 2i + 1 is tile sk_split_tiles + i
};

// Tile id -> (tile_m, tile_n).  Row-major by default (tile_n fastest: the column tiles of one row panel run together and share
// the A panel through L2).  With a supertile edge S the grid is walked in S x S blocks of tiles: a big square-ish GEMM whose
// B operand does not fit an XCD's 4 MB L2 (the correlation volume: B = all of fmap2, 7.2 MB) otherwise streams B from HBM once
// per row of tiles; inside a supertile both panels (S*BM and S*BN rows) are L2-resident, which cuts operand traffic ~S/2-fold.
// Ragged edges: the last band / last column block are narrower; every earlier one is full, so offsets stay closed-form.
__device__ __forceinline__ void tile_of(int bid, int tiles_m, int tiles_n, int S, int& tile_m, int& tile_n) {
  if (S <= 0) { tile_n = bid % tiles_n; tile_m = bid / tiles_n; return; }
  const int per_band = S * tiles_n;
  int band = bid / per_band;
  const int full_bands = tiles_m / S;
  if (band > full_bands) band = full_bands;
  const int m_base = band * S;
  const int hb = (tiles_m - m_base) < S ? (tiles_m - m_base) : S;
  const int local = bid - band * per_band;
  int col = local / (hb * S);
  const int full_cols = tiles_n / S;
  if (col > full_cols) col = full_cols;
  const int n_base = col * S;
  const int wb = (tiles_n - n_base) < S ? (tiles_n - n_base) : S;
  const int l2 = local - col * hb * S;
  tile_m = m_base + l2 / wb;
  tile_n = n_base + l2 - (l2 / wb) * wb;
}

// Division of n < 2^31 by a launch-invariant d through a host-made multiplier (Granlund & Montgomery 1994, N = 32):
//   l = ceil(log2 d), mul = floor(2^32 (2^l - d) / d) + 1,  n / d = (mulhi(mul, n) + n) >> l   (no overflow: mulhi(..) <= n < 2^31)
// three instructions instead of the ~30 of a runtime 32-bit (or ~100 of a 64-bit) division.
static inline void fastdiv_make(unsigned d, unsigned& mul, int& sh) {
  int l = 0;
  while ((1ull << l) < d) ++l;
  mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1ull);
  sh = l;
}
__host__ __device__ __forceinline__ unsigned fastdiv_u32(unsigned n, unsigned mul, int sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (__umulhi(n, mul) + n) >> sh;
#else
  return (unsigned)((((unsigned long long)n * mul) >> 32) + n) >> sh;
#endif
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;  // >= num_records of every descriptor below

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// Epilogue for accumulator registers [R0, R1) of every 32x32 block of the wave tile.
template <int MT, int NT, int EPI, int R0, int R1>
__device__ __forceinline__ void epilogue(const GemmArgs& a, const f32x16 (&acc)[MT][NT], long long m_base,
                                         int n_base, int lane, long long batch) {
  const int col_l = lane & 31;
  const int row_l = (lane >> 5) * 4;
  float* outp = a.out + batch * a.o_bs;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n_base + nt * 32 + col_l;
    const bool nok = n < a.b_rows;
    const float bias = (a.bias != nullptr && nok) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = R0; r < R1; ++r) {
        const long long p = m_base + mt * 32 + (r & 3) + 8 * (r >> 2) + row_l;
        if (!nok || p >= a.M) continue;
        float v = acc[mt][nt][r] + bias;
        if constexpr (EPI == PFK_EPI_LINEAR) {
          if (a.relu) v = (v < 0.f) ? 0.f : v;  // NaN-propagating like torch.relu (fmaxf would drop NaN)
          v *= a.scale;
          if (a.residual != nullptr) v = a.residual[p * a.residual_ld + n] + v;
          if (a.relu2) v = (v < 0.f) ? 0.f : v;
          if (a.out_bf16) reinterpret_cast<__bf16*>(a.out)[batch * a.o_bs + p * a.out_ld + a.out_coff + n] = (__bf16)v;
          else outp[p * a.out_ld + a.out_coff + n] = v;
        } else if constexpr (EPI == PFK_EPI_GRU_ZR) {
          const int ch = a.ch_hidden;
          if (a.residual != nullptr) v += a.residual[p * a.residual_ld + n];   // loop-invariant part of the pre-activation (pfk.h)
          const float g = sigmoid_f(v);
          if (n < ch) {
            a.aux_z[p * ch + n] = g;
          } else {
            const int c = n - ch;
            a.aux_rh[p * ch + c] = g * a.h[p * a.h_ld + c];
          }
        } else {  // PFK_EPI_GRU_Q
          const int ch = a.ch_hidden;
          if (a.residual != nullptr) v += a.residual[p * a.residual_ld + n];
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


// Epilogue through LDS: the MFMA accumulator layout gives a lane one column of 16 scattered rows, i.e. sixteen 4-byte stores per
// 32x32 block, each wave instruction covering two 128-byte row segments.  Here a wave parks one 32x32 block at a time in a private
// 4 KB LDS region ([32][32] floats; the ds_write_b32 of a half-wave covers 32 consecutive floats, the ds_read_b128 lane groups hit
// 16 distinct 16-byte slots: conflict-free unpadded) and re-reads it row-wise: lane = (row l >> 3 of 8, four columns (l & 7) * 4),
// so the fused arithmetic runs on float4 and every global access (bias, residual, h, z, out) is a 16-byte access of a 128-byte
// row segment — four store instructions per block instead of sixteen, 8 rows x 128 B per instruction.  The accumulation order of
// the arithmetic per element is unchanged (bit-identical results).  `lds` = block-shared scratch >= 16 KB that no wave reads
// any more (callers sit behind their K loop's final barrier).
template <int MT, int NT, int EPI, bool PRELOAD = false>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& a, const f32x16 (&acc)[MT][NT], long long m_base, int n_base, int lane,
                                             long long batch, float* lds, int wid) {
  float* reg = lds + wid * 1024;                       // this wave's [32][32] block
  const int col_l = lane & 31, row_l = (lane >> 5) * 4;
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;     // read side: 8 rows per pass, 4 columns per lane
  // explicit global address space: with the plain generic pointers hipcc emitted flat_store for the LINEAR branch
  typedef __attribute__((address_space(1))) float gfloat;
  typedef __attribute__((address_space(1))) f32x4 gf32x4;
  typedef __attribute__((address_space(1))) __bf16 gbf16;
  gfloat* outp = (gfloat*)(a.out + batch * a.o_bs);
  // 16-byte accesses need the row stride / offsets to be multiples of 4 floats and the bases 16-byte aligned (pfk.h asks for it;
  // the training path's fresh [M, cout] outputs with cout = 126 or 2 do not comply): wave-uniform fall-back to scalar accesses
  // (decided on the host: a pointer-to-integer cast here makes the compiler lose the global address space of every store below)
  const bool vout = a.vec_flags & 1, vres = a.vec_flags & 2, vbias = a.vec_flags & 4;
  // GRU epilogues on wave tiles of at most two 32x32 blocks (the fp32 kernels): the epilogue's global operands — the loop-invariant
  // pre-activation term (`residual`), h for r*h, z and h for the blend — are loaded for the WHOLE wave tile up front, so their
  // latency runs under the LDS transposes instead of being paid once per block behind each `s_waitcnt lgkmcnt(0)` (round 4: the gate
  // epilogues cost 2.7 % / 6.7 % of a z|r / q launch, gpurun_out/r4m_conv_b8.log).  Same values, same arithmetic: same bits.
  // (PRELOAD: only the tile-grid kernel asks for it — the stream-K / persistent kernels are held to 168 registers for three blocks
  //  per CU and spilled with the 32-48 extra ones)
  constexpr bool PRE = PRELOAD && EPI != PFK_EPI_LINEAR && MT * NT <= 2;
  f32x4 pre_res[PRE ? NT : 1][PRE ? MT : 1][4], pre_a[PRE ? NT : 1][PRE ? MT : 1][4], pre_b[PRE ? NT : 1][PRE ? MT : 1][4];
  if constexpr (PRE) {
    const int ch = a.ch_hidden;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n_base + nt * 32 + c4;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const long long p = m_base + mt * 32 + pass * 8 + rrow;
          const bool ok = p < a.M && n < a.b_rows;
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          pre_res[nt][mt][pass] = (ok && a.residual != nullptr) ? *reinterpret_cast<const f32x4*>(a.residual + p * a.residual_ld + n) : zero;
          if constexpr (EPI == PFK_EPI_GRU_ZR) {
            pre_a[nt][mt][pass] = (ok && n >= ch) ? *reinterpret_cast<const f32x4*>(a.h + p * a.h_ld + (n - ch)) : zero;
          } else {
            pre_a[nt][mt][pass] = ok ? *reinterpret_cast<const f32x4*>(a.aux_z + p * ch + n) : zero;
            pre_b[nt][mt][pass] = ok ? *reinterpret_cast<const f32x4*>(a.h + p * a.h_ld + n) : zero;
          }
        }
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n_base + nt * 32 + c4;
    const bool full = n + 3 < a.b_rows;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) {
      if (full && vbias) bias = *reinterpret_cast<const f32x4*>(a.bias + n);
      else {
        if (n + 0 < a.b_rows) bias[0] = a.bias[n + 0];
        if (n + 1 < a.b_rows) bias[1] = a.bias[n + 1];
        if (n + 2 < a.b_rows) bias[2] = a.bias[n + 2];
        if (n + 3 < a.b_rows) bias[3] = a.bias[n + 3];
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) reg[((r & 3) + 8 * (r >> 2) + row_l) * 32 + col_l] = acc[mt][nt][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: in-order LDS, no workgroup barrier needed
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + rrow;
        f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * 32 + c4);
        const long long p = m_base + mt * 32 + row;
        // (columns past cout are filtered per element below, not here: knowing `n < cout` the compiler hoists the scalar tail's first
        // store above the vector / scalar branch and every float4 store is preceded by a redundant dword store: +25 % write traffic)
        if (p >= a.M) continue;
        v += bias;
        if constexpr (EPI == PFK_EPI_LINEAR) {
          if (a.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];   // NaN-propagating like torch.relu
          }
          {
            // (element-wise on purpose: `v *= a.scale` made hipcc park a 16-byte slice of the kernel arguments in SCRATCH at
            // kernel entry and re-load it here — 16 B per thread of extra HBM writes, +25 % on WRITE_SIZE for every LINEAR launch)
            const float sc = a.scale;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= sc;
          }
          if (a.residual != nullptr) {
            if (full && vres) v = *reinterpret_cast<const f32x4*>(a.residual + p * a.residual_ld + n) + v;
            else {
              const float* rp = a.residual + p * a.residual_ld + n;
              if (n + 0 < a.b_rows) v[0] = rp[0] + v[0];
              if (n + 1 < a.b_rows) v[1] = rp[1] + v[1];
              if (n + 2 < a.b_rows) v[2] = rp[2] + v[2];
              if (n + 3 < a.b_rows) v[3] = rp[3] + v[3];
            }
          }
          if (a.relu2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
          }
          if (a.out_bf16) {
            gbf16* ob = (gbf16*)(reinterpret_cast<__bf16*>(a.out) + batch * a.o_bs + p * a.out_ld + a.out_coff + n);
            if (n + 0 < a.b_rows) ob[0] = (__bf16)v[0];
            if (n + 1 < a.b_rows) ob[1] = (__bf16)v[1];
            if (n + 2 < a.b_rows) ob[2] = (__bf16)v[2];
            if (n + 3 < a.b_rows) ob[3] = (__bf16)v[3];
          } else if (full && vout) {
            *(gf32x4*)(outp + p * a.out_ld + a.out_coff + n) = v;
          } else {
            gfloat* op = outp + p * a.out_ld + a.out_coff + n;
            if (n + 0 < a.b_rows) op[0] = v[0];
            if (n + 1 < a.b_rows) op[1] = v[1];
            if (n + 2 < a.b_rows) op[2] = v[2];
            if (n + 3 < a.b_rows) op[3] = v[3];
          }
        } else if constexpr (EPI == PFK_EPI_GRU_ZR) {     // cout = 2 * ch, ch % 4 == 0: a float4 never straddles z | r
          if (n >= a.b_rows) continue;
          const int ch = a.ch_hidden;
          // the loop-invariant part of the gate pre-activations (context features x their weight slice + bias, computed once per
          // forward — pfk.h, `residual` on the GRU epilogues): [M][cout] rows, 16-byte aligned (desc_to_args checks)
          if constexpr (PRE) { if (a.residual != nullptr) v += pre_res[nt][mt][pass]; }
          else if (a.residual != nullptr) v += *reinterpret_cast<const f32x4*>(a.residual + p * a.residual_ld + n);
          f32x4 g;
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = sigmoid_f(v[e]);
          if (n < ch) {
            *reinterpret_cast<f32x4*>(a.aux_z + p * ch + n) = g;
          } else {
            const int c = n - ch;
            f32x4 hv;
            if constexpr (PRE) hv = pre_a[nt][mt][pass];
            else hv = *reinterpret_cast<const f32x4*>(a.h + p * a.h_ld + c);
            *reinterpret_cast<f32x4*>(a.aux_rh + p * ch + c) = g * hv;
          }
        } else {  // PFK_EPI_GRU_Q
          if (n >= a.b_rows) continue;
          const int ch = a.ch_hidden;
          if constexpr (PRE) { if (a.residual != nullptr) v += pre_res[nt][mt][pass]; }
          else if (a.residual != nullptr) v += *reinterpret_cast<const f32x4*>(a.residual + p * a.residual_ld + n);
          f32x4 z, hv;
          if constexpr (PRE) { z = pre_a[nt][mt][pass]; hv = pre_b[nt][mt][pass]; }
          else {
            z = *reinterpret_cast<const f32x4*>(a.aux_z + p * ch + n);
            hv = *reinterpret_cast<const f32x4*>(a.h + p * a.h_ld + n);
          }
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q = tanhf(v[e]);
            // (1 - z) * h + z * q, each product rounded (no contraction), as update.py:64,71
            o[e] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, z[e]), hv[e]), __fmul_rn(z[e], q));
          }
          *reinterpret_cast<f32x4*>(a.h + p * a.h_ld + n) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next block overwrites the region
    }
  }
}

// alignment facts the float4 epilogue needs, evaluated where pointers are still integers
static inline int gemm_vec_flags(const GemmArgs& a) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  int f = 0;
  if (a.out && al(a.out) && (((a.out_ld | a.out_coff) & 3) == 0) && ((a.o_bs & 3) == 0)) f |= 1;
  if (!a.residual || (al(a.residual) && (a.residual_ld & 3) == 0)) f |= 2;
  if (!a.bias || al(a.bias)) f |= 4;
  return f;
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
}

// pfk_gemm_bf.hip: split-bf16 implicit GEMM (nsplit 1..3)
extern int g_bf_cfg;
int launch_bf(const GemmArgs& a, int epi, int nsplit, hipStream_t st);

// One (pixel, sub-pixel) of the fused epilogue: the nine logits -> softmax -> convex combination of the neighbours' flows, with the
// arithmetic of the unfused pair operation for operation — conv epilogue (pfk_gemm.h, LINEAR): + bias, x scale; convex_upsample_kernel
// (pfk_misc.hip, compiled with -ffp-contract=off): max, exp, sum in tap order, one reciprocal, multiply then add.  In the unfused
// pair the mask goes through memory between the two kernels and pfk_misc.hip never contracts, so NOTHING here may be contracted
// into an FMA either: this translation unit is compiled with contraction on and HIP's __fmul_rn / __fadd_rn are plain operators
// (they do not stop it), hence the pragma.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mask_upsample_combine(const float (&m_in)[9], const f32x2* nfp, float& ox, float& oy) {
#pragma clang fp contract(off)
  float m[9];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) { m[k] = m_in[k]; mx = fmaxf(mx, m[k]); }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); sum = sum + m[k]; }
  const float inv = 1.0f / sum;
  float ax = 0.f, ay = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const f32x2 f = nfp[k];
    const float wk = m[k] * inv;
    const float tx = wk * f[0], ty = wk * f[1];
    ax = ax + tx;
    ay = ay + ty;
  }
  ox = ax;
  oy = ay;
}
// the conv epilogue's + bias, x scale on one logit, un-contracted
__device__ __forceinline__ float mask_logit(float acc, float bias, float sc) {
#pragma clang fp contract(off)
  float v = acc + bias;
  v = v * sc;
  return v;
}


}  // namespace pfkg
