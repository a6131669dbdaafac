// K4-K6 on the gfx950 bf16 matrix cores with *split* operands (v_mfma_f32_32x32x16_bf16: 8 passes for a K of 16,
// i.e. 16x the fp32 MFMA's multiply rate, fp32 accumulate).
//
// An fp32 operand x is written as a sum of bf16 planes, x = x0 + x1 (+ x2) + e, with x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (each subtraction is exact in fp32).  The product of two split
// operands keeps the terms a_i * b_j with i + j < nsplit:
//   nsplit 1:  a0 b0                                   plain bf16 operands          (rel. product error ~2^-9)
//   nsplit 2:  a0 b0 + a0 b1 + a1 b0                   3 MFMAs per K-block          (~2^-17)
//   nsplit 3:  ... + a0 b2 + a1 b1 + a2 b0             6 MFMAs per K-block          (~2^-24: fp32-grade)
// Terms of equal order i + j share an accumulator; the accumulators are added smallest first at the end.
// The weights arrive pre-split (planes [nsplit][cout][ktot] bf16, host packing); the activations are split
// while they are staged: global fp32 -> VGPR -> cvt/sub/cvt -> LDS bf16 planes, so HBM traffic is the fp32
// path's and LDS holds K = 64 channels per 128-byte row.
//
// Same implicit-GEMM structure as pfk_gemm.hip: K-step = (source, tap, 64 channels), raw buffer loads with
// hardware zero fill for the conv padding, un-padded LDS rows with the 16-byte chunk index XOR-swizzled by
// (row >> 1) & 7 (conflict-free ds_read_b128 / ds_write_b128), two LDS stages, fused epilogues.
// Fragment use (32x32x16): lane l holds A[i = l & 31][k = 8*(l >> 5) .. +7] — one ds_read_b128 of chunk
// kb*2 + (l >> 5) per plane per 16-channel K-block kb.
#include "pfk_gemm.h"

using namespace pfkg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BKB = 64;            // channels per K-step
constexpr int ROWB = 128;          // bytes per LDS row: 64 bf16
constexpr int PLANE = 64 * ROWB;   // one 64-row operand plane, 8 KB

template <int NS>
__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, u32x4 (&out)[NS]) {
  float r[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
  for (int pl = 0; pl < NS; ++pl) {
    bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h[e] = (__bf16)r[e];                       // round to nearest even (v_cvt_pk_bf16_f32)
      if (pl + 1 < NS) r[e] = r[e] - (float)h[e];  // exact
    }
    out[pl] = __builtin_bit_cast(u32x4, h);
  }
}

// 256 threads stage one K-step: thread t owns channels (t & 7)*8 .. +7 of rows (t >> 3) and (t >> 3) + 32 of
// both operands.
template <int NS>
struct StagerBF {
  int H, W, kh, kw, ph, pw, nsrc;
  int ld0, ld1, ld2, ch0, ch1, ch2;
  __amdgpu_buffer_rsrc_t rs0, rs1, rs2, rsw, rs;
  int cld, cch;
  int plane_bytes;
  int seg = 0, ky = 0, kx = 0, c0 = 0, kofs = 0;
  int c8, r0;
  unsigned sbyte;             // byte offset of this thread's 16-byte chunk inside a plane (row r0; row r0+32 = +32*ROWB)
  int prow[2], py[2], px[2];
  bool pok[2];
  unsigned abase[2], aoff[2], wvoff[2];
  f32x4 ra[2][2];
  u32x4 rb[NS][2];

  __device__ __forceinline__ StagerBF(const GemmArgs& a, long long m0, int n0, int t) {
    H = a.H; W = a.W; kh = a.kh; kw = a.kw; ph = a.kh >> 1; pw = a.kw >> 1; nsrc = a.nsrc;
    ld0 = a.ld0; ld1 = a.ld1; ld2 = a.ld2; ch0 = a.ch0; ch1 = a.ch1; ch2 = a.ch2;
    rs0 = make_rsrc(a.src0);
    rs1 = make_rsrc(a.src1);
    rs2 = make_rsrc(a.src2);
    rsw = make_rsrc(a.wbf);
    plane_bytes = (int)a.wbf_plane_bytes;
    c8 = (t & 7) * 8;
    r0 = t >> 3;
    sbyte = (unsigned)(r0 * ROWB + (((t & 7) ^ ((r0 >> 1) & 7)) << 4));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long p = m0 + r0 + 32 * i;
      pok[i] = p < a.M;
      prow[i] = (int)p;
      px[i] = (int)(p % a.W);
      py[i] = (int)((p / a.W) % a.H);
      const int n = n0 + r0 + 32 * i;
      wvoff[i] = n < a.b_rows ? (unsigned)(n * a.ktot + c8) * 2u : OOB;
    }
    set_segment(0);
    set_tap();
  }

  __device__ __forceinline__ void set_segment(int s) {
    if (s == 0) { rs = rs0; cld = ld0; cch = ch0; }
    else if (s == 1) { rs = rs1; cld = ld1; cch = ch1; }
    else { rs = rs2; cld = ld2; cch = ch2; }
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = (unsigned)(prow[i] * cld + c8) * 4u;
  }

  __device__ __forceinline__ void set_tap() {
    const int dy = ky - ph, dx = kx - pw;
    const unsigned toff = (unsigned)((dy * W + dx) * cld * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = pok[i] && (unsigned)(py[i] + dy) < (unsigned)H && (unsigned)(px[i] + dx) < (unsigned)W;
      aoff[i] = ok ? abase[i] + toff : OOB;
    }
  }

  __device__ __forceinline__ int total_steps() const {
    const int taps = kh * kw;
    int s = taps * ((ch0 + BKB - 1) / BKB);
    if (nsrc > 1) s += taps * ((ch1 + BKB - 1) / BKB);
    if (nsrc > 2) s += taps * ((ch2 + BKB - 1) / BKB);
    return s;
  }

  // live = false: every A lane out of range (zeros), B parked on K-step 0 (valid memory) — branch-free last step
  __device__ __forceinline__ void load(bool live) {
    const int lim = live ? cch - c0 : 0;
    const bool ok0 = c8 < lim, ok1 = c8 + 4 < lim;   // sources have a multiple of 4 channels
    const int coff = c0 * 4;
    const int koff = live ? kofs * 2 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned o = aoff[i];
      ra[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok0 ? o : OOB, coff, 0));
      ra[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (ok1 && o != OOB) ? o + 16u : OOB, coff, 0));
    }
#pragma unroll
    for (int pl = 0; pl < NS; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        rb[pl][i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], koff + pl * plane_bytes, 0);
  }

  __device__ __forceinline__ void advance() {
    kofs += BKB;
    c0 += BKB;
    if (c0 >= cch) {
      c0 = 0;
      if (++kx == kw) {
        kx = 0;
        if (++ky == kh) {
          ky = 0;
          ++seg;
          if (seg < nsrc) set_segment(seg);
        }
      }
      set_tap();
    }
  }

  // stage layout: [A planes 0..NS-1][B planes 0..NS-1], each [64 rows][128 B]
  __device__ __forceinline__ void store(char* stage) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 pa[NS];
      split8<NS>(ra[i][0], ra[i][1], pa);
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) {
        *reinterpret_cast<u32x4*>(stage + pl * PLANE + i * 32 * ROWB + sbyte) = pa[pl];
        *reinterpret_cast<u32x4*>(stage + (NS + pl) * PLANE + i * 32 * ROWB + sbyte) = rb[pl][i];
      }
    }
  }
};

template <int EPI, int NS>
__global__ __launch_bounds__(256) void conv_gemm_bf_kernel(const GemmArgs a) {
  constexpr int STAGE = 2 * NS * PLANE;
  extern __shared__ __attribute__((aligned(16))) char smem_bf[];   // [2][STAGE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm0 = (wid >> 1) * 32;
  const int wn0 = (wid & 1) * 32;

  const int bid = pfk_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n;
  const int tile_m = bid / a.tiles_n;
  const long long m0 = (long long)tile_m * 64;
  const int n0 = tile_n * 64;

  StagerBF<NS> st(a, m0, n0, tid);
  const int total_steps = st.total_steps();

  f32x16 acc[NS];   // acc[o]: products of order i + j = o
#pragma unroll
  for (int o = 0; o < NS; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;

  st.load(true);
  st.advance();
  st.store(smem_bf);
  __syncthreads();

  const int frow = lane & 31;
  const int hl = lane >> 5;
  const int key = (frow >> 1) & 7;   // wave / tile row offsets are multiples of 32: same key
  int ko[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) ko[kb] = ((kb * 2 + hl) ^ key) << 4;

  for (int step = 0; step < total_steps; ++step) {
    const int buf = step & 1;
    const bool more = (step + 1) < total_steps;
    const char* cA = smem_bf + buf * STAGE + (wm0 + frow) * ROWB;
    const char* cB = smem_bf + buf * STAGE + NS * PLANE + (wn0 + frow) * ROWB;
    st.load(more);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      bf16x8 fa[NS], fb[NS];
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) {
        fa[pl] = *reinterpret_cast<const bf16x8*>(cA + pl * PLANE + ko[kb]);
        fb[pl] = *reinterpret_cast<const bf16x8*>(cB + pl * PLANE + ko[kb]);
      }
#pragma unroll
      for (int o = 0; o < NS; ++o)
#pragma unroll
        for (int i = 0; i <= o; ++i)
          acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[o - i], acc[o], 0, 0, 0);
    }
    st.store(smem_bf + (buf ^ 1) * STAGE);   // last step: zeros / parked weights into the dead stage
    if (more) st.advance();
    __syncthreads();
  }

  f32x16 sum[1][1];
  sum[0][0] = acc[NS - 1];
#pragma unroll
  for (int o = NS - 2; o >= 0; --o) sum[0][0] += acc[o];
  epilogue<1, 1, EPI, 0, 16>(a, sum, m0 + wm0, n0 + wn0, lane, 0);
}

template <int EPI, int NS>
int launch_bf_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
  constexpr size_t smem = 2 * 2 * NS * PLANE;
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = conv_gemm_bf_kernel<EPI, NS>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, g);
  return pfk_launch_status();
}

template <int NS>
int launch_bf_ns(const GemmArgs& g, int epi, dim3 grid, hipStream_t st) {
  switch (epi) {
    case PFK_EPI_LINEAR: return launch_bf_one<PFK_EPI_LINEAR, NS>(g, grid, st);
    case PFK_EPI_GRU_ZR: return launch_bf_one<PFK_EPI_GRU_ZR, NS>(g, grid, st);
    case PFK_EPI_GRU_Q:  return launch_bf_one<PFK_EPI_GRU_Q, NS>(g, grid, st);
    default: return PFK_ERR_BAD_ARG;
  }
}

}  // namespace

namespace pfkg {

int launch_bf(const GemmArgs& a, int epi, int nsplit, hipStream_t st) {
  GemmArgs g = a;
  const long long tiles_m = (a.M + 63) / 64;
  g.tiles_n = (a.b_rows + 63) / 64;
  const long long nblk = tiles_m * g.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nblk);
  switch (nsplit) {
    case 1: return launch_bf_ns<1>(g, epi, grid, st);
    case 2: return launch_bf_ns<2>(g, epi, grid, st);
    case 3: return launch_bf_ns<3>(g, epi, grid, st);
    default: return PFK_ERR_BAD_ARG;
  }
}

}  // namespace pfkg
