// K8b — the convolutions of the update block (ptlflow/models/raft/update.py:6-153) as implicit GEMMs on the gfx950 bf16 matrix cores
// with bf16 ACTIVATION STORAGE: what the reference computes under its reduced-precision switch (model_benchmark.py:317-319,
// validate.py:243-244 / torch.autocast: every nn.Conv2d reads and writes 16-bit tensors).  pfk_gemm_bf.hip keeps fp32 activations in
// HBM and splits them into bf16 planes while staging (fp32 traffic + a cvt / ds_write pass per tap); here the producers' epilogues
// emit bf16 once and BOTH operands of a K-step go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no VGPR round trip, no
// conversion, no ds_write.  The hidden state, the gate z and every accumulator stay fp32.
//
// Geometry.  K-step = (source, tap, 64 channels): one 128-byte row piece per pixel / weight row.  A wave's LDS-DMA instruction lands
// 64 lanes x 16 B = 1 KiB of consecutive LDS = 8 rows x 128 B, so the image is row-major [rows][128 B] and the bank swizzle is put on
// the SOURCE side: lane (row r, physical 16-byte chunk c) fetches logical chunk c ^ ((r >> 1) & 7).  Fragment reads
// (v_mfma_f32_32x32x16_bf16: lane l holds row l & 31, k = 8 (l >> 5) .. +7 of a 16-channel block kb) are one ds_read_b128 of physical
// chunk (2 kb + (l >> 5)) ^ ((l >> 1) & 7) — sixteen distinct 16-byte slots of the 256-byte bank row for each of the instruction's
// lane groups (MI355X_MICROARCH.md, LDS table): conflict-free.
// Zero padding of the convolution, rows past M / cout and channels past a source's width are lanes whose buffer offset is out of
// range: the hardware writes zeros into LDS for them.
// Pipeline.  NST LDS stages (3: 144 KB for the 256x128 tile), the DMA of step j + NST - 1 is issued at the top of step j; a counted
// `s_waitcnt vmcnt(pieces per step)` + a raw `s_barrier` at the end of step j retire step j + 1 while step j + 2 stays in flight
// (cdna_hip_programming.md §5: LDS-DMA survives a raw barrier, `__syncthreads()` would drain it).
#include "pfk_gemm.h"

#include <utility>

using namespace pfkg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK16 = 64;          // channels per K-step
constexpr int ROW16 = 128;        // bytes per LDS row

struct B16Args {
  const void* src[3];
  int ld[3], ch[3];               // bf16 elements
  int nsrc;
  int H, W, Ho, Wo, stride, kh, kw;
  const void* weight;             // bf16 [b_rows][ktot]
  int ktot, b_rows;
  const float* bias;
  int relu, relu2;
  float scale;
  void* out; int out_ld, out_coff, out_bf16;
  int out_vec16;                  // LINEAR, bf16 rows, no residual, every row piece 16-byte aligned: the 8-columns-per-lane epilogue (16-byte stores)
  const void* residual; int residual_ld;   // LINEAR: fp32 rows (bf16 with residual_bf16); GRU epilogues: bf16 rows
  int residual_bf16;
  // batched GEMM (LINEAR, one source, 1x1): blockIdx.y selects the problem; byte strides of the source / weight / out / residual
  int batches;
  long long src_bs, w_bs, out_bs, res_bs;
  float* h; int h_ld;
  void* h_b16; int hb_ld;         // GRU_Q: bf16 copy of the new hidden state (the next convolutions' A operand), may be null
  void* aux_z; void* aux_rh;      // z, r*h: bf16 [M][Ch]
  int ch_hidden;
  long long M;
  int tiles_n, steps;
  unsigned wo_mul, ho_mul;
  int wo_sh, ho_sh;
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)       // (the host pass of hipcc does not know the LDS-DMA builtin)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
#endif
}

// Global -> LDS stager of one thread: rows (t >> 3) + RPP * i of both operands, physical chunk t & 7.
template <int BM, int BN, int TH>
struct Stager16 {
  static constexpr int RPP = TH / 8;                     // rows per pass (8 per wave instruction)
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows staged per pass");
  static constexpr int A_PT = BM / RPP, B_PT = BN / RPP;
  static constexpr int PIECES = A_PT + B_PT;             // LDS-DMA instructions per thread and K-step
  static constexpr int A_BYTES = BM * ROW16, STAGE = (BM + BN) * ROW16;
  __amdgpu_buffer_rsrc_t rs0, rs1, rs2, rsw, rs;
  int ld0, ld1, ld2, ch0, ch1, ch2, nsrc, W, kh, kw, ph, pw;
  int cld, cch;
  int seg = 0, ky = 0, kx = 0, tap = 0, c0 = 0, kofs = 0, pos = 0, total;
  int lc8;                       // first channel of this lane's logical chunk inside a K-step
  unsigned wave_off;             // byte offset of this wave's 8 rows inside a pass (wave-uniform)
  int prow[A_PT];
  unsigned vmask[A_PT];          // bit t: tap t of this row lies inside the image (and the row inside M)
  unsigned abase[A_PT], aoff[A_PT], wvoff[B_PT];

  __device__ __forceinline__ Stager16(const B16Args& a, long long m0, int n0, int t) {
    W = a.W; kh = a.kh; kw = a.kw; ph = a.kh >> 1; pw = a.kw >> 1; nsrc = a.nsrc;
    ld0 = a.ld[0]; ld1 = a.ld[1]; ld2 = a.ld[2]; ch0 = a.ch[0]; ch1 = a.ch[1]; ch2 = a.ch[2];
    rs0 = make_rsrc(a.src[0]); rs1 = make_rsrc(a.src[1]); rs2 = make_rsrc(a.src[2]); rsw = make_rsrc(a.weight);
    total = a.steps;
    const int r0 = t >> 3;
    // rows of one thread differ by multiples of RPP (a multiple of 16): the swizzle key is the same for all of them
    lc8 = ((t & 7) ^ ((r0 >> 1) & 7)) * 8;
    wave_off = (unsigned)__builtin_amdgcn_readfirstlane(t >> 6) * 1024u;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const unsigned p = (unsigned)m0 + (unsigned)(r0 + RPP * i);
      const bool pok = (long long)p < a.M;
      const unsigned prow_o = fastdiv_u32(p, a.wo_mul, a.wo_sh);            // b*Ho + yo
      const unsigned bimg = fastdiv_u32(prow_o, a.ho_mul, a.ho_sh);
      const int px = (int)(p - prow_o * (unsigned)a.Wo) * a.stride;         // input coordinates of the centre tap
      const int py = (int)(prow_o - bimg * (unsigned)a.Ho) * a.stride;
      prow[i] = (int)((bimg * (unsigned)a.H + (unsigned)py) * (unsigned)a.W + (unsigned)px);
      // taps inside the image: tx in [pw - px, W + pw - px), ty likewise — two bit ranges, combined row by row
      const int x_lo = max(pw - px, 0), x_hi = min(a.W + pw - px, kw), y_lo = max(ph - py, 0), y_hi = min(a.H + ph - py, kh);
      const unsigned xm = (pok && x_hi > x_lo) ? ((1u << x_hi) - 1u) & ~((1u << x_lo) - 1u) : 0u;
      unsigned m = 0;
      for (int ty = y_lo; ty < y_hi; ++ty) m |= xm << (ty * kw);
      vmask[i] = m;
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int n = n0 + r0 + RPP * i;
      wvoff[i] = n < a.b_rows ? (unsigned)(n * a.ktot + lc8) * 2u : OOB;
    }
    set_segment(0);
    set_tap();
  }

  __device__ __forceinline__ void set_segment(int s) {
    if (s == 0) { rs = rs0; cld = ld0; cch = ch0; }
    else if (s == 1) { rs = rs1; cld = ld1; cch = ch1; }
    else { rs = rs2; cld = ld2; cch = ch2; }
#pragma unroll
    for (int i = 0; i < A_PT; ++i) abase[i] = (unsigned)(prow[i] * cld + lc8) * 2u;
  }

  __device__ __forceinline__ void set_tap() {
    const unsigned toff = (unsigned)(((ky - ph) * W + (kx - pw)) * cld * 2);
#pragma unroll
    for (int i = 0; i < A_PT; ++i) aoff[i] = ((vmask[i] >> tap) & 1u) ? abase[i] + toff : OOB;
  }

  __device__ __forceinline__ void advance() {
    kofs += BK16;
    c0 += BK16;
    if (c0 >= cch) {
      c0 = 0;
      ++tap;
      if (++kx == kw) {
        kx = 0;
        if (++ky == kh) {
          ky = 0; tap = 0;
          ++seg;
          if (seg < nsrc) set_segment(seg);
        }
      }
      set_tap();
    }
  }

  // One K-step's pieces into `stage`.  Past the end of K every A lane is out of range (zeros) and B is parked on K-step 0 (valid
  // memory, multiplied by zeros): the K loop stays branch-free and the counted vmcnt uniform.
  int abl = 0;      // PFK_BENCH_VARIANTS ablations 6 / 7: the A pieces of every tap but the first fetch nothing (6: zero-fill, 7: not issued)
  template <int I>
  __device__ __forceinline__ void piece(char* stage, bool live, bool cok, int coff, int koff) const {
    if constexpr (I < A_PT) {
#ifdef PFK_BENCH_VARIANTS
      if (abl == 7 && tap != 0) return;
      if (abl == 6 && tap != 0) { dma16(rs, stage + I * RPP * ROW16 + wave_off, OOB, coff); return; }
#endif
      dma16(rs, stage + I * RPP * ROW16 + wave_off, (live && cok) ? aoff[I] : OOB, coff);
    } else {
      dma16(rsw, stage + A_BYTES + (I - A_PT) * RPP * ROW16 + wave_off, wvoff[I - A_PT], koff);
    }
  }
  // the step's pieces as separate units (the kernel places them between MFMAs): begin_issue, piece_at<I> x PIECES, end_issue
  bool i_live, i_cok;
  int i_coff, i_koff;
  __device__ __forceinline__ void begin_issue() {
    i_live = pos < total;
    i_cok = c0 + lc8 < cch;            // sources have a multiple of 8 channels
    i_coff = c0 * 2; i_koff = i_live ? kofs * 2 : 0;
  }
  template <int I>
  __device__ __forceinline__ void piece_at(char* stage) const { piece<I>(stage, i_live, i_cok, i_coff, i_koff); }
  __device__ __forceinline__ void end_issue() {
    ++pos;
    if (pos < total) advance();
  }
  template <int... I>
  __device__ __forceinline__ void issue_seq(char* stage, std::integer_sequence<int, I...>) {
    begin_issue();
    (piece_at<I>(stage), ...);
    end_issue();
  }
  __device__ __forceinline__ void issue(char* stage) { issue_seq(stage, std::make_integer_sequence<int, PIECES>{}); }
};

// Gate non-linearities on the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp each): sigmoid in 4 instructions, tanh in 5,
// instead of the ~25 / ~40 of the IEEE-exact library forms — on these launches the epilogue's VALU work is not hidden behind matrix
// work (14.4 M sigmoids per z|r launch at batch 8 = ~9 us of a 60 us launch), and the results are rounded to bf16 / blended into a
// state whose inputs carry 2^-9 relative error anyway.  Limits: x -> -inf gives 0 / -1, +inf gives 1 / 1, NaN propagates.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008f)); }

// fp32 -> bf16 (round to nearest even) of four values as one 8-byte word pair
__device__ __forceinline__ u32x2 pack_bf16x4(const f32x4 v) {
  bf16x4 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
  return __builtin_bit_cast(u32x2, h);
}

// Epilogue through a wave-private LDS region of 32 rows x (NT * 32) fp32 columns: the MFMA accumulator layout gives a lane one column
// of 16 scattered rows; the wave parks one 32-row band of its tile (bias already added: a lane owns ONE column per 32x32 block, so the
// bias is a register), then re-reads it row-wise — lane = row l / LPR of a pass, four columns (l % LPR) * 4 — so that every global
// access of the fused arithmetic is an 8/16-byte piece of a contiguous row segment (bf16: 128 bytes per row of a 64-column wave tile).
// At batch 8 these launches are bound by the HBM traffic of their epilogues, not by the matrix pipe (q: 130 MB of fp32 z / h / context
// rows against 18 GFLOP), so (a) the side operands are 16-bit where the consumer's arithmetic allows it — the loop-invariant context
// term (the GRU epilogues' `residual`), z, and h for r * h come in bf16 — and (b) everything the arithmetic READS from global memory is requested
// early: band 0's operands BEFORE the K loop (they land under the MFMAs), band b + 1's before band b's first store (on gfx9 stores sit
// in the same in-order vmcnt queue as loads: a load issued behind a store waits for that store's completion).
template <int MT, int NT, int EPI>
struct Epilogue16 {
  static constexpr bool LIN = EPI == PFK_EPI_LINEAR, ZR = EPI == PFK_EPI_GRU_ZR, Q = EPI == PFK_EPI_GRU_Q;
  // columns per lane of the row-wise pass: 4 (LINEAR: fp32 rows / residuals; its bf16 rows without a residual take the 8-column branch of
  // band()), 8 for the GRU epilogues — every bf16 side operand (context term, h twin, z) and every bf16 result (z, r * h, the h twin) is ONE
  // 16-byte access per lane and pass, the fp32 h two; half the passes and half the store instructions of the 4-column form
  static constexpr int COLS = LIN ? 4 : 8;
  static constexpr int CW = NT * 32, LPR = CW / COLS, RPS = 64 / LPR, PASSES = 32 / RPS;
  typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u32x4e gu32x4e;
  typedef __attribute__((address_space(1))) float gfloat;
  typedef __attribute__((address_space(1))) f32x4 gf32x4;
  typedef __attribute__((address_space(1))) u32x2 gu32x2;
  typedef __attribute__((address_space(1))) __bf16 gbf16;
  long long m_base;
  int n, rrow, c4;
  bool full, any, has_res;
  // band operands.  LINEAR: fp32 residual rows.  GRU: bf16 pre-activation term (res16), bf16 h for r * h (ZR) / bf16 z (Q) in
  // aux16, fp32 h (Q) in h32.  Two register sets where they fit (band b + 1 is requested while band b is being processed); the q
  // epilogue (8 registers per pass) has one: its later bands are requested behind the previous band's stores and wait for them.
  static constexpr int NSETS = Q ? 1 : 2;
  f32x4 res32[LIN ? 2 : 1][LIN ? PASSES : 1];
  u32x4e res16[LIN ? 1 : NSETS][LIN ? 1 : PASSES], aux16[LIN ? 1 : NSETS][LIN ? 1 : PASSES];      // eight bf16 each
  f32x4 h32[1][Q ? PASSES : 1][2];

  __device__ __forceinline__ Epilogue16(const B16Args& a, long long m_base_, int n_base, int lane) : m_base(m_base_) {
    rrow = lane / LPR; c4 = (lane % LPR) * COLS;
    n = n_base + c4;
    full = n + COLS - 1 < a.b_rows; any = n < a.b_rows;
    has_res = a.residual != nullptr;
  }

  static __device__ __forceinline__ f32x4 widen(const u32x2 w) {      // four bf16 -> fp32 (exact)
    f32x4 v;
    v[0] = __builtin_bit_cast(float, w[0] << 16); v[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, w[1] << 16); v[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
    return v;
  }

  static __device__ __forceinline__ f32x4 widen_lo(const u32x4e w) { const u32x2 h = {w[0], w[1]}; return widen(h); }
  static __device__ __forceinline__ f32x4 widen_hi(const u32x4e w) { const u32x2 h = {w[2], w[3]}; return widen(h); }
  static __device__ __forceinline__ u32x4e pack_bf16x8(const f32x4 a, const f32x4 b) {
    const u32x2 lo = pack_bf16x4(a), hi = pack_bf16x4(b);
    const u32x4e w = {lo[0], lo[1], hi[0], hi[1]};
    return w;
  }

  // request band MTI's global operands into register set MTI & 1
  template <int MTI>
  __device__ __forceinline__ void prefetch(const B16Args& a) {
    constexpr int S = MTI & (NSETS - 1);
    if (LIN && !has_res) return;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int ch = a.ch_hidden;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const long long p = m_base + MTI * 32 + pass * RPS + rrow;
      const bool ok = p < a.M && any;
      if constexpr (LIN) {
        res32[S][pass] = zero4;
        if (ok && a.residual_bf16) {       // bf16 rows (cout % 4 == 0 on these launches): widened at once
          res32[S][pass] = widen(*reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(a.residual) + p * a.residual_ld + n));
        } else if (ok) {
          const float* rp = reinterpret_cast<const float*>(a.residual) + p * a.residual_ld + n;
          if (full) res32[S][pass] = *reinterpret_cast<const f32x4*>(rp);
          else {
            if (n + 0 < a.b_rows) res32[S][pass][0] = rp[0];
            if (n + 1 < a.b_rows) res32[S][pass][1] = rp[1];
            if (n + 2 < a.b_rows) res32[S][pass][2] = rp[2];
          }
        }
      } else {
        const u32x4e zero8 = {0u, 0u, 0u, 0u};
        res16[S][pass] = (ok && has_res) ? *reinterpret_cast<const u32x4e*>(reinterpret_cast<const __bf16*>(a.residual) + p * a.residual_ld + n) : zero8;
        if constexpr (ZR) {
          aux16[S][pass] = (ok && n >= ch) ? *reinterpret_cast<const u32x4e*>(reinterpret_cast<const __bf16*>(a.h_b16) + p * a.hb_ld + (n - ch)) : zero8;
        } else {
          aux16[S][pass] = ok ? *reinterpret_cast<const u32x4e*>(reinterpret_cast<const __bf16*>(a.aux_z) + p * ch + n) : zero8;
          h32[S][pass][0] = ok ? *reinterpret_cast<const f32x4*>(a.h + p * a.h_ld + n) : zero4;
          h32[S][pass][1] = ok ? *reinterpret_cast<const f32x4*>(a.h + p * a.h_ld + n + 4) : zero4;
        }
      }
    }
  }

  // band MTI: accumulators (+ bias) through LDS, fused arithmetic, stores
  template <int MTI>
  __device__ __forceinline__ void band(const B16Args& a, const f32x16 (&acc)[MT][NT], const float (&bias)[NT], float* reg, int lane) {
    constexpr int S = MTI & (NSETS - 1);
    const int col_l = lane & 31, row_l = (lane >> 5) * 4;
    const int ch = a.ch_hidden;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) reg[((r & 3) + 8 * (r >> 2) + row_l) * CW + nt * 32 + col_l] = acc[MTI][nt][r] + bias[nt];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: in-order LDS, no workgroup barrier needed
    if constexpr (LIN) {
      // bf16 rows without a residual (the update block's and most encoder launches): EIGHT columns per lane — one 16-byte store per
      // lane and pass instead of two 8-byte ones (the store tail of these epilogues is issue-bound: MI355X_MICROARCH.md, "8x dwordx4
      // halves it"); a wave instruction then writes 8 rows x 64..128 contiguous bytes
      if (a.out_vec16) {
        constexpr int LPR8 = CW / 8, RPS8 = 64 / LPR8, PASSES8 = 32 / RPS8;
        const int rrow8 = lane / LPR8, c8 = (lane % LPR8) * 8;
        const int n8 = n - c4 + c8;                     // (n = n_base + c4)
        const float sc = a.scale;
#pragma unroll
        for (int pass = 0; pass < PASSES8; ++pass) {
          const int row = pass * RPS8 + rrow8;
          f32x4 v0 = *reinterpret_cast<const f32x4*>(reg + row * CW + c8), v1 = *reinterpret_cast<const f32x4*>(reg + row * CW + c8 + 4);
          const long long p = m_base + MTI * 32 + row;
          if (p >= a.M || n8 >= a.b_rows) continue;
          if (a.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = (v0[e] < 0.f) ? 0.f : v0[e]; v1[e] = (v1[e] < 0.f) ? 0.f : v1[e]; }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] *= sc; v1[e] *= sc; }
          if (a.relu2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = (v0[e] < 0.f) ? 0.f : v0[e]; v1[e] = (v1[e] < 0.f) ? 0.f : v1[e]; }
          }
          __bf16* ob = reinterpret_cast<__bf16*>(a.out) + p * a.out_ld + a.out_coff + n8;
          if (n8 + 7 < a.b_rows) {
            const u32x2 lo = pack_bf16x4(v0), hi = pack_bf16x4(v1);
            typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
            const u32x4v w = {lo[0], lo[1], hi[0], hi[1]};
            *(__attribute__((address_space(1))) u32x4v*)ob = w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (n8 + e < a.b_rows) ((gbf16*)ob)[e] = (__bf16)v0[e];
              if (n8 + 4 + e < a.b_rows) ((gbf16*)ob)[4 + e] = (__bf16)v1[e];
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
      }
    }
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const int row = pass * RPS + rrow;
      f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * CW + c4);
      const long long p = m_base + MTI * 32 + row;
      if (p >= a.M || !any) continue;
      if constexpr (LIN) {
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];   // NaN-propagating like torch.relu
        }
        const float sc = a.scale;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= sc;
        if (has_res) v = res32[S][pass] + v;
        if (a.relu2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
        }
        if (a.out_bf16) {
          __bf16* ob = reinterpret_cast<__bf16*>(a.out) + p * a.out_ld + a.out_coff + n;
          if (full) *(gu32x2*)ob = pack_bf16x4(v);
          else {
            if (n + 0 < a.b_rows) ((gbf16*)ob)[0] = (__bf16)v[0];
            if (n + 1 < a.b_rows) ((gbf16*)ob)[1] = (__bf16)v[1];
            if (n + 2 < a.b_rows) ((gbf16*)ob)[2] = (__bf16)v[2];
          }
        } else {
          float* op = reinterpret_cast<float*>(a.out) + p * a.out_ld + a.out_coff + n;
          if (full) *(gf32x4*)op = v;
          else {
            if (n + 0 < a.b_rows) ((gfloat*)op)[0] = v[0];
            if (n + 1 < a.b_rows) ((gfloat*)op)[1] = v[1];
            if (n + 2 < a.b_rows) ((gfloat*)op)[2] = v[2];
          }
        }
      } else if constexpr (ZR) {     // cout = 2 * ch, ch % 8 == 0: eight columns never straddle z | r
        f32x4 w = *reinterpret_cast<const f32x4*>(reg + row * CW + c4 + 4);      // columns 4..7 of this lane (v holds 0..3)
        v += widen_lo(res16[S][pass]);
        w += widen_hi(res16[S][pass]);
        f32x4 g0, g1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { g0[e] = fast_sigmoid(v[e]); g1[e] = fast_sigmoid(w[e]); }
        if (n < ch) *(gu32x4e*)(reinterpret_cast<__bf16*>(a.aux_z) + p * ch + n) = pack_bf16x8(g0, g1);
        else *(gu32x4e*)(reinterpret_cast<__bf16*>(a.aux_rh) + p * ch + (n - ch)) =
            pack_bf16x8(g0 * widen_lo(aux16[S][pass]), g1 * widen_hi(aux16[S][pass]));
      } else {  // PFK_EPI_GRU_Q
        f32x4 w = *reinterpret_cast<const f32x4*>(reg + row * CW + c4 + 4);
        v += widen_lo(res16[S][pass]);
        w += widen_hi(res16[S][pass]);
        const f32x4 z0 = widen_lo(aux16[S][pass]), z1 = widen_hi(aux16[S][pass]), h0 = h32[S][pass][0], h1 = h32[S][pass][1];
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float q0 = fast_tanh(v[e]), q1 = fast_tanh(w[e]);
          o0[e] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, z0[e]), h0[e]), __fmul_rn(z0[e], q0));   // update.py:64,71
          o1[e] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, z1[e]), h1[e]), __fmul_rn(z1[e], q1));
        }
        *(gf32x4*)(a.h + p * a.h_ld + n) = o0;
        *(gf32x4*)(a.h + p * a.h_ld + n + 4) = o1;
        if (a.h_b16 != nullptr) *(gu32x4e*)(reinterpret_cast<__bf16*>(a.h_b16) + p * a.hb_ld + n) = pack_bf16x8(o0, o1);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next band overwrites the region
  }

  // after the K loop: band b + 1's operands are requested before band b's first store
  template <int... B>
  __device__ __forceinline__ void run(const B16Args& a, const f32x16 (&acc)[MT][NT], const float (&bias)[NT], float* reg, int lane,
                                      std::integer_sequence<int, B...>) {
    if constexpr (NSETS == 2) (((B + 1 < MT ? prefetch<(B + 1 < MT ? B + 1 : 0)>(a) : (void)0), band<B>(a, acc, bias, reg, lane)), ...);
    else ((band<B>(a, acc, bias, reg, lane), (B + 1 < MT ? prefetch<(B + 1 < MT ? B + 1 : 0)>(a) : (void)0)), ...);
  }
};

constexpr int b16_blocks_per_cu(int bm, int bn, int nst, int waves) {
  const int smem = nst * (bm + bn) * ROW16;
  int k = 160 * 1024 / smem;
  if (k * waves > 8) k = 8 / waves;       // keep >= 128 registers per lane
  return k < 1 ? 1 : k;
}

// One K-step of a wave as a hand-placed instruction stream: MFMA, filler, MFMA, filler ... (`sched_barrier(0)` after every item pins
// the order; a wave issues in order, so whatever does not sit between two MFMAs in program order runs with its matrix pipe idle, and
// an LDS-DMA piece costs 60-180 issue cycles — MI355X_MICROARCH.md).  A step has four 16-channel K-blocks of MT x NT MFMAs; fragments
// are register double-buffered per K-block: the reads of block kb + 1 are the first fillers of block kb, the step's LDS-DMA pieces are
// spread evenly over all MFMA slots, the scalar bookkeeping of the K iterator follows the last piece.  Only the first block's
// fragment reads (right behind the barrier) are exposed — the SIMD's other wave covers them.
template <class St, int MT, int NT, int ABL>
struct KStep16 {
  static constexpr int MPB = MT * NT, NM = 4 * MPB, NFR = MT + NT;
  static constexpr int RPS = (NFR + MPB - 1) / MPB;      // fragment reads per MFMA slot
  int a_row, b_row, ko[4];
  bf16x8 fa[2][MT], fb[2][NT];

  template <int KB, int E>
  __device__ __forceinline__ void read_one(const char* sc) {
    if constexpr (E < NFR) {       // order A0, B0, A1, B1, ...: the first MFMA's operands arrive first
      constexpr int pair = E / 2, second = E & 1;
      constexpr bool isA = (pair < MT && pair < NT) ? !second : (MT > NT);
      constexpr int idx = (pair < MT && pair < NT) ? pair : (E - (MT < NT ? MT : NT));
      if constexpr (isA) fa[KB & 1][idx] = *reinterpret_cast<const bf16x8*>(sc + a_row + idx * 32 * ROW16 + ko[KB]);
      else fb[KB & 1][idx] = *reinterpret_cast<const bf16x8*>(sc + b_row + idx * 32 * ROW16 + ko[KB]);
    }
  }
  template <int KB, int... E>
  __device__ __forceinline__ void read_block(const char* sc, std::integer_sequence<int, E...>) { (read_one<KB, E>(sc), ...); }

  template <int Q>
  __device__ __forceinline__ void slot(f32x16 (&acc)[MT][NT], St& st, const char* sc, char* nx) {
    constexpr int KB = Q / MPB, R = Q % MPB, mt = R / NT, nt = R % NT;
    if constexpr (ABL != 1) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[KB & 1][mt], fb[KB & 1][nt], acc[mt][nt], 0, 0, 0);
    else acc[mt][nt][0] += (float)fa[KB & 1][mt][0] * (float)fb[KB & 1][nt][0];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KB + 1 < 4 && ABL != 4) {
      if constexpr (R * RPS < NFR) {
        read_one<KB + 1, R * RPS>(sc);
        if constexpr (RPS > 1) read_one<KB + 1, R * RPS + 1>(sc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (ABL != 2) {
      // piece k sits behind MFMA floor((2k + 1) NM / (2 PIECES))
      constexpr int P = St::PIECES;
      constexpr int k = ((2 * Q + 1) * P) / (2 * NM);          // candidate piece for this slot
      if constexpr (k < P && ((2 * k + 1) * NM) / (2 * P) == Q) {
        if constexpr (k == 0) st.begin_issue();
        st.template piece_at<k>(nx);
        if constexpr (k == P - 1) st.end_issue();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  template <int... Q>
  __device__ __forceinline__ void slots(f32x16 (&acc)[MT][NT], St& st, const char* sc, char* nx, std::integer_sequence<int, Q...>) {
    (slot<Q>(acc, st, sc, nx), ...);
  }
  __device__ __forceinline__ void run(f32x16 (&acc)[MT][NT], St& st, const char* sc, char* nx) {
    if constexpr (ABL != 4) read_block<0>(sc, std::make_integer_sequence<int, NFR>{});
    __builtin_amdgcn_sched_barrier(0);
    slots(acc, st, sc, nx, std::make_integer_sequence<int, NM>{});
  }
};

// ABL (timing ablations behind PFK_BENCH_VARIANTS, results are garbage): 1 = no MFMAs, 2 = no DMA inside the K loop, 3 = no epilogue,
// 4 = no fragment reads inside the K loop, 5 = no workgroup barrier
template <int EPI, int BM, int BN, int WM, int WN, int NST, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN, b16_blocks_per_cu(BM, BN, NST, WM * WN) * WM * WN / 4 > 0 ? b16_blocks_per_cu(BM, BN, NST, WM * WN) * WM * WN / 4 : 1)
void conv_gemm_b16_kernel(const B16Args a_in) {
  B16Args a = a_in;
  if (a.batches > 1) {        // batched GEMM: this block's problem
    const long long by = blockIdx.y;
    a.src[0] = static_cast<const char*>(a.src[0]) + by * a.src_bs;
    a.weight = static_cast<const char*>(a.weight) + by * a.w_bs;
    a.out = static_cast<char*>(a.out) + by * a.out_bs;
    if (a.residual != nullptr) a.residual = static_cast<const char*>(a.residual) + by * a.res_bs;
  }
  constexpr int TH = 64 * WM * WN;
  using St = Stager16<BM, BN, TH>;
  constexpr int MT = BM / (32 * WM), NT = BN / (32 * WN);
  constexpr int STAGE = St::STAGE, A_BYTES = St::A_BYTES;
  static_assert(NST >= 2 && NST <= 4, "stages");
  extern __shared__ __attribute__((aligned(16))) char smem_b16[];   // [NST][STAGE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wid / WN) * (BM / WM);
  const int wn0 = (wid % WN) * (BN / WN);

  const int bid = pfk_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n;
  const int tile_m = bid / a.tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  St st(a, m0, n0, tid);
  st.abl = ABL;
  const int nsteps = st.total;

  f32x16 acc[MT][NT];
  zero_acc<MT, NT>(acc);

  // fragment addresses: row (lane & 31) of a 32-row block, physical chunk (2 kb + (lane >> 5)) ^ key
  const int frow = lane & 31, hl = lane >> 5, key = (lane >> 1) & 7;
  int ko[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) ko[kb] = ((2 * kb + hl) ^ key) << 4;
  const int a_row = (wm0 + frow) * ROW16;
  const int b_row = A_BYTES + (wn0 + frow) * ROW16;

  // the bias of this lane's columns first (accumulator layout: one column per 32x32 block); the compiler barrier keeps these loads
  // OLDER than every LDS-DMA piece, so that the counted waits below count pieces only
  Epilogue16<MT, NT, EPI> ep(a, m0 + wm0, n0 + wn0, lane);
  float bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int nb = n0 + wn0 + nt * 32 + (lane & 31);
    bias[nt] = (a.bias != nullptr && nb < a.b_rows) ? a.bias[nb] : 0.f;
  }
  asm volatile("" ::: "memory");
  // prologue: steps 0 .. NST-2 in flight, step 0 landed; behind them the requests for the first epilogue band's global operands
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) st.issue(smem_b16 + s * STAGE);
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * St::PIECES) : "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (ABL != 3) ep.template prefetch<0>(a);
  asm volatile("" ::: "memory");

  int cur = 0, nxt = NST - 1;       // stage read by this step / stage the step's DMA fills
  KStep16<St, MT, NT, ABL> ks{a_row, b_row, {ko[0], ko[1], ko[2], ko[3]}};
  for (int step = 0; step < nsteps; ++step) {
    ks.run(acc, st, smem_b16 + cur * STAGE, smem_b16 + nxt * STAGE);
    // step + 1 has landed (this wave's pieces; the barrier extends that to every wave's), the pieces issued above stay in flight;
    // this wave's fragment reads of `cur` are done before any wave's next DMA overwrites it
    if constexpr (ABL == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (ABL == 5) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NST - 2) * St::PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((NST - 2) * St::PIECES) : "memory");
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing may land in the epilogue's LDS
  __builtin_amdgcn_s_barrier();
  if constexpr (ABL == 3) {
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[mt][nt][r];
    if (sum == 123.456f) reinterpret_cast<float*>(a.out)[0] = sum;
  }
  else {
    float* reg = reinterpret_cast<float*>(smem_b16) + wid * (32 * NT * 32);
    ep.run(a, acc, bias, reg, lane, std::make_integer_sequence<int, MT>{});
  }
}

template <int EPI, int BM, int BN, int WM, int WN, int NST, int ABL = 0>
int launch_b16_one(const B16Args& a0, hipStream_t st) {
  B16Args a = a0;
  const long long tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.b_rows + BN - 1) / BN;
  const long long nblk = tiles_m * a.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  constexpr size_t smem = (size_t)NST * (BM + BN) * ROW16;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static_assert(WM * WN * 4096 * (BN / (32 * WN)) <= (int)smem, "the LDS epilogue needs 4 KB per wave and 32 columns");
  auto kern = conv_gemm_b16_kernel<EPI, BM, BN, WM, WN, NST, ABL>;
  static pfk_device_once attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)(a.batches > 1 ? a.batches : 1)), dim3(64 * WM * WN), smem, st, a);
  return pfk_launch_status();
}

template <int BM, int BN, int WM, int WN, int NST>
int launch_b16_epi(const B16Args& a, int epi, hipStream_t st) {
  switch (epi) {
    case PFK_EPI_LINEAR: return launch_b16_one<PFK_EPI_LINEAR, BM, BN, WM, WN, NST>(a, st);
    case PFK_EPI_GRU_ZR: return launch_b16_one<PFK_EPI_GRU_ZR, BM, BN, WM, WN, NST>(a, st);
    case PFK_EPI_GRU_Q:  return launch_b16_one<PFK_EPI_GRU_Q, BM, BN, WM, WN, NST>(a, st);
    default: return PFK_ERR_BAD_ARG;
  }
}

int g_b16_cfg = 0;       // pfk_debug_set_b16(cfg): 0 = heuristic

// tile configurations: 1 = 256x128 / 8 waves / 3 stages (144 KB, one block per CU), 2 = 128x128 / 4 waves / 2 stages (64 KB, two),
// 3 = 256x64 / 8 waves / 3 stages (120 KB), 4 = 128x64 / 4 waves / 3 stages (72 KB, two), 5 = 128x128 / 4 waves / 3 stages (96 KB, one),
// 6 = 256x256 / 8 waves / 2 stages (128 KB, one), 8 = 128x128 / 8 waves / 2 stages (64 KB, two), 9 = 256x64 / 8 waves / 2 stages (80 KB, two)
int launch_b16(const B16Args& a0, int epi, hipStream_t st) {
  B16Args a = a0;
  int cfg = g_b16_cfg;
  if (cfg >= 50 && cfg < 60) { cfg -= 50; a.out_vec16 = 0; }      // tuning knob: 50 + cfg = the same configuration with 8-byte epilogue stores
  if (cfg == 0) {
    // Measured on MI355X (scripts/conv_b16_bench.py, RAFT update-block shapes at 55x128, gpurun_out/r6g_pref.log batch 8, r6a_b16_b1.log
    // batch 1).  These launches are as much HBM / L2-stream bound as matrix bound, so the tile that moves the fewest operand bytes per
    // MFMA wins once the grid fills the chip: 256x256 (eight waves, 128x64 wave tiles; z|r 59 vs 62 us, fh|mask conv1 77.5 vs 83,
    // convc2 65.5 vs 70 although a quarter of its columns is padding) — except on short K with padded columns, where the epilogue
    // dominates (mask conv2, 4 K-steps: 49.6 us on 128x128 vs 52.6); cout <= 128: 256x128; cout 64: 256x64 (15.6 vs 21 us);
    // grids under ~200 big tiles (batch 1): 128x64 with two blocks per CU (177 us per iteration against 208-296 for the others).
    const int pad64 = (a.b_rows + 63) / 64 * 64, pad128 = (a.b_rows + 127) / 128 * 128;
    const long long tm = ((a.M + 255) / 256) * (a.batches > 1 ? a.batches : 1);
    // Round 6, the encoders' grids (many rounds of tiles, K = 9 or 2 steps): with ONE block per CU nothing runs under a tile's
    // prologue and epilogue — ablations on 64 -> 64 3x3 at 16 x 218 x 512: no epilogue 275 -> 195 us, no MFMAs 279 (!), A for the first
    // tap only 265 — so the configurations with TWO resident blocks win there: 256x64 x 2 stages (cfg 9: 294 -> 225 us; 164 -> 116 at 8
    // images), 128x128 / eight waves / 2 stages (cfg 8) for cout <= 128 (96 -> 96 at 16 x 109 x 256: 161.6 -> 146.0; 128 -> 128 at
    // 16 x 55 x 128: 50.6 -> 48.6) and for every short-K LINEAR launch (1x1 128 -> 256: 34.4 -> 28.8; mask conv2 50.2 -> 44.0) —
    // gpurun_out/r6y_l1.log, r6z_l1.log, r6z2.log.  The update block's batch-8 grids (220 row tiles) keep their tiles (convf2 16.9 vs
    // 17.8, conv 39.1 vs 40.0, q 38 vs 49 us).
    const bool lin = epi == PFK_EPI_LINEAR;
    if (tm * (pad128 / 128) < 200) cfg = 4;
    else if (pad64 < pad128 && a.b_rows < 192) cfg = (lin && tm >= 512) ? 9 : 3;
    else if (lin && a.steps <= 4) cfg = 8;
    else if (a.b_rows <= 128) cfg = (lin && tm >= 400) ? 8 : 1;
    else if (epi == PFK_EPI_GRU_Q) cfg = 1;
    else cfg = (a.steps >= 6 || a.b_rows % 256 == 0) ? 6 : 2;
  }
#ifdef PFK_BENCH_VARIANTS
  if (cfg >= 10 && epi == PFK_EPI_LINEAR) {
    switch (cfg) {
      case 11: return launch_b16_one<PFK_EPI_LINEAR, 256, 128, 4, 2, 3, 1>(a, st);
      case 21: return launch_b16_one<PFK_EPI_LINEAR, 256, 128, 4, 2, 3, 2>(a, st);
      case 31: return launch_b16_one<PFK_EPI_LINEAR, 256, 128, 4, 2, 3, 3>(a, st);
      case 41: return launch_b16_one<PFK_EPI_LINEAR, 256, 128, 4, 2, 3, 4>(a, st);
      case 51: return launch_b16_one<PFK_EPI_LINEAR, 256, 128, 4, 2, 3, 5>(a, st);
      case 42: return launch_b16_one<PFK_EPI_LINEAR, 128, 128, 2, 2, 2, 4>(a, st);
      case 52: return launch_b16_one<PFK_EPI_LINEAR, 128, 128, 2, 2, 2, 5>(a, st);
      case 12: return launch_b16_one<PFK_EPI_LINEAR, 128, 128, 2, 2, 2, 1>(a, st);
      case 22: return launch_b16_one<PFK_EPI_LINEAR, 128, 128, 2, 2, 2, 2>(a, st);
      case 32: return launch_b16_one<PFK_EPI_LINEAR, 128, 128, 2, 2, 2, 3>(a, st);
      // the 256x256 tile (cfg 6): no MFMAs / no DMA / no epilogue / A fetched for the first tap only (zero-filled, not issued)
      case 16: return launch_b16_one<PFK_EPI_LINEAR, 256, 256, 2, 4, 2, 1>(a, st);
      case 26: return launch_b16_one<PFK_EPI_LINEAR, 256, 256, 2, 4, 2, 2>(a, st);
      case 36: return launch_b16_one<PFK_EPI_LINEAR, 256, 256, 2, 4, 2, 3>(a, st);
      case 66: return launch_b16_one<PFK_EPI_LINEAR, 256, 256, 2, 4, 2, 6>(a, st);
      case 76: return launch_b16_one<PFK_EPI_LINEAR, 256, 256, 2, 4, 2, 7>(a, st);
      case 61: return launch_b16_one<PFK_EPI_LINEAR, 256, 128, 4, 2, 3, 6>(a, st);
      // the 256x64 tile (cfg 3: cout 64 — the encoders' first layer, convf2): A for the first tap only / no DMA / no MFMAs / no epilogue
      case 63: return launch_b16_one<PFK_EPI_LINEAR, 256, 64, 4, 2, 3, 6>(a, st);
      case 23: return launch_b16_one<PFK_EPI_LINEAR, 256, 64, 4, 2, 3, 2>(a, st);
      case 13: return launch_b16_one<PFK_EPI_LINEAR, 256, 64, 4, 2, 3, 1>(a, st);
      case 33: return launch_b16_one<PFK_EPI_LINEAR, 256, 64, 4, 2, 3, 3>(a, st);
      default: return PFK_ERR_BAD_ARG;
    }
  }
#endif
  switch (cfg) {
    case 1: return launch_b16_epi<256, 128, 4, 2, 3>(a, epi, st);
    case 2: return launch_b16_epi<128, 128, 2, 2, 2>(a, epi, st);
    case 3: return launch_b16_epi<256, 64, 4, 2, 3>(a, epi, st);
    case 4: return launch_b16_epi<128, 64, 2, 2, 3>(a, epi, st);
    case 5: return launch_b16_epi<128, 128, 2, 2, 3>(a, epi, st);
    // 128x128 / EIGHT waves / 2 stages (64 KB): two blocks = 16 waves per CU.  Twice cfg 2's waves on the same tile buy 0-4 % (fm 79.8 vs
    // 83.4 us, 80.2 on the 256x256 tile; convc2 69.5 vs 69.8; z|r 55.2 vs 57.0; q slower; mask conv2 44.0 vs 50.2 — gpurun_out/r6u_cfg8.log):
    // residency is not what holds these launches back.  Not selected by the heuristic (mask conv2 runs fused, K13b).
    case 8: return launch_b16_epi<128, 128, 4, 2, 2>(a, epi, st);
    case 9: return launch_b16_epi<256, 64, 4, 2, 2>(a, epi, st);       // 256x64 / eight waves / TWO stages (80 KB): two blocks per CU
    case 6:       // (no q epilogue on this tile: its 128-row wave tile has four bands of fp32 h / z / context operands — spills; q has cout = Ch <= 128 anyway)
      if (epi == PFK_EPI_LINEAR) return launch_b16_one<PFK_EPI_LINEAR, 256, 256, 2, 4, 2>(a, st);
      if (epi == PFK_EPI_GRU_ZR) return launch_b16_one<PFK_EPI_GRU_ZR, 256, 256, 2, 4, 2>(a, st);
      return PFK_ERR_BAD_ARG;
    default: return PFK_ERR_BAD_ARG;
  }
}

// -------------------------------------------------------------------------------------------------
// K13b — mask head conv2 + softmax + convex upsampling on the K8b path (raft/update.py:152 + raft/raft.py:112-123), the bf16 form of
// pfk_gemm.hip::mask_upsample_kernel: same geometry (a block = 128 pixels x 16 sub-pixels x all nine taps: four waves stacked in M, wave
// tile 32 x 160 = five 32x32 accumulators, weight / bias rows pre-permuted to [quarter][tile][32] by the host), operands bf16 through
// LDS-DMA (64-channel K-steps, two stages of (128 + 160) x 128 B = 72 KB: two blocks per CU), products on v_mfma_f32_32x32x16_bf16.
// On this path the unfused pair writes 65 MB of bf16 logits per iteration at batch 8 and reads them back on a side stream next to the
// main stream's HBM-bound launches; here the logits never leave the registers.  Each logit is rounded to bf16 exactly where the unfused
// launch stores it ((acc + bias) * scale, round to nearest even) and the epilogue is the shared mask_upsample_combine: the 8x flow is
// bit-identical to pfk_conv2d_b16 (bf16 out) + pfk_convex_upsample_pm_b16.
// -------------------------------------------------------------------------------------------------
struct MuB16Args {
  const void* x; int x_ld, cin;          // bf16 [M][x_ld]
  const void* w;                         // bf16 [640][cin], rows permuted
  const float* bias;                     // [640] permuted, may be null
  float scale;
  const float* flow; int flow_ld;        // fp32 pixel-major flow
  float* out;                            // [B][2][8H][8W]
  int H, W;
  long long M;
  unsigned wo_mul, ho_mul; int wo_sh, ho_sh;
};

constexpr int MUB_BM = 128, MUB_BN = 160, MUB_ROWS = 640;

__global__ __launch_bounds__(256, 2) void mask_upsample_b16_kernel(const MuB16Args a) {
  constexpr int BM = MUB_BM, BN = MUB_BN, NT = 5;
  constexpr int A_BYTES = BM * ROW16, STAGE = (BM + BN) * ROW16;
  constexpr int A_PT = BM / 32, B_PT = BN / 32, PIECES = A_PT + B_PT;     // 32 rows per pass of the 256 threads
  extern __shared__ __attribute__((aligned(16))) char smem_mu[];           // [2][STAGE]; the epilogue's flow neighbourhoods afterwards

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = pfk_xcd_remap(blockIdx.x, gridDim.x);   // the four quarters of a pixel tile are neighbours: one A panel through one L2
  const int quarter = bid & 3;
  const long long m0 = (long long)(bid >> 2) * BM;
  const int n0 = quarter * BN;

  const __amdgpu_buffer_rsrc_t rsx = make_rsrc(a.x), rsw = make_rsrc(a.w);
  const int r0 = tid >> 3;
  const int lc8 = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;      // logical chunk of this lane's physical slot (rows of one thread share the key)
  const unsigned wave_off = (unsigned)wid * 1024u;
  unsigned avo[A_PT], bvo[B_PT];
#pragma unroll
  for (int i = 0; i < A_PT; ++i) {
    const long long p = m0 + r0 + 32 * i;
    avo[i] = p < a.M ? (unsigned)(p * a.x_ld + lc8) * 2u : OOB;
  }
#pragma unroll
  for (int i = 0; i < B_PT; ++i) bvo[i] = (unsigned)((n0 + r0 + 32 * i) * a.cin + lc8) * 2u;
  const int nsteps = a.cin / BK16;

  auto issue = [&](int step, char* stage) {
    const int off = step * (BK16 * 2);
#pragma unroll
    for (int i = 0; i < A_PT; ++i) dma16(rsx, stage + i * 32 * ROW16 + wave_off, avo[i], off);
#pragma unroll
    for (int i = 0; i < B_PT; ++i) dma16(rsw, stage + A_BYTES + i * 32 * ROW16 + wave_off, bvo[i], off);
  };

  f32x16 acc[1][NT];
  zero_acc<1, NT>(acc);
  const int frow = lane & 31, hl = lane >> 5, key = (lane >> 1) & 7;
  int ko[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) ko[kb] = ((2 * kb + hl) ^ key) << 4;
  const int a_row = (wid * 32 + frow) * ROW16;
  const int b_row = A_BYTES + frow * ROW16;

  issue(0, smem_mu);
  for (int step = 0; step < nsteps; ++step) {
    char* cur = smem_mu + (step & 1) * STAGE;
    if (step + 1 < nsteps) {
      issue(step + 1, smem_mu + ((step + 1) & 1) * STAGE);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");      // this step's pieces have landed, the next step's stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(cur + a_row + ko[kb]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(cur + b_row + nt * 32 * ROW16 + ko[kb]);
        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0][nt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // every wave is done with `cur` before the step after next lands in it
  }

  // ---- epilogue (pfk_gemm.hip::mask_upsample_kernel's, with the logits rounded to bf16 as the unfused launch stores them)
  f32x2* nf = reinterpret_cast<f32x2*>(smem_mu);     // [BM][9]: 8 x flow of every pixel's 3x3 neighbourhood, zero outside the image
  const unsigned H = (unsigned)a.H, W = (unsigned)a.W;
  for (int e = tid; e < BM * 9; e += 256) {
    const int pl = e / 9, k = e - pl * 9;
    const unsigned p = (unsigned)m0 + (unsigned)pl;
    f32x2 v = {0.f, 0.f};
    if ((long long)p < a.M) {
      const unsigned prow = fastdiv_u32(p, a.wo_mul, a.wo_sh);          // b*H + y
      const unsigned bimg = fastdiv_u32(prow, a.ho_mul, a.ho_sh);
      const int x = (int)(p - prow * W), y = (int)(prow - bimg * H);
      const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
      if ((unsigned)yy < H && (unsigned)xx < W) {
        const float* f = a.flow + ((size_t)(bimg * H + (unsigned)yy) * W + (unsigned)xx) * (size_t)a.flow_ld;
        v[0] = 8.0f * f[0];
        v[1] = 8.0f * f[1];
      }
    }
    nf[e] = v;
  }
  __syncthreads();

  const int col = lane & 31;
  const int odd = col >> 4;                          // this lane's own taps are 2j + odd
  const int s = quarter * 16 + (col & 15), sy = s >> 3, sx = s & 7;
  const int wm0 = wid * 32;
  float bias[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bias[j] = a.bias != nullptr ? a.bias[n0 + j * 32 + col] : 0.f;
  const float sc = a.scale;
  const size_t HW8 = (size_t)H * W * 64;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const int rr = r + odd;                          // the row this lane finishes
    const int row = wm0 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
    const unsigned p = (unsigned)m0 + (unsigned)row;
    float own[NT], oth[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float l0 = (float)(__bf16)mask_logit(acc[0][j][r], bias[j], sc), l1 = (float)(__bf16)mask_logit(acc[0][j][r + 1], bias[j], sc);
      own[j] = odd ? l1 : l0;                        // tap 2j + odd of my row
      oth[j] = __shfl_xor(odd ? l0 : l1, 16, 64);    // send the partner's row, receive tap 2j + (1 - odd) of mine
    }
    float m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int j = k >> 1;
      m[k] = ((k & 1) == odd) ? own[j] : oth[j];
    }
    float ox, oy;
    mask_upsample_combine(m, nf + row * 9, ox, oy);
    if ((long long)p < a.M) {
      const unsigned prow = fastdiv_u32(p, a.wo_mul, a.wo_sh);
      const unsigned bimg = fastdiv_u32(prow, a.ho_mul, a.ho_sh);
      const unsigned x = p - prow * W, y = prow - bimg * H;
      const size_t o = (size_t)(8 * y + (unsigned)sy) * (8 * W) + 8 * x + (unsigned)sx;
      a.out[((size_t)bimg * 2 + 0) * HW8 + o] = ox;
      a.out[((size_t)bimg * 2 + 1) * HW8 + o] = oy;
    }
  }
}

int b16_ktot(const pfk_conv_b16_desc* d) {
  if (!d || d->num_src < 1 || d->num_src > 3) return PFK_ERR_BAD_ARG;
  int k = 0;
  for (int s = 0; s < d->num_src; ++s) k += d->kh * d->kw * ((d->src[s].channels + BK16 - 1) / BK16 * BK16);
  return k;
}

}  // namespace

extern "C" {

int pfk_debug_set_b16(int cfg) {
  if (!pfk_debug_knobs_enabled()) return PFK_ERR_DISABLED;
  g_b16_cfg = cfg < 0 ? 0 : cfg;
  return PFK_OK;
}

int pfk_conv_ktot_b16(const pfk_conv_b16_desc* d) { return b16_ktot(d); }

int pfk_mask_upsample_b16(const void* x_bf16, int x_ld, int cin, const void* weight_perm_bf16, const float* bias_perm, float scale,
                          const float* flow_pm, int flow_ld, float* out, int B, int H, int W, pfk_stream_t stream) {
  if (!x_bf16 || !weight_perm_bf16 || !flow_pm || !out || B <= 0 || H <= 0 || W <= 0 || cin <= 0 || x_ld < cin || flow_ld < 2) return PFK_ERR_BAD_ARG;
  if ((cin & 63) || (x_ld & 7) || !pfk_aligned16(x_bf16) || !pfk_aligned16(weight_perm_bf16)) return PFK_ERR_ALIGNMENT;
  const long long M = (long long)B * H * W;
  if (M >= 0x7fffffffLL || M * x_ld * 2 >= 0x7fffffffLL || (long long)MUB_ROWS * cin * 2 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  MuB16Args a{};
  a.x = x_bf16; a.x_ld = x_ld; a.cin = cin; a.w = weight_perm_bf16; a.bias = bias_perm; a.scale = scale;
  a.flow = flow_pm; a.flow_ld = flow_ld; a.out = out; a.H = H; a.W = W; a.M = M;
  fastdiv_make((unsigned)W, a.wo_mul, a.wo_sh);
  fastdiv_make((unsigned)H, a.ho_mul, a.ho_sh);
  const long long nblk = ((M + MUB_BM - 1) / MUB_BM) * 4;
  if (nblk > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  constexpr size_t smem = 2 * (MUB_BM + MUB_BN) * ROW16;
  static_assert(MUB_BM * 9 * 8 <= (int)smem, "the epilogue's flow neighbourhoods reuse the staging LDS");
  static pfk_device_once attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mask_upsample_b16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL(mask_upsample_b16_kernel, dim3((unsigned)nblk), dim3(256), smem, static_cast<hipStream_t>(stream), a);
  return pfk_launch_status();
}

int pfk_conv2d_b16(const pfk_conv_b16_desc* d, pfk_stream_t stream) {
  if (!d || !d->weight || d->num_src < 1 || d->num_src > 3) return PFK_ERR_BAD_ARG;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cout <= 0) return PFK_ERR_BAD_ARG;
  if (d->kh <= 0 || d->kw <= 0 || !(d->kh & 1) || !(d->kw & 1) || d->kh * d->kw > 32) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(d->weight)) return PFK_ERR_ALIGNMENT;
  B16Args a{};
  for (int i = 0; i < d->num_src; ++i) {
    const pfk_conv_src_b16& s = d->src[i];
    if (!s.ptr || s.channels <= 0 || s.ld < s.channels) return PFK_ERR_BAD_ARG;
    if (!pfk_aligned16(s.ptr) || (s.ld & 7) || (s.channels & 7)) return PFK_ERR_ALIGNMENT;
    if ((long long)d->B * d->H * d->W * s.ld * 2 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;     // 32-bit byte offsets
    a.src[i] = s.ptr; a.ld[i] = s.ld; a.ch[i] = s.channels;
  }
  a.nsrc = d->num_src;
  const int stride = d->stride > 0 ? d->stride : 1;
  a.H = d->H; a.W = d->W; a.kh = d->kh; a.kw = d->kw; a.stride = stride;
  a.Ho = (d->H - 1) / stride + 1; a.Wo = (d->W - 1) / stride + 1;
  a.M = (long long)d->B * a.Ho * a.Wo;
  if (a.M >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  a.weight = d->weight; a.ktot = b16_ktot(d); a.b_rows = d->cout; a.steps = a.ktot / BK16;
  if ((long long)d->cout * a.ktot * 2 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  a.bias = d->bias;
  if (d->bias && !pfk_aligned16(d->bias)) return PFK_ERR_ALIGNMENT;
  a.relu = d->relu; a.relu2 = d->relu_after_residual; a.scale = d->scale;
  fastdiv_make((unsigned)a.Wo, a.wo_mul, a.wo_sh);
  fastdiv_make((unsigned)a.Ho, a.ho_mul, a.ho_sh);
  if (d->residual) {      // LINEAR: fp32 rows (bf16 with residual_bf16); GRU epilogues: bf16 rows
    if (d->residual_ld < d->cout) return PFK_ERR_BAD_ARG;
    const bool r16 = d->epilogue != PFK_EPI_LINEAR || d->residual_bf16;
    const bool gru = d->epilogue != PFK_EPI_LINEAR;      // the GRU epilogues read the term as 16-byte pieces
    if ((reinterpret_cast<uintptr_t>(d->residual) & ((r16 && !gru) ? 7u : 15u)) || (d->residual_ld & (gru ? 7 : 3)) || (r16 && (d->cout & 3)))
      return PFK_ERR_ALIGNMENT;
    a.residual = d->residual; a.residual_ld = d->residual_ld; a.residual_bf16 = d->epilogue == PFK_EPI_LINEAR && d->residual_bf16;
  }
  if (d->batches > 1) {
    // batched GEMM: `batches` independent problems of the same shape in one grid (blockIdx.y) — GMA's attn @ v per frame pair
    if (d->epilogue != PFK_EPI_LINEAR || d->num_src != 1 || d->kh != 1 || d->kw != 1 || d->B != 1) return PFK_ERR_BAD_ARG;
    if ((d->src_batch_stride & 7) || (d->weight_batch_stride & 7) || (d->out_batch_stride & 3) || (d->residual_batch_stride & 3)) return PFK_ERR_ALIGNMENT;
    a.batches = d->batches;
    a.src_bs = d->src_batch_stride * 2; a.w_bs = d->weight_batch_stride * 2;
    a.out_bs = d->out_batch_stride * (d->out_bf16 ? 2 : 4);
    a.res_bs = d->residual_batch_stride * (d->residual_bf16 ? 2 : 4);
  }
  switch (d->epilogue) {
    case PFK_EPI_LINEAR:
      if (!d->out || d->out_ld < d->out_coff + d->cout) return PFK_ERR_BAD_ARG;
      if ((d->out_ld & 3) || (d->out_coff & 3) || (reinterpret_cast<uintptr_t>(d->out) & (d->out_bf16 ? 7u : 15u))) return PFK_ERR_ALIGNMENT;
      a.out = d->out; a.out_ld = d->out_ld; a.out_coff = d->out_coff; a.out_bf16 = d->out_bf16;
      a.out_vec16 = d->out_bf16 && !d->residual && (d->batches <= 1 || (d->out_batch_stride & 7) == 0) &&
                    !((reinterpret_cast<uintptr_t>(d->out) & 15u) || (d->out_ld & 7) || (d->out_coff & 7));
      break;
    case PFK_EPI_GRU_ZR:      // (16-byte row pieces: Ch % 8 == 0, 16-byte aligned bf16 buffers, row strides multiples of 8 elements)
      if (!d->h_b16 || !d->aux_z || !d->aux_rh || (d->cout & 63)) return PFK_ERR_BAD_ARG;
      if ((reinterpret_cast<uintptr_t>(d->h_b16) & 15u) || (d->h_b16_ld & 7) || (reinterpret_cast<uintptr_t>(d->aux_z) & 15u) ||
          (reinterpret_cast<uintptr_t>(d->aux_rh) & 15u)) return PFK_ERR_ALIGNMENT;
      a.h_b16 = const_cast<void*>(d->h_b16); a.hb_ld = d->h_b16_ld; a.aux_z = d->aux_z; a.aux_rh = d->aux_rh; a.ch_hidden = d->cout / 2;
      break;
    case PFK_EPI_GRU_Q:
      if (!d->h || !d->aux_z || (d->cout & 31)) return PFK_ERR_BAD_ARG;
      if (!pfk_aligned16(d->h) || (d->h_ld & 3) || (reinterpret_cast<uintptr_t>(d->aux_z) & 15u)) return PFK_ERR_ALIGNMENT;
      if (d->h_b16 && ((reinterpret_cast<uintptr_t>(d->h_b16) & 15u) || (d->h_b16_ld & 7) || d->h_b16_ld < d->cout)) return PFK_ERR_ALIGNMENT;
      a.h = d->h; a.h_ld = d->h_ld; a.aux_z = d->aux_z; a.ch_hidden = d->cout;
      a.h_b16 = d->h_b16; a.hb_ld = d->h_b16_ld;
      break;
    default:
      return PFK_ERR_BAD_ARG;
  }
  return launch_b16(a, d->epilogue, static_cast<hipStream_t>(stream));
}

}  // extern "C"
