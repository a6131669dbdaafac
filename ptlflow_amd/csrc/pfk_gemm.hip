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
#include "pfk_gemm.h"
#ifndef PP_EXP
#define PP_EXP 0
#endif

#include <mutex>
#include <utility>

using namespace pfkg;

namespace {

constexpr int BK = 32;       // channels per K-step
constexpr int LDS_LD = 36;   // padded LDS row length (floats), 2-stage pipeline (v1)
constexpr int LDS_LDX = 32;  // un-padded rows + XOR swizzle of the 16-byte chunk index, 3-stage pipelines (v3, stream-K):
                             // 48 KB per 64x64 block instead of 55 KB => three resident blocks per CU instead of two

// -------------------------------------------------------------------------------------------------
// Staging: 256 threads move one K-step (BM + BN rows x 32 floats) global -> VGPR -> LDS.
// Thread t owns float4 column (t & 7) of rows (t >> 3) + 32*i.  The K iterator walks
// source -> tap (ky, kx) -> 32-channel chunk.
//
// All loads are raw *buffer* loads with 32-bit byte offsets: a lane whose tap falls outside the
// image (the convolution's zero padding), whose row is past M / cout, or whose channel chunk is
// past the source's channel count gets an out-of-range offset and the hardware returns zeros —
// no branches, no 64-bit address arithmetic in the K loop.  The buffer descriptors are built from
// kernel arguments only (provably wave-uniform, so hipcc emits no waterfall loops).
// -------------------------------------------------------------------------------------------------
template <int BM, int BN, int LD = LDS_LD>
struct Stager {
  static constexpr int A_PT = BM / 32;
  static constexpr int B_PT = BN / 32;
  // uniform
  int H, W, kh, kw, ph, pw, nsrc;
  int ld0, ld1, ld2, ch0, ch1, ch2;
  __amdgpu_buffer_rsrc_t rs0, rs1, rs2, rsw, rs;
  int cld, cch;
  int seg = 0, ky = 0, kx = 0, c0 = 0, kofs = 0;
  // per thread
  int c4, r0;
  int scol;   // LDS column of this thread's float4: c4, or its XOR-swizzled chunk when rows are un-padded (LD == 32)
  int prow[A_PT], py[A_PT], px[A_PT];
  bool pok[A_PT];
  unsigned abase[A_PT];   // byte offset of (row, channel c4) in the current source
  unsigned aoff[A_PT];    // abase + current tap's offset, or OOB when the tap falls outside the image for this row
  unsigned wvoff[B_PT];
  f32x4 ra[A_PT], rb[B_PT];

  struct NoTile {};
  // wave-uniform state and the thread's place in the staging pattern; the tile is chosen by retarget()
  __device__ __forceinline__ Stager(const GemmArgs& a, int t, NoTile) {
    H = a.H; W = a.W; kh = a.kh; kw = a.kw; ph = a.kh >> 1; pw = a.kw >> 1; nsrc = a.nsrc;
    ld0 = a.ld0; ld1 = a.ld1; ld2 = a.ld2; ch0 = a.ch0; ch1 = a.ch1; ch2 = a.ch2;
    rs1 = make_rsrc(a.src1);
    rs2 = make_rsrc(a.src2);
    c4 = (t & 7) * 4;
    r0 = t >> 3;
    // LD == 32: chunk' = chunk ^ ((row >> 1) & 7) — a 16-lane ds_read_b128 group (16 rows, one logical chunk) then
    // covers all 16 sixteen-byte slots of the 256-byte bank row: conflict-free without padding.  (row + 32*i keeps the key.)
    scol = LD == 32 ? ((((t & 7) ^ ((r0 >> 1) & 7))) << 2) : c4;
  }

  __device__ __forceinline__ Stager(const GemmArgs& a, long long m0, int n0, int t, long long batch) : Stager(a, t, NoTile{}) {
    // multiplier arithmetic (launch() fills the multipliers) instead of six 64-bit divisions: 640 instructions fewer at the head of
    // every tile — which an A/B on the MI355X priced at 0.3 % (4 % on the 8-step mask conv2): the co-resident blocks hide them
    retarget(a, m0, n0, batch);
    set_segment(0);
    set_tap();
  }

  // Point the stager at output tile (m0, n0) of batch element `batch`: per-row pixel coordinates / base rows, weight-row offsets
  // and the batch-dependent descriptors.  32-bit arithmetic with the host-made multipliers of GemmArgs (M < 2^31 rows is implied
  // by the 32-bit byte offsets the kernels address their sources with; launch() / launch_pp() fill the multipliers).
  __device__ __forceinline__ void retarget(const GemmArgs& a, long long m0, int n0, long long batch) {
    rs0 = make_rsrc(a.src0 + batch * a.a_bs);
    rsw = make_rsrc(a.weight + batch * a.b_bs);
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const unsigned p = (unsigned)m0 + (unsigned)(r0 + 32 * i);
      pok[i] = (long long)p < a.M;
      const unsigned prow_o = fastdiv_u32(p, a.wo_mul, a.wo_sh);          // b*Ho + yo
      const unsigned bimg = fastdiv_u32(prow_o, a.ho_mul, a.ho_sh);
      px[i] = (int)(p - prow_o * (unsigned)a.Wo) * a.stride;              // input coordinates of the centre tap
      py[i] = (int)(prow_o - bimg * (unsigned)a.Ho) * a.stride;
      prow[i] = (int)((bimg * (unsigned)a.H + (unsigned)py[i]) * (unsigned)a.W + (unsigned)px[i]);
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int n = n0 + r0 + 32 * i;
      wvoff[i] = n < a.b_rows ? (unsigned)(n * a.ktot + c4) * 4u : OOB;
    }
  }

  // K iterator back to the first K-step of the (re-targeted) tile
  __device__ __forceinline__ void rewind() {
    seg = 0; ky = 0; kx = 0; c0 = 0; kofs = 0;
    set_segment(0);
    set_tap();
  }

  __device__ __forceinline__ void set_segment(int s) {
    if (s == 0) { rs = rs0; cld = ld0; cch = ch0; }
    else if (s == 1) { rs = rs1; cld = ld1; cch = ch1; }
    else { rs = rs2; cld = ld2; cch = ch2; }
#pragma unroll
    for (int i = 0; i < A_PT; ++i) abase[i] = (unsigned)(prow[i] * cld + c4) * 4u;
  }

  // Per-tap work (once every cch/32 K-steps): zero-padding predicate and tap offset folded into one VGPR per row.
  __device__ __forceinline__ void set_tap() {
    const int dy = ky - ph, dx = kx - pw;
    const unsigned toff = (unsigned)((dy * W + dx) * cld * 4);
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const bool ok = pok[i] && (unsigned)(py[i] + dy) < (unsigned)H && (unsigned)(px[i] + dx) < (unsigned)W;
      aoff[i] = ok ? abase[i] + toff : OOB;
    }
  }

  // Position the iterator at K-step `step` (stream-K segments start mid-tile).
  __device__ __forceinline__ void seek(int step) {
    const int taps = kh * kw;
    int r = step, sg = 0, cps = (ch0 + BK - 1) / BK;
    if (nsrc > 1 && r >= taps * cps) {
      r -= taps * cps; sg = 1; cps = (ch1 + BK - 1) / BK;
      if (nsrc > 2 && r >= taps * cps) { r -= taps * cps; sg = 2; cps = (ch2 + BK - 1) / BK; }
    }
    const int tap = r / cps;
    seg = sg; ky = tap / kw; kx = tap - ky * kw; c0 = (r - tap * cps) * BK; kofs = step * BK;
    set_segment(sg);
    set_tap();
  }

  // seek() for the persistent kernel, whose stager lives across tile boundaries: there the segment's parameters are read from
  // the kernel-argument block at a runtime offset (scalar loads) — selecting among the struct's own members by a runtime index
  // made hipcc keep them in a scratch array and lose the uniformity of everything derived from it (waterfall loops around
  // every buffer load of the K loop).  GemmArgs keeps src0..2 / ld0..2 / ch0..2 contiguous for this.
  __device__ __forceinline__ void seek_args(const GemmArgs& a, long long batch, int step) {
    const int taps = kh * kw;
    int r = step, sg = 0, cps = (a.ch0 + BK - 1) / BK;
    if (nsrc > 1 && r >= taps * cps) {
      r -= taps * cps; sg = 1; cps = (a.ch1 + BK - 1) / BK;
      if (nsrc > 2 && r >= taps * cps) { r -= taps * cps; sg = 2; cps = (a.ch2 + BK - 1) / BK; }
    }
    const int tap = r / cps;
    seg = sg; ky = tap / kw; kx = tap - ky * kw; c0 = (r - tap * cps) * BK; kofs = step * BK;
    const float* const* srcv = &a.src0;
    const float* p = srcv[sg];
    if (sg == 0) p += batch * a.a_bs;
    rs = make_rsrc(p);
    cld = (&a.ld0)[sg];
    cch = (&a.ch0)[sg];
#pragma unroll
    for (int i = 0; i < A_PT; ++i) abase[i] = (unsigned)(prow[i] * cld + c4) * 4u;
    set_tap();
  }

  __device__ __forceinline__ int total_steps() const {
    const int taps = kh * kw;
    int s = taps * ((ch0 + BK - 1) / BK);
    if (nsrc > 1) s += taps * ((ch1 + BK - 1) / BK);
    if (nsrc > 2) s += taps * ((ch2 + BK - 1) / BK);
    return s;
  }

  // Per-step work: the channel chunk goes into the scalar offset of the buffer instruction; the only vector work
  // is the partial-chunk / dead-step predicate (1 compare + 1 select per row).  `live` = false turns every A lane
  // out-of-range (zeros) and parks the B loads on K-step 0 (valid memory, multiplied by zeros), which keeps the
  // K loop body branch-free.
  __device__ __forceinline__ void load(bool live = true) {
    load_setup(live);
#pragma unroll
    for (int i = 0; i < A_PT; ++i)
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, p_cok ? aoff[i] : OOB, p_coff, 0));
#pragma unroll
    for (int i = 0; i < B_PT; ++i)
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], p_koff, 0));
  }

  // ---- the same work in small pieces, for hand-placed interleaving with MFMAs (variant 3) -------------
  int p_coff, p_koff;
  bool p_cok;
  __device__ __forceinline__ void load_setup(bool live) {
    const int lim = live ? cch - c0 : 0;
    p_cok = c4 < lim;
    p_coff = c0 * 4;
    p_koff = live ? kofs * 4 : 0;
  }
  template <int i>
  __device__ __forceinline__ void load_a() {
    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, p_cok ? aoff[i] : OOB, p_coff, 0));
  }
  template <int i>
  __device__ __forceinline__ void load_b() {
    rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], p_koff, 0));
  }
  // piece q in [0, A_PT + B_PT): one ds_write_b128 of the staged registers into `stage`
  template <int q>
  __device__ __forceinline__ void store_piece(float* stage) const {
    if constexpr (q < A_PT) *reinterpret_cast<f32x4*>(stage + (r0 + 32 * q) * LD + scol) = ra[q];
    else *reinterpret_cast<f32x4*>(stage + BM * LD + (r0 + 32 * (q - A_PT)) * LD + scol) = rb[q - A_PT];
  }

  __device__ __forceinline__ void advance() {
    kofs += BK;
    c0 += BK;
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

  __device__ __forceinline__ void store(float* dA, float* dB) const {
#pragma unroll
    for (int i = 0; i < A_PT; ++i)
      *reinterpret_cast<f32x4*>(dA + (r0 + 32 * i) * LD + scol) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PT; ++i)
      *reinterpret_cast<f32x4*>(dB + (r0 + 32 * i) * LD + scol) = rb[i];
  }
};

// One K-step (32 channels) of MFMAs for a wave tile of MT x NT 32x32 blocks, operands from LDS.
// Split in two so the caller can issue the next K-step's address arithmetic + buffer loads right
// after the first fragment reads (they then run in the shadow of the LDS latency / first MFMAs).
// NB: no lambdas here — capturing the Stager by reference forces it into scratch memory and turns
// its wave-uniform state into VGPRs (waterfall loops around every buffer load).
template <int MT, int NT>
struct Frags {
  f32x4 fa[MT], fb[NT];
};

// Fragment read for sub-step kk: lane l wants row (l & 31), logical 16-byte chunk kk*2 + (l >> 5); `ko[kk]` is that
// chunk's float offset inside the row (swizzled or not), precomputed per lane by frag_offsets().
template <int MT, int NT, int LD>
__device__ __forceinline__ void frag_read(Frags<MT, NT>& f, const float* cA, const float* cB, const int (&ko)[4], int kk) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) f.fa[mt] = *reinterpret_cast<const f32x4*>(cA + mt * 32 * LD + ko[kk]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) f.fb[nt] = *reinterpret_cast<const f32x4*>(cB + nt * 32 * LD + ko[kk]);
}

template <int LD>
__device__ __forceinline__ void frag_offsets(int (&ko)[4], int lane) {
  const int hl = lane >> 5;
  const int key = LD == 32 ? (((lane & 31) >> 1) & 7) : 0;   // (row >> 1) & 7; wave/tile row offsets are multiples of 32
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) ko[kk] = ((kk * 2 + hl) ^ key) << 2;
}

template <int MT, int NT>
__device__ __forceinline__ void mma_one(f32x16 (&acc)[MT][NT], const Frags<MT, NT>& f) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.fa[mt][s], f.fb[nt][s], acc[mt][nt], 0, 0, 0);
}

// Remaining three sub-steps after the caller read sub-step 0: reads for kk+1 are pinned in front of the
// MFMAs of kk so the LDS latency hides under the matrix pipe.
template <int MT, int NT>
__device__ __forceinline__ void mma_rest(f32x16 (&acc)[MT][NT], Frags<MT, NT>& f, const float* cA, const float* cB,
                                         const int (&ko)[4]) {
  Frags<MT, NT> g;
  frag_read<MT, NT, LDS_LD>(g, cA, cB, ko, 1);
  __builtin_amdgcn_sched_barrier(0);
  mma_one<MT, NT>(acc, f);
  __builtin_amdgcn_sched_barrier(0);
  frag_read<MT, NT, LDS_LD>(f, cA, cB, ko, 2);
  __builtin_amdgcn_sched_barrier(0);
  mma_one<MT, NT>(acc, g);
  __builtin_amdgcn_sched_barrier(0);
  frag_read<MT, NT, LDS_LD>(g, cA, cB, ko, 3);
  __builtin_amdgcn_sched_barrier(0);
  mma_one<MT, NT>(acc, f);
  __builtin_amdgcn_sched_barrier(0);
  mma_one<MT, NT>(acc, g);
}

// -------------------------------------------------------------------------------------------------
// Variant 1: 4 waves, one K pipeline, LDS double buffer (next step's loads in flight during the MFMAs).
// Good when the grid has several blocks per CU to overlap with.
// -------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const GemmArgs a) {
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
  constexpr int MT = WM / 32, NT = WN / 32;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                          // [2][BM][LDS_LD]
  float* sB = smem + 2 * BM * LDS_LD;        // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm0 = (wid / WAVES_N) * WM;
  const int wn0 = (wid % WAVES_N) * WN;

  const int bid = pfk_xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  tile_of(bid, (int)gridDim.x / a.tiles_n, a.tiles_n, a.supertile, tile_m, tile_n);
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;
  const long long batch = blockIdx.y;

  Stager<BM, BN> st(a, m0, n0, tid, batch);
  const int total_steps = st.total_steps();

  f32x16 acc[MT][NT];
  zero_acc<MT, NT>(acc);

  st.load();
  st.advance();
  st.store(sA, sB);
  __syncthreads();

  const int frow = lane & 31;
  int ko[4];
  frag_offsets<LDS_LD>(ko, lane);

  for (int step = 0; step < total_steps; ++step) {
    const int buf = step & 1;
    const bool more = (step + 1) < total_steps;
    const float* cA = sA + buf * BM * LDS_LD + (wm0 + frow) * LDS_LD;
    const float* cB = sB + buf * BN * LDS_LD + (wn0 + frow) * LDS_LD;
    Frags<MT, NT> fr;
    frag_read<MT, NT, LDS_LD>(fr, cA, cB, ko, 0);
    st.load(more);                      // branch-free; all lanes out of range on the last step
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in front of the MFMAs (hipcc otherwise sinks it to its use)
    mma_rest<MT, NT>(acc, fr, cA, cB, ko);
    __builtin_amdgcn_sched_barrier(0);
    st.store(sA + (buf ^ 1) * BM * LDS_LD, sB + (buf ^ 1) * BN * LDS_LD);   // last step: zeros into a dead buffer
    if (more) st.advance();
    __syncthreads();
  }
  epilogue_lds<MT, NT, EPI>(a, acc, m0 + wm0, n0 + wn0, lane, batch, smem, wid);   // behind the loop's final barrier
}

// -------------------------------------------------------------------------------------------------
// K13: fused mask head conv2 + softmax + 8x convex upsampling (raft/update.py:138-142,152 + raft/raft.py:112-123), round 5.
//
// The reference's loop writes a [B, 576, h, w] mask (mask[k*64 + sy*8 + sx] for tap k of the 3x3 neighbourhood and sub-pixel
// (sy, sx)) and re-reads it for softmax over k + the convex combination: 16 MB per pair and iteration through HBM, 32 times.
// Here ONE kernel does the 1x1 convolution and consumes its output in registers.  A block owns 128 pixels x 16 sub-pixels and ALL
// NINE taps: four waves stacked in M, wave tile 32 pixels x 160 columns = five 32x32 accumulators, column c of tile j holding tap
// 2j + (c >> 4) of sub-pixel c & 15 (the tenth "tap" is a zero weight row).  The nine logits of a (pixel, sub-pixel) therefore sit
// in TWO lanes (c and c ^ 16) of the same accumulator registers: one cross-lane exchange per register hands both lanes all nine,
// and softmax + the weighted sum of the neighbours' flows are lane-local from there (the two lanes split the rows between them).
// The mask never exists in memory.
// (First form of this kernel: 32 sub-pixels x nine full tiles per wave — no exchange, no pad row — needed 117 KB of LDS and 144
//  accumulator registers: ONE block per CU, so nothing ran under its prologue or under the 144 expf per lane of its epilogue, and a
//  block could not share a CU with the other stream's 48-72 KB blocks: 0.50 of the matrix peak standalone, slower in situ.  This
//  form is 72 KB and <= 256 registers: two blocks per CU.)
// The host permutes the weight / bias rows to [quarter (4)][tile (5)][32] (zero rows for the pad tap), which makes the block's B
// rows the plain range [quarter*160, quarter*160 + 160) of the 2-stage kernel's stager.  K loop = variant 1's on the swizzled
// layout (same K order per output element as every other tile shape: the logits are bit-identical to the unfused launch's), 8
// K-steps of 80 MFMAs per wave for the 256 input channels; LDS 2 x (128 + 160) x 32 floats = 72 KB.  Epilogue arithmetic = the
// unfused pair's, operation for operation => bit-identical flow.
// -------------------------------------------------------------------------------------------------
// (mask_upsample_combine / mask_logit — one (pixel, sub-pixel) of the fused epilogue — live in pfk_gemm.h: the K8b form of this kernel
//  in pfk_gemm_b16.hip shares them)
template <int MT, int NT, int LD>
__device__ __forceinline__ void mma_rest_ld(f32x16 (&acc)[MT][NT], Frags<MT, NT>& f, const float* cA, const float* cB,
                                            const int (&ko)[4]) {
  Frags<MT, NT> g;
  frag_read<MT, NT, LD>(g, cA, cB, ko, 1);
  __builtin_amdgcn_sched_barrier(0);
  mma_one<MT, NT>(acc, f);
  __builtin_amdgcn_sched_barrier(0);
  frag_read<MT, NT, LD>(f, cA, cB, ko, 2);
  __builtin_amdgcn_sched_barrier(0);
  mma_one<MT, NT>(acc, g);
  __builtin_amdgcn_sched_barrier(0);
  frag_read<MT, NT, LD>(g, cA, cB, ko, 3);
  __builtin_amdgcn_sched_barrier(0);
  mma_one<MT, NT>(acc, f);
  __builtin_amdgcn_sched_barrier(0);
  mma_one<MT, NT>(acc, g);
}

constexpr int MU_BM = 128, MU_BN = 160, MU_ROWS = 640;   // 4 quarters x 5 tiles x 32 weight rows (9 real taps + 1 zero tap per 16 sub-pixels)

__global__ __launch_bounds__(256, 2) void mask_upsample_kernel(const GemmArgs a) {
  constexpr int BM = MU_BM, BN = MU_BN, MT = 1, NT = 5, LD = LDS_LDX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                      // [2][BM][LD]
  float* sB = smem + 2 * BM * LD;        // [2][BN][LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm0 = wid * 32;

  const int bid = pfk_xcd_remap(blockIdx.x, gridDim.x);   // the four quarters of a pixel tile are neighbours: one A panel through one L2
  const int quarter = bid & 3;
  const long long m0 = (long long)(bid >> 2) * BM;
  const int n0 = quarter * BN;

  Stager<BM, BN, LD> st(a, m0, n0, tid, 0);
  const int total_steps = st.total_steps();

  f32x16 acc[MT][NT];
  zero_acc<MT, NT>(acc);

  st.load();
  st.advance();
  st.store(sA, sB);
  __syncthreads();

  const int frow = lane & 31;
  int ko[4];
  frag_offsets<LD>(ko, lane);

  for (int step = 0; step < total_steps; ++step) {
    const int buf = step & 1;
    const bool more = (step + 1) < total_steps;
    const float* cA = sA + buf * BM * LD + (wm0 + frow) * LD;
    const float* cB = sB + buf * BN * LD + frow * LD;
    Frags<MT, NT> fr;
    frag_read<MT, NT, LD>(fr, cA, cB, ko, 0);
    st.load(more);
    __builtin_amdgcn_sched_barrier(0);
    mma_rest_ld<MT, NT, LD>(acc, fr, cA, cB, ko);
    __builtin_amdgcn_sched_barrier(0);
    st.store(sA + (buf ^ 1) * BM * LD, sB + (buf ^ 1) * BN * LD);
    if (more) st.advance();
    __syncthreads();
  }

  // ---- epilogue.  8 * flow of every pixel's 3x3 neighbourhood (zero outside the image: F.unfold's padding) -> LDS
  f32x2* nf = reinterpret_cast<f32x2*>(smem);     // [BM][9]; every stage read sits behind the loop's final barrier
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
        const float* f = a.mu_flow + ((size_t)(bimg * H + (unsigned)yy) * W + (unsigned)xx) * (size_t)a.mu_flow_ld;
        v[0] = 8.0f * f[0];      // exact (a power of two)
        v[1] = 8.0f * f[1];
      }
    }
    nf[e] = v;
  }
  __syncthreads();

  const int col = lane & 31;
  const int odd = col >> 4;                          // this lane's own taps are 2j + odd
  const int s = quarter * 16 + (col & 15), sy = s >> 3, sx = s & 7;
  float bias[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bias[j] = a.bias != nullptr ? a.bias[n0 + j * 32 + col] : 0.f;
  const float sc = a.scale;
  const size_t HW8 = (size_t)H * W * 64;
  // The two lanes of a pair split the ROWS: of every row pair (r, r + 1) the lane with the even taps finishes row r, its partner
  // row r + 1 — one exchange per tile and row pair serves both (each lane sends the logits of the row the other finishes and
  // receives those of its own), and each lane runs the softmax (9 expf) for 8 of the wave tile's 16 rows instead of all 16.
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const int rr = r + odd;                          // the row this lane finishes
    const int row = wm0 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
    const unsigned p = (unsigned)m0 + (unsigned)row;
    float own[NT], oth[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float l0 = mask_logit(acc[0][j][r], bias[j], sc), l1 = mask_logit(acc[0][j][r + 1], bias[j], sc);
      own[j] = odd ? l1 : l0;                        // tap 2j + odd of my row
      oth[j] = __shfl_xor(odd ? l0 : l1, 16, 64);    // send the partner's row, receive tap 2j + (1 - odd) of mine
    }
    float m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int j = k >> 1;
      m[k] = ((k & 1) == odd) ? own[j] : oth[j];     // (tap 9 — tile 4 of the odd lanes — is the zero pad row: never read)
    }
    float ox, oy;
    mask_upsample_combine(m, nf + row * 9, ox, oy);
    if ((long long)p < a.M) {
      const unsigned prow = fastdiv_u32(p, a.wo_mul, a.wo_sh);
      const unsigned bimg = fastdiv_u32(prow, a.ho_mul, a.ho_sh);
      const unsigned x = p - prow * W, y = prow - bimg * H;
      const size_t o = (size_t)(8 * y + (unsigned)sy) * (8 * W) + 8 * x + (unsigned)sx;
      a.mu_out[((size_t)bimg * 2 + 0) * HW8 + o] = ox;
      a.mu_out[((size_t)bimg * 2 + 1) * HW8 + o] = oy;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Variant 3 (the tuned path): three LDS stages per wave group, global prefetch distance 2, register
// double-buffered MFMA fragments, one s_barrier per K-step.
//
//   step j:  regs(j+2) -> LDS stage (j+2)%3      [loads issued a full step earlier: no vmcnt stall]
//            issue buffer loads for step j+3 -> regs
//            MFMAs on stage j%3; the fragments of sub-step kk+1 are read while sub-step kk is on the
//            matrix pipe, and the first fragments of step j+1 are read (from stage (j+1)%3, stored
//            during step j-1, a barrier ago) before the barrier — so after the barrier the wave goes
//            straight to MFMAs: no LDS or global latency is exposed at the step boundary.
//
// GROUPS = 2 adds in-block split-K: two 4-wave groups (K-steps g, g+2, ...) with private LDS stages
// share the barriers; each SIMD then hosts two waves whose address arithmetic / LDS traffic overlaps
// the other's MFMAs — the regime of the RAFT update block at batch 1, where M = 7040 pixels gives
// fewer 32x32 output tiles than the chip has SIMDs.  The groups swap accumulator halves through LDS at
// the end (deterministic 2-term sum) and each finishes the epilogue for 8 of the 16 registers.
// -------------------------------------------------------------------------------------------------
template <int MT, int NT>
__device__ __forceinline__ void mma_frags(f32x16 (&acc)[MT][NT], const Frags<MT, NT>& f) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.fa[mt][s], f.fb[nt][s], acc[mt][nt], 0, 0, 0);
}

// Hand-placed instruction stream for one K-step of variant 3.  A wave issues in order, so anything that does
// not sit BETWEEN two MFMAs in program order runs while the matrix pipe idles; each fp32 32x32x2 MFMA covers
// 64 issue cycles, enough for one small "filler" (a ds_write, a buffer load with its address arithmetic, a
// pair of fragment reads).  sched_barrier(0) after every item pins the order:
//   MFMA, [fragment reads of the next sub-step, right after the first MFMA of this one], filler, MFMA, ...
// Fillers in order: LDS stores of step j+2 (loaded a full step ago), scalar setup, the buffer loads of
// step j+3.  All indices are template constants so every register array stays in registers.
template <int BM, int BN, int MT, int NT, int ABL, int LD, int Q>
__device__ __forceinline__ void v3_slot(f32x16 (&acc)[MT][NT], Frags<MT, NT>& f0, Frags<MT, NT>& f1,
                                        Stager<BM, BN, LD>& st, const float* cA, const float* cB, const float* nA,
                                        const float* nB, float* s_fill, bool live, const int (&ko)[4]) {
  constexpr int N0 = 4 * MT * NT;
  constexpr int kk = Q / N0, i = Q % N0;
  constexpr int sidx = i / (MT * NT), mt = (i / NT) % MT, nt = i % NT;
  constexpr int A_PT = Stager<BM, BN, LD>::A_PT, B_PT = Stager<BM, BN, LD>::B_PT, NW = A_PT + B_PT;
  static_assert(N0 + NW <= 4 * N0, "not enough MFMA slots for the fillers");
  // ABL (timing ablations, results are garbage): 1 no buffer loads, 2 + no LDS stores, 3 + no fragment reads, 4 no MFMAs
  if constexpr (ABL != 4) {
    if constexpr (kk & 1) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.fa[mt][sidx], f1.fb[nt][sidx], acc[mt][nt], 0, 0, 0);
    else                  acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.fa[mt][sidx], f0.fb[nt][sidx], acc[mt][nt], 0, 0, 0);
  } else {
    if constexpr (kk & 1) asm volatile("" :: "v"(f1.fa[mt][sidx]), "v"(f1.fb[nt][sidx]));
    else                  asm volatile("" :: "v"(f0.fa[mt][sidx]), "v"(f0.fb[nt][sidx]));
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (i == 0 && (ABL < 3 || ABL == 4)) {  // (ABL 3,5,6: MFMA-only skeletons)   // next sub-step's fragments, >= 3 MFMAs (192 cycles) ahead of their first use
    if constexpr (kk == 0) frag_read<MT, NT, LD>(f1, cA, cB, ko, 1);
    else if constexpr (kk == 1) frag_read<MT, NT, LD>(f0, cA, cB, ko, 2);
    else if constexpr (kk == 2) frag_read<MT, NT, LD>(f1, cA, cB, ko, 3);
    else frag_read<MT, NT, LD>(f0, nA, nB, ko, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // Filler q (from the second sub-step on): store staged register q into the free stage and immediately reload
  // that register with its share of step j+3 — every load then has one full K-step (~1000+ cycles) to land
  // before its ds_write in the next step, wherever in the step it sits.
  constexpr int S0 = N0;                       // first filler slot
  if constexpr (Q == S0 - 1) st.load_setup(live);
  else if constexpr (Q >= S0 && Q < S0 + NW) {
    constexpr int q = Q - S0;
    if constexpr (ABL < 2 || ABL == 4) st.template store_piece<q>(s_fill);
    if constexpr (ABL < 1 || ABL == 4) {
      if constexpr (q < A_PT) st.template load_a<q>();
      else st.template load_b<q - A_PT>();
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int BM, int BN, int MT, int NT, int ABL, int LD, int... Qs>
__device__ __forceinline__ void v3_step(f32x16 (&acc)[MT][NT], Frags<MT, NT>& f0, Frags<MT, NT>& f1,
                                        Stager<BM, BN, LD>& st, const float* cA, const float* cB, const float* nA,
                                        const float* nB, float* s_fill, bool live, const int (&ko)[4],
                                        std::integer_sequence<int, Qs...>) {
  (v3_slot<BM, BN, MT, NT, ABL, LD, Qs>(acc, f0, f1, st, cA, cB, nA, nB, s_fill, live, ko), ...);
}

// (the kernel's body as a device function of (arguments, tile id, tiles in this problem's grid, batch element): conv_gemm_v3_kernel
//  runs it on its own grid, conv_gemm_v3_group_kernel on one problem's share of a grouped grid)
template <int BM, int BN, int WM, int WN, int EPI, int GROUPS, int ABL = 0, int LD = LDS_LD>
__device__ __forceinline__ void conv_gemm_v3_body(const GemmArgs& a, const int bid, const int nblk, const long long batch) {
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per group");
  static_assert(BK == 32, "fragment schedule below is written for 4 sub-steps");
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int STAGE = (BM + BN) * LD;      // floats per stage: A rows then B rows
  constexpr int GROUP_LDS = 3 * STAGE;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int grp = GROUPS == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);  // wave-uniform
  const int gtid = threadIdx.x & 255;
  float* base = smem + grp * GROUP_LDS;

  const int lane = gtid & 63;
  const int wid = gtid >> 6;
  const int wm0 = (wid / WAVES_N) * WM;
  const int wn0 = (wid % WAVES_N) * WN;

  int tile_m, tile_n;
  tile_of(bid, nblk / a.tiles_n, a.tiles_n, a.supertile, tile_m, tile_n);
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  Stager<BM, BN, LD> st(a, m0, n0, gtid, batch);
  const int S = st.total_steps();
  const int my_steps = (S - grp + GROUPS - 1) / GROUPS;
  const int max_steps = (S + GROUPS - 1) / GROUPS;   // same trip count for every wave (barriers)
  if (GROUPS > 1 && grp) st.advance();

  f32x16 acc[MT][NT];
  zero_acc<MT, NT>(acc);

  float* s_cur = base;               // stage holding step j
  float* s_nxt = base + STAGE;       // step j+1
  float* s_fill = base + 2 * STAGE;  // receives step j+2

  // prologue: steps 0 and 1 into LDS, step 2 into registers (dead steps load zeros)
  st.load(0 < my_steps);
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) st.advance();
  st.store(s_cur, s_cur + BM * LD);
  st.load(1 < my_steps);
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) st.advance();
  st.store(s_nxt, s_nxt + BM * LD);
  st.load(2 < my_steps);
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) st.advance();
  __syncthreads();

  const int foff_a = (wm0 + (lane & 31)) * LD;
  const int foff_b = BM * LD + (wn0 + (lane & 31)) * LD;
  int ko[4];
  frag_offsets<LD>(ko, lane);
  Frags<MT, NT> f0, f1;
  frag_read<MT, NT, LD>(f0, s_cur + foff_a, s_cur + foff_b, ko, 0);

  for (int j = 0; j < max_steps; ++j) {
    v3_step<BM, BN, MT, NT, ABL, LD>(acc, f0, f1, st, s_cur + foff_a, s_cur + foff_b, s_nxt + foff_a, s_nxt + foff_b, s_fill,
                            j + 3 < my_steps, ko, std::make_integer_sequence<int, 16 * MT * NT>{});
    if constexpr (ABL != 6) {
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) st.advance();
    }
    float* t = s_cur; s_cur = s_nxt; s_nxt = s_fill; s_fill = t;
    if constexpr (ABL != 5) __syncthreads();
  }

  if constexpr (GROUPS == 1) {
    epilogue_lds<MT, NT, EPI, true>(a, acc, m0 + wm0, n0 + wn0, lane, batch, smem, wid);   // behind the loop's final barrier
  } else {
    // exchange accumulator halves: group g finishes registers [8g, 8g+8).
    // red[group][wave][tile][reg8][lane]: lane-contiguous, conflict-free.  (All stage reads are
    // behind the loop's final barrier, so the staging LDS can be reused.)
    float* red = smem;
    constexpr int PER_WAVE = MT * NT * 8 * 64;
    {
      float* mine = red + ((grp * 4 + wid) * PER_WAVE) + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float v = grp ? acc[mt][nt][r] : acc[mt][nt][8 + r];   // the half the partner finishes
            mine[((mt * NT + nt) * 8 + r) * 64] = v;
          }
    }
    __syncthreads();
    {
      const float* theirs = red + (((1 - grp) * 4 + wid) * PER_WAVE) + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float v = theirs[((mt * NT + nt) * 8 + r) * 64];
            if (grp) acc[mt][nt][8 + r] += v; else acc[mt][nt][r] += v;
          }
    }
    if (grp) epilogue<MT, NT, EPI, 8, 16>(a, acc, m0 + wm0, n0 + wn0, lane, batch);
    else     epilogue<MT, NT, EPI, 0, 8>(a, acc, m0 + wm0, n0 + wn0, lane, batch);
  }
}

template <int BM, int BN, int WM, int WN, int EPI, int GROUPS, int ABL = 0, int LD = LDS_LD>
__global__ __launch_bounds__(256 * GROUPS) void conv_gemm_v3_kernel(const GemmArgs a) {
  conv_gemm_v3_body<BM, BN, WM, WN, EPI, GROUPS, ABL, LD>(a, pfk_xcd_remap(blockIdx.x, gridDim.x), (int)gridDim.x, blockIdx.y);
}

// -------------------------------------------------------------------------------------------------
// Grouped launch (round 6): up to PFK_CONV_GROUP_MAX INDEPENDENT convolutions (LINEAR epilogue, 64x64 tiles) in ONE grid — block b
// belongs to problem k with first[k] <= b < first[k + 1] and runs the tile kernel's body on that problem's arguments.  At batch 1
// (7040 pixels) the update block's launches have 110..990 tiles for 256 CUs and run for 8..25 us each, mostly the serial latency
// of ONE tile's K loop plus launch / drain: convc1 (440 tiles x 11 K-steps), convf2 (110 x 36) and the previous iteration's mask
// conv2 (990 x 8) do not depend on each other (update.py:105-108: `cor` and `flo` only meet in `conv`; :152: the mask head reads
// `net` only), so one grid of 1540 tiles replaces three launches and their three tails.  The runtime does not co-run kernels of one
// stream and forked streams / graph branches were measured neutral (history, rounds 3-4): a single grid is the form that co-runs.
// Every tile is computed exactly as its own launch would: same K order, same bits.
// -------------------------------------------------------------------------------------------------
struct GemmGroupArgs {
  GemmArgs p[PFK_CONV_GROUP_MAX];
  int first[PFK_CONV_GROUP_MAX + 1];
  int n;
};

template <int LD, int BN = 64, int WN = 32>
__global__ __launch_bounds__(256) void conv_gemm_v3_group_kernel(const GemmGroupArgs g) {
  const int b = blockIdx.x;
  int k = 0;
#pragma unroll
  for (int i = 1; i < PFK_CONV_GROUP_MAX; ++i)
    if (i < g.n && b >= g.first[i]) k = i;
  const int f0 = g.first[k], nblk = g.first[k + 1] - f0;
  conv_gemm_v3_body<64, BN, 32, WN, PFK_EPI_LINEAR, 1, 0, LD>(g.p[k], pfk_xcd_remap(b - f0, nblk), nblk, 0);
}

// -------------------------------------------------------------------------------------------------
// Variant 4: stream-K on 64x64 tiles.  The update block at batch 1 has M = 7040 pixels: 220..880 tiles of
// 64x64 for 256 CUs, so a tile-per-block grid leaves the busiest CU with up to 1.6x the average work.
// Here the (tile, K-step) space is cut into gridDim.x equal contiguous ranges (two resident blocks per CU);
// a block walks its range from the top tile down, running the variant-3 pipeline on each tile segment:
//   * a segment that does not contain the tile's LAST K-step (only possible for the block's top segment) is
//     a contribution: the 64x64 partial goes to the block's workspace slot, released with a flag;
//   * the segment containing the last K-step owns the tile: it adds the partials of the lower-numbered
//     blocks that hold the rest of the tile (fixed order => deterministic sum) and runs the epilogue.
// Owners wait only on LOWER block ids, which produced their contribution as the FIRST thing they did, so the
// wait is short and cannot deadlock whatever the residency.  Cross-workgroup visibility follows the
// agent-scope release/acquire recipe of cdna_hip_programming.md (Guideline 16).  Flags: the caller zero-fills the workspace
// once (pfk.h); a contribution's flag is set by its producer and cleared by its single consumer, so every launch finds and
// leaves the region zeroed (no memset node per launch: 5.6 us each at batch 1).  The spin is bounded; after a timed-out
// fix-up (fault word != 0) the caller must zero the workspace again before reusing it.
// -------------------------------------------------------------------------------------------------
constexpr int SK_MAX_BLOCKS = 768;    // three resident blocks per CU on the swizzled 48 KB layout
constexpr int SK_FAULT_SLOT = 1024;   // u32 index in the flag region (flags use [0, 768)): sticky count of timed-out fix-ups
constexpr size_t SK_WS_BYTES = (size_t)SK_MAX_BLOCKS * (64 * 64 * 4 + 64);   // partials + flags (flags after the partials)

// LD = LDS row stride (36: padded, 55 KB per block; 32: XOR-swizzled, 48 KB), BPC = resident blocks per CU the launch counts on.
// a.sk_groups == 8: the tile list is cut into 8 contiguous ranges at TILE boundaries, one per XCD (blockIdx.x % 8 — the hardware's
// round-robin), and the stream-K split runs inside each range over the blocks of that XCD in dispatch order: neighbouring unit
// ranges then share their A rows / weight panels through ONE L2, and a contribution still only ever flows from a lower to a higher
// dispatch index (blocks x, x + 8, x + 16, ...), which is what makes the fix-up wait deadlock-free whatever the residency.
template <int EPI, int LD, int BPC>
__global__ __launch_bounds__(256, BPC) void conv_gemm_sk_kernel(const GemmArgs a) {
  constexpr int BM = 64, BN = 64, MT = 1, NT = 1;
  constexpr int STAGE = (BM + BN) * LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm0 = (wid >> 1) * 32, wn0 = (wid & 1) * 32;
  const int foff_a = (wm0 + (lane & 31)) * LD;
  const int foff_b = BM * LD + (wn0 + (lane & 31)) * LD;
  int ko[4];
  frag_offsets<LD>(ko, lane);

  __shared__ int s_lost;   // set by thread 0 when a partner's contribution timed out (read after the next barrier)
  if (threadIdx.x == 0) s_lost = 0;
  // NO free XCD remap here: a block may only wait on blocks with a LOWER dispatch index (those are running or done whatever
  // the residency), so inside a group the logical order of the unit ranges is the dispatch order.
  // 32-bit unit arithmetic: the host only launches this kernel when units x blocks < 2^31
  const int X = a.sk_groups;
  const int d = blockIdx.x;
  const int grp = d % X, rank = d / X;
  const int Gx = (int)gridDim.x / X;                  // host: gridDim.x % X == 0
  const int S = a.sk_steps;
  const int T = (int)a.sk_tiles;
  const int tb = (int)((long long)T * grp / X), te = (int)((long long)T * (grp + 1) / X);
  const int ubase = tb * S, Ux = (te - tb) * S;
  const int u0 = ubase + (int)((unsigned)(rank * Ux) / (unsigned)Gx), u1 = ubase + (int)((unsigned)((rank + 1) * Ux) / (unsigned)Gx);

  int hi = u1;
  while (hi > u0) {
    const int tile = (hi - 1) / S;
    const int tbeg = tile * S;
    const int lo = u0 > tbeg ? u0 : tbeg;
    const int s0 = lo - tbeg, s1 = hi - tbeg, nsteps = s1 - s0;
    int tile_n = tile % a.tiles_n;
    if (a.sk_split_tiles > 0) tile_n = (tile_n & 1) ? a.sk_split_tiles + (tile_n >> 1) : (tile_n >> 1);   // two output groups interleaved
    if (a.active_tiles_n > 0 && tile_n >= a.active_tiles_n) {
      // a column tile nobody asked for (pfk_conv_desc.cout_active): every block that holds a piece of it skips it — no partial, no
      // flag, no epilogue; the split points of the tiles that ARE computed stay those of the full launch (same bits)
      hi = lo;
      continue;
    }
    const long long m0 = (long long)(tile / a.tiles_n) * BM;
    const int n0 = tile_n * BN;

    Stager<BM, BN, LD> st(a, m0, n0, tid, 0);
    if (s0) st.seek(s0);
    f32x16 acc[MT][NT];
    zero_acc<MT, NT>(acc);

    float* s_cur = smem;
    float* s_nxt = smem + STAGE;
    float* s_fill = smem + 2 * STAGE;
    st.load(0 < nsteps); st.advance(); st.store(s_cur, s_cur + BM * LD);
    st.load(1 < nsteps); st.advance(); st.store(s_nxt, s_nxt + BM * LD);
    st.load(2 < nsteps); st.advance();
    __syncthreads();
    Frags<MT, NT> f0, f1;
    frag_read<MT, NT, LD>(f0, s_cur + foff_a, s_cur + foff_b, ko, 0);
    for (int j = 0; j < nsteps; ++j) {
      v3_step<BM, BN, MT, NT, 0, LD>(acc, f0, f1, st, s_cur + foff_a, s_cur + foff_b, s_nxt + foff_a, s_nxt + foff_b, s_fill,
                                 j + 3 < nsteps, ko, std::make_integer_sequence<int, 16 * MT * NT>{});
      st.advance();
      float* t = s_cur; s_cur = s_nxt; s_nxt = s_fill; s_fill = t;
      __syncthreads();
    }

    if (s1 < S) {
      // contribution: partial -> workspace slot of this block, then publish
      float* mine = a.sk_ws + ((long long)d * 4 + wid) * (16 * 64) + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][0][r];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.sk_flags + d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (s0 > 0) {
        // owner of a split tile: add the partials of the group's blocks rank-1, rank-2, ... that cover [tbeg, lo)
        for (int k = rank - 1; k >= 0; --k) {
          const int slot = grp + k * X;                // that block's dispatch index
          if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(a.sk_flags + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(4);
              if (++spins > (1u << 24)) {
                // bounded spin (no hang under preemption / a debugger) — but a contribution that never arrived must not pass
                // for a result: count it in the workspace's sticky fault word (pfk_conv_workspace_fault_offset) and poison
                // this tile with NaN, which every consumer downstream propagates into the flow.
                atomicAdd(a.sk_flags + SK_FAULT_SLOT, 1u);
                s_lost = 1;
                break;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // every contribution has exactly one consumer (this owner): hand the flag back as 0, so the flag region is all-zero
            // again when the kernel ends and the next launch needs no memset in front of it
            __hip_atomic_store(a.sk_flags + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          __syncthreads();
          const float* theirs = a.sk_ws + ((long long)slot * 4 + wid) * (16 * 64) + lane;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][0][r] += theirs[r * 64];
          if (ubase + (int)((unsigned)(k * Ux) / (unsigned)Gx) <= tbeg) break;   // block k's range starts at or before the tile: it was the last contributor
        }
        if (s_lost) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][0][r] = __builtin_nanf("");
        }
      }
      epilogue_lds<MT, NT, EPI>(a, acc, m0 + wm0, n0 + wn0, lane, 0, smem, wid);
    }
    hi = lo;
    __syncthreads();   // LDS stages are reused by the next segment
  }
}

// -------------------------------------------------------------------------------------------------
// Variant 5: PERSISTENT, cross-tile PIPELINED stream-K ("pp").  Same 64x64 tile, 3-stage LDS ring and hand-placed MFMA / filler
// stream as variants 3 / 4, but the K pipeline never drains between tiles: a block owns a contiguous range of (tile, K-step)
// units — the stream-K split of variant 4, balanced to +-1 K-step over all blocks, per-XCD tile groups — and the stager simply
// keeps running three K-steps ahead of the MFMAs ACROSS tile (segment) boundaries.  What variants 3 / 4 pay per tile and this
// one pays once per block:
//   * the pipeline prologue (two dependent global round trips before the first MFMA) — the next tile's first K-steps are already
//     in LDS / in flight while the current tile finishes;
//   * workgroup launch + Stager construction (six 64-bit divisions per thread) — re-targeting uses host-made multipliers;
//   * the dispatch tail: with tiles handed out whole, the last round of a 3-blocks-per-CU grid runs 1-2 blocks per CU.
// The epilogue of a finished tile borrows the LDS stage the pipeline has just freed (its data was consumed by the tile's last
// K-step; it is only overwritten by the NEXT step's fillers, behind one extra barrier per tile), so the LDS footprint — and the
// three (LD = 32) / two (LD = 36) resident blocks per CU — is unchanged; the co-resident blocks' MFMAs cover the epilogue.
// Partial tiles: identical protocol to variant 4 (contribution from the block's FIRST segment, fixed-order fix-up by the owner,
// self-clearing agent-scope flags, bounded spin + sticky fault word).  pp_whole = 1 cuts the unit ranges at tile boundaries
// instead (no workspace needed: short-K launches such as the correlation volume, or callers without a workspace).
// -------------------------------------------------------------------------------------------------
// A block's work list.  Phase A: its share [u0, u1) of the stream-K REMAINDER (unit space of the group's last R tiles, R < blocks of
// the group), top tile first so that a contribution is published as early as possible.  Phase B: whole tiles, one per round,
// round-robin over the group's blocks — tile = dp_base + k * dp_stride — so that at any time the blocks of an XCD work on
// CONSECUTIVE tiles: the column tiles of a row panel (and a supertile's neighbours) run side by side and share their A rows
// through the XCD's L2.  (Contiguous per-block tile ranges — what variant 4 does for its few-hundred-tile grids — make every block
// re-stream its own A panel once per column tile: PMC on fh|mask conv1 at batch 8 read 616 MB from HBM instead of 60 MB, L2 hit
// rate 0.83 instead of 0.96, and the kernel lost everything the missing prologues had gained.)
struct SegIter {
  int tile, tbeg, hi, u0, S;          // phase A (tile indices relative to sk_tile0)
  int sk_tile0;
  int dp_k, dp_rounds, dp_base, dp_stride;
  __device__ __forceinline__ void init(int u0_, int u1_, int S_, int sk_tile0_, int dp_base_, int dp_stride_, int dp_rounds_) {
    u0 = u0_; hi = u1_; S = S_; sk_tile0 = sk_tile0_;
    tile = u1_ > u0_ ? (u1_ - 1) / S_ : 0;
    tbeg = tile * S_;
    dp_k = 0; dp_rounds = dp_rounds_; dp_base = dp_base_; dp_stride = dp_stride_;
  }
  // next segment: absolute tile `t`, K-steps [s0, s1); `rel` = the tile's index inside the remainder (phase A) or -1
  __device__ __forceinline__ bool next(int& t, int& s0, int& s1, int& rel) {
    if (hi > u0) {
      const int lo = u0 > tbeg ? u0 : tbeg;
      rel = tile; t = sk_tile0 + tile; s0 = lo - tbeg; s1 = hi - tbeg;
      hi = lo; --tile; tbeg -= S;
      return true;
    }
    if (dp_k < dp_rounds) {
      rel = -1; t = dp_base + dp_k * dp_stride; s0 = 0; s1 = S;
      ++dp_k;
      return true;
    }
    return false;
  }
};

__device__ __forceinline__ void pp_decode(const GemmArgs& a, int tile, long long& m0, int& n0, long long& batch) {
  int local = tile, b = 0;
  if (a.pp_batches > 1) { b = tile / a.pp_tiles_pb; local = tile - b * a.pp_tiles_pb; }
  int tm, tn;
  if (a.supertile > 0) {
    tile_of(local, a.pp_tiles_m, a.tiles_n, a.supertile, tm, tn);
  } else {
    tm = (int)fastdiv_u32((unsigned)local, a.tn_mul, a.tn_sh);
    tn = local - tm * a.tiles_n;
  }
  m0 = (long long)tm * 64; n0 = tn * 64; batch = b;
}

// the stager moves ONE step forward in the block's flattened (segment, K-step) order
template <int LD>
__device__ __forceinline__ void pp_advance(const GemmArgs& a, Stager<64, 64, LD>& st, SegIter& ps, int& left) {
  if (--left > 0) { st.advance(); return; }
  int t, s0, s1, rel;
  if (!ps.next(t, s0, s1, rel)) return;     // past the block's last step: the remaining loads are dead (live = false)
  long long m0, batch; int n0;
  pp_decode(a, t, m0, n0, batch);
  st.retarget(a, m0, n0, batch);
  if (s0) st.seek_args(a, batch, s0); else st.rewind();
  left = s1 - s0;
}

template <int EPI, int LD, int BPC, int ABL = 0>
__global__ __launch_bounds__(256, BPC) void conv_gemm_pp_kernel(const GemmArgs a) {
  constexpr int BM = 64, BN = 64, MT = 1, NT = 1;
  constexpr int STAGE = (BM + BN) * LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm0 = (wid >> 1) * 32, wn0 = (wid & 1) * 32;
  const int foff_a = (wm0 + (lane & 31)) * LD;
  const int foff_b = BM * LD + (wn0 + (lane & 31)) * LD;
  int ko[4];
  frag_offsets<LD>(ko, lane);

  __shared__ int s_lost;   // set by thread 0 when a partner's contribution timed out (read after the next barrier)
  if (threadIdx.x == 0) s_lost = 0;
  // unit range of this block: group = XCD (blockIdx.x % X, the hardware's round-robin), rank = dispatch order inside the group;
  // a contribution only ever flows from a lower to a higher rank (variant 4's deadlock-freedom argument)
  const int X = a.sk_groups;
  const int d = blockIdx.x;
  const int grp = d % X, rank = d / X;
  const int Gx = (int)gridDim.x / X;                  // host: gridDim.x % X == 0
  const int S = a.sk_steps;
  const int T = (int)a.sk_tiles;
  const int tb = (int)((long long)T * grp / X), te = (int)((long long)T * (grp + 1) / X);
  const int Tx = te - tb;
  int full = Tx / Gx;                          // whole-tile rounds every block of the group takes part in
  const int R = Tx - full * Gx;                // remainder tiles (< Gx)
  int u0 = 0, u1 = 0, Ur = 0;
  if (a.pp_whole) {
    if (rank < R) ++full;                      // one more whole tile for the first R blocks, nothing is split
  } else {                                     // 32-bit unit arithmetic: the host only launches this split when units x blocks < 2^31
    Ur = R * S;
    u0 = (int)((unsigned)(rank * Ur) / (unsigned)Gx);
    u1 = (int)((unsigned)((rank + 1) * Ur) / (unsigned)Gx);
  }
  const int total = (u1 - u0) + full * S;
  if (total <= 0) return;   // block-uniform, before any barrier

  SegIter cs, ps;
  const int sk_tile0 = tb + (Tx / Gx) * Gx;    // first remainder tile
  cs.init(u0, u1, S, sk_tile0, tb + rank, Gx, full);
  ps.init(u0, u1, S, sk_tile0, tb + rank, Gx, full);
  Stager<BM, BN, LD> st(a, tid, typename Stager<BM, BN, LD>::NoTile{});
  int p_left;
  {
    int t, s0, s1, rel;
    ps.next(t, s0, s1, rel);
    long long m0, batch; int n0;
    pp_decode(a, t, m0, n0, batch);
    st.retarget(a, m0, n0, batch);
    if (s0) st.seek_args(a, batch, s0); else st.rewind();
    p_left = s1 - s0;
  }

  float* s_cur = smem;
  float* s_nxt = smem + STAGE;
  float* s_fill = smem + 2 * STAGE;
  // the ONLY pipeline prologue of the block: flattened steps 0 and 1 into LDS, step 2 into registers
  st.load(0 < total); pp_advance<LD>(a, st, ps, p_left); st.store(s_cur, s_cur + BM * LD);
  st.load(1 < total); pp_advance<LD>(a, st, ps, p_left); st.store(s_nxt, s_nxt + BM * LD);
  st.load(2 < total); pp_advance<LD>(a, st, ps, p_left);
  __syncthreads();
  Frags<MT, NT> f0, f1;
  frag_read<MT, NT, LD>(f0, s_cur + foff_a, s_cur + foff_b, ko, 0);

  int g = 0;            // flattened step index of the MFMA side
  int tile, s0, s1, rel;
  while (cs.next(tile, s0, s1, rel)) {
    long long m0, batch; int n0;
    pp_decode(a, tile, m0, n0, batch);
    const int nsteps = s1 - s0;
    f32x16 acc[MT][NT];
    zero_acc<MT, NT>(acc);
    for (int j = 0; j < nsteps; ++j, ++g) {
      v3_step<BM, BN, MT, NT, ABL, LD>(acc, f0, f1, st, s_cur + foff_a, s_cur + foff_b, s_nxt + foff_a, s_nxt + foff_b, s_fill,
                                       g + 3 < total, ko, std::make_integer_sequence<int, 16 * MT * NT>{});
      pp_advance<LD>(a, st, ps, p_left);
      float* t = s_cur; s_cur = s_nxt; s_nxt = s_fill; s_fill = t;
      __syncthreads();
    }
    // Behind the segment's last barrier: s_fill is the stage the last K-step consumed — free until the next step's fillers —
    // and f0 already holds the next segment's first fragments (read from s_cur before the barrier).

    if (s1 < S) {
      // contribution: partial -> workspace slot of this block, then publish
      float* mine = a.sk_ws + ((long long)d * 4 + wid) * (16 * 64) + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[r * 64] = acc[0][0][r];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.sk_flags + d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (s0 > 0) {
        // owner of a split tile: add the partials of the group's blocks rank-1, rank-2, ... that cover [tbeg, tbeg + s0)
        // (remainder unit space: tile `rel` of the group's remainder starts at unit rel * S, block k's share at k * Ur / Gx)
        const int tbeg = rel * S;
        for (int k = rank - 1; k >= 0; --k) {
          const int ku0 = (int)((unsigned)(k * Ur) / (unsigned)Gx), ku1 = (int)((unsigned)((k + 1) * Ur) / (unsigned)Gx);
          if (ku1 <= ku0) continue;                    // an empty share (fewer remainder units than blocks): nothing was published
          const int slot = grp + k * X;                // that block's dispatch index
          if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(a.sk_flags + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(4);
              if (++spins > (1u << 24)) {      // see variant 4: count it in the sticky fault word and poison the tile
                atomicAdd(a.sk_flags + SK_FAULT_SLOT, 1u);
                s_lost = 1;
                break;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(a.sk_flags + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // single consumer: hand it back as 0
          }
          __syncthreads();
          const float* theirs = a.sk_ws + ((long long)slot * 4 + wid) * (16 * 64) + lane;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][0][r] += theirs[r * 64];
          if (ku0 <= tbeg) break;   // block k's share starts at or before the tile: last contributor
        }
        if (s_lost) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][0][r] = __builtin_nanf("");
        }
      }
      epilogue_lds<MT, NT, EPI>(a, acc, m0 + wm0, n0 + wn0, lane, batch, s_fill, wid);
    }
    // Drain the vector-memory counter HERE, once per tile (the epilogue's stores, the fix-up's loads, the prefetch issued three
    // steps ago): with stores of unknown count pending at the K loop's entry hipcc waited `vmcnt(0)` inside every K-step instead
    // of the counted `vmcnt(3)` the tile kernel gets.  (A real S_WAITCNT, which the compiler's counter model sees.)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) expcnt(7) lgkmcnt(15)
    if (g < total) __syncthreads();   // the epilogue's scratch (s_fill) is the target of the next step's LDS fillers
  }
}


template <int EPI, int LD, int BPC, int ABL = 0>
int launch_pp_one(const GemmArgs& g, unsigned G, hipStream_t st) {
  constexpr size_t smem = 3 * 128 * LD * sizeof(float);
  auto kern = conv_gemm_pp_kernel<EPI, LD, BPC, ABL>;
  static pfk_device_once attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL(kern, dim3(G), dim3(256), smem, st, g);
  return pfk_launch_status();
}

template <int LD, int BPC>
int launch_pp_epi(const GemmArgs& g, int epi, unsigned G, hipStream_t st) {
  switch (epi) {
    case PFK_EPI_LINEAR: return launch_pp_one<PFK_EPI_LINEAR, LD, BPC>(g, G, st);
    case PFK_EPI_GRU_ZR: return launch_pp_one<PFK_EPI_GRU_ZR, LD, BPC>(g, G, st);
    case PFK_EPI_GRU_Q:  return launch_pp_one<PFK_EPI_GRU_Q, LD, BPC>(g, G, st);
    default: return PFK_ERR_BAD_ARG;
  }
}

// variant bits: 1 = swizzled 48 KB LDS layout, 2 = per-XCD tile groups, 16 = whole tiles only; blocks per CU = 1 + ((variant >> 2) & 3)
// (default: as many as the layout allows — 3 swizzled, 2 padded — when those bits are 0)
int launch_pp(const GemmArgs& a, int epi, int batches, hipStream_t st, int variant, int abl = 0) {
  GemmArgs g = a;
  if (a.M >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  g.tiles_n = (a.b_rows + 63) / 64;
  g.pp_tiles_m = (int)((a.M + 63) / 64);
  const long long tiles_pb = (long long)g.pp_tiles_m * g.tiles_n;
  const long long tiles = tiles_pb * batches;
  if (tiles <= 0 || tiles_pb > 0x7fffffffLL || tiles * a.sk_steps >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  g.pp_tiles_pb = (int)tiles_pb;
  g.pp_batches = batches;
  g.sk_tiles = tiles;
  g.supertile = (g.pp_tiles_m >= 32 && g.tiles_n >= 32) ? 16 : 0;
  fastdiv_make((unsigned)a.Wo, g.wo_mul, g.wo_sh);
  fastdiv_make((unsigned)a.Ho, g.ho_mul, g.ho_sh);
  fastdiv_make((unsigned)g.tiles_n, g.tn_mul, g.tn_sh);
  const bool swz = variant & 1;
  int bpc = (variant >> 2) & 3;
  bpc = bpc ? bpc : (swz ? 3 : 2);
  if (bpc > 3 || (bpc == 3 && !swz)) return PFK_ERR_BAD_ARG;     // three padded blocks (165 KB) do not fit a CU's LDS
  const long long U = tiles * a.sk_steps;
  long long G = 256LL * bpc;
  g.pp_whole = ((variant & 16) || a.sk_ws == nullptr || batches > 1 || a.sk_steps < 12) ? 1 : 0;
  if (g.pp_whole) { if (G > tiles) G = tiles; }
  else if (U / G < 6) G = U / 6 > 0 ? U / 6 : 1;   // keep segments long enough to amortise a fix-up
  if (!g.pp_whole && U * (G + 1) >= 0x7fffffffLL) g.pp_whole = 1;   // the split's unit arithmetic is 32-bit
  g.sk_groups = 1;
  if ((variant & 2) && tiles >= 64 && G >= 64) { g.sk_groups = 8; G -= G % 8; }
  if (abl) {   // timing ablations (results are garbage; scripts/conv_bench.py): linear epilogue, swizzled x3 or padded x2 only
#ifdef PFK_BENCH_VARIANTS     // (not in the shipped library: ptlflow_amd/_build.py, PFK_BENCH_VARIANTS=1)
    if (epi != PFK_EPI_LINEAR || !((swz && bpc == 3) || (!swz && bpc == 2))) return PFK_ERR_BAD_ARG;
    switch (abl) {
      case 1: return swz ? launch_pp_one<PFK_EPI_LINEAR, LDS_LDX, 3, 1>(g, (unsigned)G, st) : launch_pp_one<PFK_EPI_LINEAR, LDS_LD, 2, 1>(g, (unsigned)G, st);
      case 2: return swz ? launch_pp_one<PFK_EPI_LINEAR, LDS_LDX, 3, 2>(g, (unsigned)G, st) : launch_pp_one<PFK_EPI_LINEAR, LDS_LD, 2, 2>(g, (unsigned)G, st);
      case 3: return swz ? launch_pp_one<PFK_EPI_LINEAR, LDS_LDX, 3, 3>(g, (unsigned)G, st) : launch_pp_one<PFK_EPI_LINEAR, LDS_LD, 2, 3>(g, (unsigned)G, st);
      default: return PFK_ERR_BAD_ARG;
    }
#else
    return PFK_ERR_BAD_ARG;
#endif
  }
  if (bpc == 1) return swz ? launch_pp_epi<LDS_LDX, 1>(g, epi, (unsigned)G, st) : launch_pp_epi<LDS_LD, 1>(g, epi, (unsigned)G, st);
  if (bpc == 2) return swz ? launch_pp_epi<LDS_LDX, 2>(g, epi, (unsigned)G, st) : launch_pp_epi<LDS_LD, 2>(g, epi, (unsigned)G, st);
  return launch_pp_epi<LDS_LDX, 3>(g, epi, (unsigned)G, st);
}

// variant bits: 1 = swizzled 48 KB LDS layout, 2 = per-XCD tile groups, blocks per CU = 1 + (variant >> 2)  (1..3)
int g_sk_variant = 6;     // padded layout, two blocks per CU, per-XCD groups: the best of the sweep (scripts/conv_bench.py cfg 30 + v at
                          // batch 1: z|r conv 69.8 us with one group -> 63.7 us with XCD groups; swizzled / three per CU: 65.6 / 70.3)

template <int EPI, int LD, int BPC>
int launch_sk_one(const GemmArgs& g, unsigned G, hipStream_t st) {
  constexpr size_t smem = 3 * 128 * LD * sizeof(float);
  auto kern = conv_gemm_sk_kernel<EPI, LD, BPC>;
  static pfk_device_once attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL(kern, dim3(G), dim3(256), smem, st, g);
  return pfk_launch_status();
}

template <int LD, int BPC>
int launch_sk_epi(const GemmArgs& g, int epi, unsigned G, hipStream_t st) {
  switch (epi) {
    case PFK_EPI_LINEAR: return launch_sk_one<PFK_EPI_LINEAR, LD, BPC>(g, G, st);
    case PFK_EPI_GRU_ZR: return launch_sk_one<PFK_EPI_GRU_ZR, LD, BPC>(g, G, st);
    case PFK_EPI_GRU_Q:  return launch_sk_one<PFK_EPI_GRU_Q, LD, BPC>(g, G, st);
    default: return PFK_ERR_BAD_ARG;
  }
}

int launch_sk(const GemmArgs& a, int epi, hipStream_t st, int variant) {
  GemmArgs g = a;
  g.tiles_n = (a.b_rows + 63) / 64;
  g.sk_tiles = ((a.M + 63) / 64) * g.tiles_n;
  const bool swz = variant & 1;
  int bpc = 1 + (variant >> 2);
  if (bpc > 3 || (bpc == 3 && !swz)) return PFK_ERR_BAD_ARG;     // three padded blocks (165 KB) do not fit a CU's LDS
  const long long U = g.sk_tiles * g.sk_steps;
  long long G = 256LL * bpc;
  if (U / G < 6) G = U / 6 > 0 ? U / 6 : 1; // keep segments long enough to amortise the pipeline prologue
  if (U * (G + 1) >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;   // the kernel's unit arithmetic is 32-bit
  g.sk_groups = 1;
  if ((variant & 2) && g.sk_tiles >= 64 && G >= 64) { g.sk_groups = 8; G -= G % 8; }
  if (bpc == 2 && !swz) return launch_sk_epi<LDS_LD, 2>(g, epi, (unsigned)G, st);       // what the shipped library's heuristic selects
#ifdef PFK_BENCH_VARIANTS     // the other LDS layouts / residencies of the round-2 schedule sweep (scripts/conv_bench.py cfg 30 + v)
  if (bpc == 1) return swz ? launch_sk_epi<LDS_LDX, 1>(g, epi, (unsigned)G, st) : launch_sk_epi<LDS_LD, 1>(g, epi, (unsigned)G, st);
  if (bpc == 2) return launch_sk_epi<LDS_LDX, 2>(g, epi, (unsigned)G, st);
  return launch_sk_epi<LDS_LDX, 3>(g, epi, (unsigned)G, st);
#else
  return PFK_ERR_BAD_ARG;
#endif
}

// VARIANT: 0 = 4-wave double-buffered pipeline (v1), 1 = v3 one group, 2 = v3 two groups (in-block split-K);
// +10*ABL = timing ablation; +100 = un-padded XOR-swizzled LDS rows (48 KB per 64x64 block: three blocks per CU)
template <int BM, int BN, int WM, int WN, int EPI, int VARIANT>
int launch_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
  if constexpr (VARIANT == 0) {
    constexpr size_t smem = 2 * (BM + BN) * LDS_LD * sizeof(float);
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, EPI>), grid, dim3(256), smem, st, g);
  } else {
    constexpr int G = VARIANT % 10, ABL = (VARIANT / 10) % 10, LD = VARIANT >= 100 ? LDS_LDX : LDS_LD;
    constexpr size_t smem = (size_t)G * 3 * (BM + BN) * LD * sizeof(float);
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = conv_gemm_v3_kernel<BM, BN, WM, WN, EPI, G, ABL, LD>;
    static pfk_device_once attr_once;   // one per template instantiation and device; safe with several host threads
    attr_once.run([&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    hipLaunchKernelGGL(kern, grid, dim3(256 * G), smem, st, g);
  }
  return pfk_launch_status();
}

template <int BM, int BN, int WM, int WN, int VARIANT>
int launch_cfg(const GemmArgs& a, int epi, int batches, hipStream_t st) {
  GemmArgs g = a;
  const long long tiles_m = (a.M + BM - 1) / BM;
  g.tiles_n = (a.b_rows + BN - 1) / BN;
  const long long nblk = tiles_m * g.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nblk, (unsigned)batches);
  // both operand panels much larger than an XCD's L2: walk the grid in 16 x 16 supertiles (tile_of)
  g.supertile = (tiles_m >= 32 && g.tiles_n >= 32) ? 16 : 0;
  switch (epi) {
    case PFK_EPI_LINEAR: return launch_one<BM, BN, WM, WN, PFK_EPI_LINEAR, VARIANT>(g, grid, st);
    case PFK_EPI_GRU_ZR: return launch_one<BM, BN, WM, WN, PFK_EPI_GRU_ZR, VARIANT>(g, grid, st);
    case PFK_EPI_GRU_Q:  return launch_one<BM, BN, WM, WN, PFK_EPI_GRU_Q, VARIANT>(g, grid, st);
    default: return PFK_ERR_BAD_ARG;
  }
}

// the same for tiles only the LINEAR epilogue is instantiated on (the encoders' widths)
template <int BM, int BN, int WM, int WN, int VARIANT>
int launch_cfg_linear(const GemmArgs& a, int epi, int batches, hipStream_t st) {
  if (epi != PFK_EPI_LINEAR) return PFK_ERR_BAD_ARG;
  GemmArgs g = a;
  const long long tiles_m = (a.M + BM - 1) / BM;
  g.tiles_n = (a.b_rows + BN - 1) / BN;
  const long long nblk = tiles_m * g.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  g.supertile = 0;
  return launch_one<BM, BN, WM, WN, PFK_EPI_LINEAR, VARIANT>(g, dim3((unsigned)nblk, (unsigned)batches), st);
}

int g_force_tile = -1;  // debug/tuning knob, see pfk_debug_set_tile
int g_small_swizzled = 0;   // tuning knob (pfk_debug_set_tile(301)): small grids on the 48 KB swizzled layout too, so that a block of a
                            // concurrently running launch (forked branches) still fits next to two of them on a CU

// Configurations: 0-3 = v1 (64x64, 64x128, 128x128, 128x64); 4-7 = v3 one group, same tiles;
// 8 = v3 two groups 64x64 (in-block split-K); 9 = stream-K on 64x64 tiles (needs a workspace); 10 = cfg 4 on swizzled LDS.
int launch(const GemmArgs& a0, int epi, int batches, hipStream_t st) {
  GemmArgs a = a0;
  a.vec_flags = gemm_vec_flags(a);
  if (a.M >= 0x7fffffffLL || a.Wo <= 0 || a.Ho <= 0) return PFK_ERR_UNSUPPORTED;   // 32-bit pixel arithmetic in the kernels
  fastdiv_make((unsigned)a.Wo, a.wo_mul, a.wo_sh);
  fastdiv_make((unsigned)a.Ho, a.ho_mul, a.ho_sh);
  int cfg;
  if (g_force_tile >= 0) {
    cfg = g_force_tile;
  } else if (a.b_rows >= 2048 && a.M >= 2048) {
    cfg = 11;   // big square-ish GEMM with a short K (correlation volume: 8 K-steps, 198 MB of output per pair): the epilogue
                // dominates, so small tiles with several resident blocks per CU (one storing while the others multiply) beat
                // 128x128 (2.05 vs 2.7 ms at batch 8); round 3: 64x128 on the swizzled layout beats 64x64 x3 — 1815 vs 1868 us at
                // batch 8 (111.9 TFLOP/s = 0.71), 276 vs 299 us at batch 1 (scripts/corr_bench.py, gpurun_out/r3p_corr.log)
  } else {
    const long long tiles64 = (a.M + 63) / 64;
    const long long blocks64 = tiles64 * ((a.b_rows + 63) / 64);
    // Measured on MI355X (scripts/conv_bench.py, batch 1 / 4 / 8): 64x64 tiles with the 3-stage hand-interleaved pipeline win at
    // every batch size; what varies is the schedule around them.
    const double fill = (double)blocks64 / (256.0 * (double)((blocks64 + 255) / 256));
    const bool sk_fits = blocks64 * (long long)a.sk_steps * (SK_MAX_BLOCKS + 1) < 0x7fffffffLL;   // 32-bit unit arithmetic
    // Stream-K where the tile grid quantises badly on 256 CUs and K is long enough to amortise segment prologues + fix-up.
    // Batch 1 (7040 pixels, round-2 sweep): z|r conv 440 tiles x 60 steps 72.5 -> 63.7 us, fh|mask conv1 880 x 36 80.5 -> 75.8,
    // convc2 330 x 72 80.4 -> 62.5; it loses for 220-tile launches (q 41.8 -> 43.5, conv 47.8 -> 48.2: one tile per CU is
    // already balanced) and for short K (convc1 18.7 -> 28.7, mask conv2 25.7 -> 37.6).
    // Round 4: the context hoist shortened the z|r launches to 440 tiles x 40 steps, where the tile grid (all 440 blocks resident
    // at two per CU, sharing the matrix pipe) is 2-4 % ahead again — 51.9 vs 54.1, 49.9 vs 51.4 us (gpurun_out/y_conv_b1.log).
    // What pays for a stream-K launch's segment prologues and fix-up is the work per CU: >= 80 K-steps (17 600 < 20 480 <=
    // 23 760 of convc2, 26 400 of the un-hoisted z|r, 31 680 of fh|mask conv1).
    const bool sk = sk_fits && a.sk_ws != nullptr && batches == 1 && blocks64 > 256 && blocks64 < 4 * 256 && fill < 0.92 &&
                    a.sk_steps >= 24 && blocks64 * (long long)a.sk_steps >= 80LL * 256;
    // a handful of output tiles with a very long K (GEMM-shaped callers with a tall reduction): only stream-K fills the chip
    const bool sk_long = sk_fits && a.sk_ws != nullptr && batches == 1 && blocks64 <= 64 && a.sk_steps >= 128;
    // (Round 4: stream-K on the 1000..4000-tile grids of batch 8 was measured and NOT taken — convc2 (2640 tiles x 72 K-steps)
    //  415 -> 390 us on one box, 412 -> 404 on another; conv (1760 x 72) 275 -> 267 / 273 -> 278; slower wherever K is 36..40 steps
    //  (fm, z|r, q) — while its HBM-side fetch is 10x the tile grid's (508 vs 54 MB per convc2 launch: an XCD's resident blocks
    //  work on distant tiles of its range at once).  profiles/r04_a section 2.)
    // Tile grids: >= 3 tiles per CU: the swizzled 48 KB layout (cfg 10) keeps three blocks resident — also for short K since the
    // LDS epilogue (c1 111 -> 104 us, mask conv2 163 -> 158 us at batch 8).  Below that the padded rows' immediate-offset fragment
    // reads are a few % faster (cfg 4), down to one tile per CU for short K too (convc1 at batch 1: 18.7 vs 19.9 us on the
    // 2-stage kernel); smaller short-K grids keep the 2-stage kernel's cheaper prologue (cfg 0).
    cfg = (sk || sk_long) ? 9 : (blocks64 >= 3 * 256 ? 10 : ((a.sk_steps < 16 && blocks64 < 256) ? 0 : 4));
    // Round 3, batch 8 (scripts/conv_bench.py, three boxes): 64x128 tiles on the swizzled layout (72 KB: two blocks per CU, two
    // accumulators and 32 MFMAs per wave and barrier, 12 instead of 16 B/clk/CU of operand traffic) beat 64x64 x3 where the
    // output width is a multiple of 128 and the grid still has >= 3 rounds of them: fh|mask conv1 526 -> 513 us (129 TFLOP/s),
    // z|r convs 457..528 -> 436 us, convc1 108 -> 101 us; they lose on cout 192 / 126 / 576 (padding) and cout 128 (one column).
    if (g_small_swizzled && (cfg == 4 || cfg == 0)) cfg = 10;
    // Round 5: 96 output channels (BasicEncoder's layer 2, raft/extractor.py:150-151) are 1.5 column tiles of 64 — a quarter of the
    // MFMAs multiplied zero weight rows.  A 128 x 96 tile (four waves stacked in M, wave tile 32 x 96, 2-stage kernel, 64.5 KB: two
    // blocks per CU) covers them exactly: 96->96 3x3 at 110x256 x 16 images 793 -> 627 us (94 -> 119 TFLOP/s of real work), the
    // stride-2 64->96 591 -> 454 us, the 1x1 downsample 132 -> 95 us; the 3-stage kernel on the same tile (84 KB, one block per CU)
    // 680 / 513 / 134 us (gpurun_out/r5b_enc.log).  Same K order per output element: same bits.
    if (epi == PFK_EPI_LINEAR && a.b_rows == 96 && batches == 1 && a.M >= 128 * 256 && cfg != 9) cfg = 15;
    if (cfg == 10 && batches == 1 && a.b_rows >= 256 && (a.b_rows & 127) == 0 && tiles64 * (a.b_rows / 128) >= 6 * 256 &&
        a.sk_steps >= 8)
      cfg = 11;
  }
  // pfk_conv_desc.cout_active: the schedule above was chosen for the FULL output width; the stream-K kernel keeps that tile space
  // and skips the inactive column tiles (split points of the computed tiles unchanged: same bits), every other kernel simply
  // launches fewer column tiles (a tile's K order does not depend on its neighbours)
  if (a.active_tiles_n > 0 && !(cfg == 9 || (cfg >= 30 && cfg <= 41))) {
    a.b_rows = a.active_tiles_n * 64;
    a.active_tiles_n = 0;
  }
  if (cfg >= 50 && cfg < 82) return launch_pp(a, epi, batches, st, cfg - 50);   // 50 + v: persistent pipelined stream-K, schedule variant v
  if (cfg >= 82 && cfg < 85) return launch_pp(a, epi, batches, st, 3, cfg - 81);   // timing ablations 1..3 of variant 3 (swizzled x3, XCD groups)
  if (cfg >= 85 && cfg < 88) return launch_pp(a, epi, batches, st, 2, cfg - 84);   // ... of variant 2 (padded x2, XCD groups)
  switch (cfg) {
    case 0: return launch_cfg<64, 64, 32, 32, 0>(a, epi, batches, st);
    case 1: return launch_cfg<64, 128, 32, 64, 0>(a, epi, batches, st);
    case 2: return launch_cfg<128, 128, 64, 64, 0>(a, epi, batches, st);
    case 3: return launch_cfg<128, 64, 64, 32, 0>(a, epi, batches, st);
    case 4: return launch_cfg<64, 64, 32, 32, 1>(a, epi, batches, st);
    case 5: return launch_cfg<64, 128, 32, 64, 1>(a, epi, batches, st);
    case 6: return launch_cfg<128, 128, 64, 64, 1>(a, epi, batches, st);
    case 7: return launch_cfg<128, 64, 64, 32, 1>(a, epi, batches, st);
    case 8: return launch_cfg<64, 64, 32, 32, 2>(a, epi, batches, st);
    case 9: return (a.sk_ws != nullptr && batches == 1) ? launch_sk(a, epi, st, g_sk_variant) : PFK_ERR_BAD_ARG;
    // 30 + v: stream-K with schedule variant v (launch_sk), tuning only
    case 30: case 31: case 32: case 33: case 34: case 35: case 36: case 37: case 38: case 39: case 40: case 41:
      return (a.sk_ws != nullptr && batches == 1) ? launch_sk(a, epi, st, cfg - 30) : PFK_ERR_BAD_ARG;
    case 10: return launch_cfg<64, 64, 32, 32, 101>(a, epi, batches, st);
    // bigger tiles on the swizzled layout: 64x128 / 128x64 (72 KB: two blocks per CU), 128x128 (96 KB: one)
    case 11: return launch_cfg<64, 128, 32, 64, 101>(a, epi, batches, st);
    case 12: return launch_cfg<128, 64, 64, 32, 101>(a, epi, batches, st);
    case 13: return launch_cfg<128, 128, 64, 64, 101>(a, epi, batches, st);
    // round 5, the encoders' 96-channel layers: a 128 x 96 tile — four waves stacked in M, wave tile 32 x 96 — covers 96 output
    // channels exactly (1.5 column tiles of 64 multiplied zero weight rows in a quarter of their MFMAs): 14 = the 3-stage kernel
    // (84 KB swizzled, one block per CU), 15 = the 2-stage kernel (64.5 KB, two blocks per CU; what launch() selects for cout 96).
    // Measured and not kept (gpurun_out/r5b_enc.log, r5c_enc.log, r5h_conv.log): 256x64 / 128x64 / 128x128 tiles with 64x64 / 32x64 /
    // 32x128 wave tiles on either kernel — 0 .. 90 % slower than the heuristic's choice on every encoder and update-block shape.
    case 14: return launch_cfg_linear<128, 96, 32, 96, 101>(a, epi, batches, st);
    case 15: return launch_cfg_linear<128, 96, 32, 96, 0>(a, epi, batches, st);
#ifdef PFK_BENCH_VARIANTS     // timing ablations (results are garbage; scripts/conv_bench.py only): not in the shipped library
    // MFMA-only skeletons of the bigger padded tiles (timing ablations)
    case 27: return epi == PFK_EPI_LINEAR ? launch_cfg<64, 128, 32, 64, 31>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
    case 28: return epi == PFK_EPI_LINEAR ? launch_cfg<128, 128, 64, 64, 31>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
    case 29: return epi == PFK_EPI_LINEAR ? launch_cfg<128, 64, 64, 32, 31>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
    // timing ablations of cfg 4 (results are garbage; used by scripts/conv_bench.py only)
    case 21: return epi == PFK_EPI_LINEAR ? launch_cfg<64, 64, 32, 32, 11>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
    case 22: return epi == PFK_EPI_LINEAR ? launch_cfg<64, 64, 32, 32, 21>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
    case 23: return epi == PFK_EPI_LINEAR ? launch_cfg<64, 64, 32, 32, 31>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
    case 24: return epi == PFK_EPI_LINEAR ? launch_cfg<64, 64, 32, 32, 41>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
    case 25: return epi == PFK_EPI_LINEAR ? launch_cfg<64, 64, 32, 32, 51>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
    case 26: return epi == PFK_EPI_LINEAR ? launch_cfg<64, 64, 32, 32, 61>(a, epi, batches, st) : PFK_ERR_BAD_ARG;
#endif
    default: return PFK_ERR_BAD_ARG;
  }
}

inline int round_up32(int c) { return (c + 31) & ~31; }

int conv_ktot(const pfk_conv_desc* d, int kpad) {
  if (!d || d->num_src < 1 || d->num_src > 3) return PFK_ERR_BAD_ARG;
  int k = 0;
  for (int s = 0; s < d->num_src; ++s) k += d->kh * d->kw * ((d->src[s].channels + kpad - 1) / kpad * kpad);
  return k;
}

// Validate a conv descriptor and fill everything but the weight operand.  kpad = channels per K-step (32 fp32, 64 bf16).
int desc_to_args(const pfk_conv_desc* d, GemmArgs& a, int kpad) {
  if (d->num_src < 1 || d->num_src > 3) return PFK_ERR_BAD_ARG;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->cout <= 0) return PFK_ERR_BAD_ARG;
  if (d->kh <= 0 || d->kw <= 0 || !(d->kh & 1) || !(d->kw & 1)) return PFK_ERR_BAD_ARG;
  const pfk_conv_src* s = d->src;
  for (int i = 0; i < d->num_src; ++i) {
    if (!s[i].ptr || s[i].channels <= 0 || s[i].ld < s[i].channels) return PFK_ERR_BAD_ARG;
    if (!pfk_aligned16(s[i].ptr) || (s[i].ld & 3) || (s[i].channels & 3)) return PFK_ERR_ALIGNMENT;
  }
  a.src0 = s[0].ptr; a.ld0 = s[0].ld; a.ch0 = s[0].channels;
  if (d->num_src > 1) { a.src1 = s[1].ptr; a.ld1 = s[1].ld; a.ch1 = s[1].channels; }
  if (d->num_src > 2) { a.src2 = s[2].ptr; a.ld2 = s[2].ld; a.ch2 = s[2].channels; }
  a.nsrc = d->num_src;
  const int stride = d->stride > 0 ? d->stride : 1;
  a.H = d->H; a.W = d->W; a.kh = d->kh; a.kw = d->kw;
  a.stride = stride; a.Ho = (d->H - 1) / stride + 1; a.Wo = (d->W - 1) / stride + 1;
  a.relu2 = d->relu_after_residual;
  a.bias = d->bias; a.b_rows = d->cout;
  a.ktot = conv_ktot(d, kpad);
  a.relu = d->relu; a.scale = d->scale;
  a.M = (long long)d->B * a.Ho * a.Wo;
  a.sk_steps = a.ktot / kpad;
  for (int i = 0; i < d->num_src; ++i)   // kernels address sources with 32-bit byte offsets
    if ((long long)d->B * d->H * d->W * s[i].ld * 4 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  switch (d->epilogue) {
    case PFK_EPI_LINEAR:
      if (!d->out || d->out_ld < d->out_coff + d->cout) return PFK_ERR_BAD_ARG;
      a.out = d->out; a.out_ld = d->out_ld; a.out_coff = d->out_coff;
      if (d->residual) {
        if (d->residual_ld < d->cout) return PFK_ERR_BAD_ARG;
        a.residual = d->residual; a.residual_ld = d->residual_ld;
      }
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
  if (d->epilogue != PFK_EPI_LINEAR && d->residual) {
    // GRU epilogues: an additive pre-activation term [M][cout] (the loop-invariant context part), float4 accesses
    if (d->residual_ld < d->cout) return PFK_ERR_BAD_ARG;
    if (!pfk_aligned16(d->residual) || (d->residual_ld & 3)) return PFK_ERR_ALIGNMENT;
    a.residual = d->residual; a.residual_ld = d->residual_ld;
  }
  return PFK_OK;
}

}  // namespace

extern "C" {

int pfk_debug_set_tile(int cfg) {
  if (!pfk_debug_knobs_enabled()) return PFK_ERR_DISABLED;
  if (cfg >= 300) { g_small_swizzled = cfg - 300; return PFK_OK; }      // 301: small grids on the swizzled layout, 300: off
  if (cfg >= 200) { g_sk_variant = cfg - 200; return PFK_OK; }          // 200 + v: stream-K schedule variant (launch_sk)
  if (cfg >= 100) g_bf_cfg = cfg - 100;   // split-bf16 tile configuration (0 = heuristic)
  else { g_force_tile = cfg; if (cfg < 0) g_bf_cfg = 0; }
  return PFK_OK;
}


unsigned pfk_debug_fastdiv(unsigned n, unsigned d) {
  unsigned mul; int sh;
  fastdiv_make(d, mul, sh);
  return fastdiv_u32(n, mul, sh);
}

long long pfk_conv_workspace_bytes(void) { return (long long)SK_WS_BYTES; }

long long pfk_conv_workspace_fault_offset(void) { return (long long)SK_MAX_BLOCKS * 64 * 64 * 4 + (long long)SK_FAULT_SLOT * 4; }

int pfk_conv_ktot(const pfk_conv_desc* d) { return conv_ktot(d, 32); }

int pfk_conv_ktot_bf16(const pfk_conv_desc* d) { return conv_ktot(d, 32); }

int pfk_conv2d_f32(const pfk_conv_desc* d, pfk_stream_t stream) {
  GemmArgs a{};
  if (!d || !d->weight) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(d->weight)) return PFK_ERR_ALIGNMENT;
  const int rc = desc_to_args(d, a, 32);
  if (rc != PFK_OK) return rc;
  a.weight = d->weight;
  if ((long long)d->cout * a.ktot * 4 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  if (d->cout_active > 0 && d->cout_active < d->cout) {
    if (d->epilogue != PFK_EPI_LINEAR || (d->cout_active & 63)) return PFK_ERR_BAD_ARG;
    a.active_tiles_n = d->cout_active / 64;
  }
  if (d->cout_split > 0) {
    if (d->epilogue != PFK_EPI_LINEAR || (d->cout_split & 63) || 2 * d->cout_split != d->cout) return PFK_ERR_BAD_ARG;
    a.sk_split_tiles = d->cout_split / 64;
  }
  if (d->workspace && d->workspace_bytes >= (long long)SK_WS_BYTES && pfk_aligned16(d->workspace)) {
    a.sk_ws = static_cast<float*>(d->workspace);
    a.sk_flags = reinterpret_cast<unsigned*>(static_cast<char*>(d->workspace) + (size_t)SK_MAX_BLOCKS * 64 * 64 * 4);
  }
  return launch(a, d->epilogue, 1, static_cast<hipStream_t>(stream));
}

int pfk_conv2d_group_f32(const pfk_conv_desc* descs, int n, pfk_stream_t stream) {
  if (!descs || n < 1 || n > PFK_CONV_GROUP_MAX) return PFK_ERR_BAD_ARG;
  GemmGroupArgs g{};
  int order[PFK_CONV_GROUP_MAX];
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    const pfk_conv_desc* d = &descs[i];
    if (!d->weight || d->epilogue != PFK_EPI_LINEAR) return PFK_ERR_BAD_ARG;
    if ((d->cout_active > 0 && d->cout_active < d->cout) || d->cout_split > 0) return PFK_ERR_UNSUPPORTED;
    if (!pfk_aligned16(d->weight)) return PFK_ERR_ALIGNMENT;
    order[i] = i;
  }
  // problems with the longest K loop first: their tiles start in the first resident round and the short ones fill in behind them
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && conv_ktot(&descs[order[j]], 32) > conv_ktot(&descs[order[j - 1]], 32); --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  // 64 x 128 tiles (72 KB swizzled: two blocks per CU) where every problem's width is a multiple of 128 and its K loop is long — GMA's
  // per-pair `attn @ v` (cout 128, 220 K-steps): four pairs are 440 tiles = ONE resident round, against 880 tiles of 64 x 64 on 768
  // slots (a second round of 112 tiles); the same K order per output element either way: same bits
  bool wide = true;
  for (int i = 0; i < n; ++i) wide = wide && (descs[i].cout % 128 == 0) && conv_ktot(&descs[i], 32) >= 64 * 32;
  const int BN = wide ? 128 : 64;
  for (int i = 0; i < n; ++i) {
    const pfk_conv_desc* d = &descs[order[i]];
    GemmArgs& a = g.p[i];
    const int rc = desc_to_args(d, a, 32);
    if (rc != PFK_OK) return rc;
    a.weight = d->weight;
    if ((long long)d->cout * a.ktot * 4 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
    a.vec_flags = gemm_vec_flags(a);
    if (a.M >= 0x7fffffffLL || a.Wo <= 0 || a.Ho <= 0) return PFK_ERR_UNSUPPORTED;
    fastdiv_make((unsigned)a.Wo, a.wo_mul, a.wo_sh);
    fastdiv_make((unsigned)a.Ho, a.ho_mul, a.ho_sh);
    a.tiles_n = (a.b_rows + BN - 1) / BN;
    a.supertile = 0;
    const long long nblk = ((a.M + 63) / 64) * a.tiles_n;
    g.first[i] = (int)total;
    total += nblk;
    if (nblk <= 0 || total > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  }
  g.first[n] = (int)total;
  g.n = n;
  // the swizzled 48 KB layout (three blocks per CU) from three tiles per CU up, the padded 55 KB one below — the single launches' rule
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (wide) {
    constexpr size_t smem = (size_t)3 * (64 + 128) * LDS_LDX * sizeof(float);
    auto kern = conv_gemm_v3_group_kernel<LDS_LDX, 128, 64>;
    static pfk_device_once attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, st, g);
  } else if (total >= 3 * 256) {
    constexpr size_t smem = (size_t)3 * (64 + 64) * LDS_LDX * sizeof(float);
    auto kern = conv_gemm_v3_group_kernel<LDS_LDX>;
    static pfk_device_once attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, st, g);
  } else {
    constexpr size_t smem = (size_t)3 * (64 + 64) * LDS_LD * sizeof(float);
    auto kern = conv_gemm_v3_group_kernel<LDS_LD>;
    static pfk_device_once attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, st, g);
  }
  return pfk_launch_status();
}

int pfk_mask_upsample_f32(const float* x, int x_ld, int cin, const float* weight_perm, const float* bias_perm, float scale,
                          const float* flow_pm, int flow_ld, float* out, int B, int H, int W, pfk_stream_t stream) {
  if (!x || !weight_perm || !flow_pm || !out || B <= 0 || H <= 0 || W <= 0 || cin <= 0 || x_ld < cin || flow_ld < 2) return PFK_ERR_BAD_ARG;
  if ((cin & 31) || (x_ld & 3) || !pfk_aligned16(x) || !pfk_aligned16(weight_perm)) return PFK_ERR_ALIGNMENT;
  const long long M = (long long)B * H * W;
  if (M >= 0x7fffffffLL || M * x_ld * 4 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;     // 32-bit byte offsets / pixel arithmetic
  GemmArgs a{};
  a.src0 = x; a.ld0 = x_ld; a.ch0 = cin; a.nsrc = 1;
  a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.stride = 1; a.kh = 1; a.kw = 1;
  a.weight = weight_perm; a.bias = bias_perm; a.b_rows = MU_ROWS; a.ktot = cin;
  a.scale = scale; a.M = M;
  a.mu_flow = flow_pm; a.mu_flow_ld = flow_ld; a.mu_out = out;
  fastdiv_make((unsigned)W, a.wo_mul, a.wo_sh);
  fastdiv_make((unsigned)H, a.ho_mul, a.ho_sh);
  const long long nblk = ((M + MU_BM - 1) / MU_BM) * 4;
  if (nblk > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  constexpr size_t smem = 2 * (MU_BM + MU_BN) * LDS_LDX * sizeof(float);
  static pfk_device_once attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mask_upsample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL(mask_upsample_kernel, dim3((unsigned)nblk), dim3(256), smem, static_cast<hipStream_t>(stream), a);
  return pfk_launch_status();
}

int pfk_conv2d_bf16s(const pfk_conv_desc* d, const void* weight_planes, int nsplit, pfk_stream_t stream) {
  if (d && ((d->cout_active > 0 && d->cout_active < d->cout) || d->cout_split > 0)) return PFK_ERR_UNSUPPORTED;   // the fp32 kernels' features (pfk.h)
  GemmArgs a{};
  if (!d || !weight_planes || nsplit < 1 || nsplit > 3) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(weight_planes)) return PFK_ERR_ALIGNMENT;
  const int rc = desc_to_args(d, a, 32);
  if (rc != PFK_OK) return rc;
  a.wbf = weight_planes;
  a.wbf_plane_bytes = (long long)d->cout * a.ktot * 2;
  if (a.wbf_plane_bytes * nsplit >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  return launch_bf(a, d->epilogue, nsplit, static_cast<hipStream_t>(stream));
}

int pfk_corr_volume_f32(const float* f1, int ld1, const float* f2, int ld2, float* out, int B,
                        int N1, int N2, int D, float scale, pfk_stream_t stream) {
  if (!f1 || !f2 || !out || B <= 0 || N1 <= 0 || N2 <= 0 || D <= 0) return PFK_ERR_BAD_ARG;
  if (ld1 < D || ld2 < D) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(f1) || !pfk_aligned16(f2) || (ld1 & 3) || (ld2 & 3) || (D & 3))
    return PFK_ERR_ALIGNMENT;
  if (ld2 != round_up32(D)) return PFK_ERR_UNSUPPORTED;  // f2 rows are read as packed weight rows
  if ((long long)N1 * ld1 * 4 >= 0x7fffffffLL || (long long)N2 * ld2 * 4 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  GemmArgs a{};
  a.src0 = f1; a.ld0 = ld1; a.ch0 = D; a.nsrc = 1;
  a.H = 1; a.W = N1; a.Ho = 1; a.Wo = N1; a.stride = 1; a.kh = 1; a.kw = 1;
  a.weight = f2; a.bias = nullptr; a.b_rows = N2; a.ktot = ld2;
  a.relu = 0; a.scale = scale;
  a.out = out; a.out_ld = N2; a.out_coff = 0;
  a.M = N1;
  a.sk_steps = a.ktot / BK;
  a.a_bs = (long long)N1 * ld1; a.b_bs = (long long)N2 * ld2; a.o_bs = (long long)N1 * N2;
  return launch(a, PFK_EPI_LINEAR, B, static_cast<hipStream_t>(stream));
}

}  // extern "C"
