// K4-K6 on the gfx950 bf16 matrix cores with *split* operands (v_mfma_f32_32x32x16_bf16: 8 passes for a K of 16,
// i.e. 16x the fp32 MFMA's multiply rate, fp32 accumulate).
//
// An fp32 operand x is written as a sum of bf16 planes, x = x0 + x1 (+ x2) + e, with x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (each subtraction is exact in fp32).  The product of two split
// operands keeps the terms a_i * b_j with i + j < nsplit:
//   nsplit 1:  a0 b0                                   plain bf16 operands          (rel. product error ~2^-9)
//   nsplit 2:  a0 b0 + a0 b1 + a1 b0                   3 MFMAs per K-block          (~2^-17)
//   nsplit 3:  ... + a0 b2 + a1 b1 + a2 b0             6 MFMAs per K-block          (~2^-24: fp32-grade)
// The weights arrive pre-split (planes [nsplit][cout][ktot] bf16, host packing); the activations are split
// while they are staged: global fp32 -> VGPR -> cvt/sub/cvt -> LDS bf16 planes, so HBM traffic is the fp32 path's.
// All kept terms go into one fp32 accumulator (smallest first within a K-block).
//
// Same implicit-GEMM structure as pfk_gemm.hip: K sub-step = (source, tap, 32 channels) in the same order and padding
// as the fp32 packed weight, raw buffer loads with hardware zero fill for the conv padding, un-padded 64-byte LDS rows
// with the 16-byte chunk index XOR-swizzled, two LDS stages, fused epilogues.
// Fragment use (32x32x16): lane l holds A[i = l & 31][k = 8*(l >> 5) .. +7] — one ds_read_b128 of chunk
// kb*2 + (l >> 5) per plane per 16-channel K-block kb.
#include "pfk_gemm.h"

#include <mutex>
#include <utility>

using namespace pfkg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BKB = 32;            // channels per K sub-step (same K order / padding as the fp32 path's packed weight)
constexpr int ROWB = 64;           // bytes per LDS row: 32 bf16

// TH threads (256 or 512) stage SUB K-sub-steps per barrier: thread t owns channels (t & 3)*8 .. +7 of rows (t >> 2) + RP*i
// (RP = TH / 4 rows per pass) of both operands.  LDS rows are 64 bytes = four 16-byte chunks; chunk' = chunk ^ ((row >> 2) & 3) makes both the
// ds_write_b128 (8 lanes = 2 rows) and the fragment ds_read_b128 (MI355X_MICROARCH.md lane groups) conflict-free.
// BDMA (round 4, experiment behind pfk_debug_set_tile(180 + t)): the weight planes — three quarters of a K sub-step's bytes, and no
// split to do on them — go global -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA: no VGPR round trip, no ds_write_b128) straight
// into the stage the NEXT step reads; a wave's 64 lanes land on 1 KiB of consecutive LDS (16 rows x 64 B), so the XOR swizzle is
// applied to WHICH global chunk a lane fetches instead of where it stores it.
template <int NS, int BM, int BN, int SUB, int TH = 256, int NSETS = 2, bool BDMA = false>
struct StagerBF {
  static constexpr int RP = TH / 4;
  static_assert(BM % RP == 0 && BN % RP == 0, "tile rows must be a multiple of the rows staged per pass");
  static constexpr int A_PT = BM / RP, B_PT = BN / RP;
  static constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB, SUBSTAGE = NS * (A_PLANE + B_PLANE);
  int H, W, kh, kw, ph, pw, nsrc;
  int ld0, ld1, ld2, ch0, ch1, ch2;
  __amdgpu_buffer_rsrc_t rs0, rs1, rs2, rsw, rs;
  int cld, cch;
  int plane_bytes;
  int seg = 0, ky = 0, kx = 0, c0 = 0, kofs = 0;
  int pos = 0, total;
  int c8, r0;
  unsigned sbyte;             // byte offset of this thread's 16-byte chunk inside a plane (row r0; row r0 + RP*i: + i*RP*ROWB)
  int prow[A_PT], py[A_PT], px[A_PT];
  bool pok[A_PT];
  unsigned abase[A_PT], aoff[A_PT], wvoff[B_PT];
  // NSETS = 2: two register sets, the loads of step j+2 are issued while step j+1's wait to be split; NSETS = 1 (the 64x64-wave-tile
  // configurations, which have no registers to spare): a register is split + stored and then reloaded within the same step
  f32x4 ra[NSETS][SUB][A_PT][2];
  u32x4 rb[BDMA ? 1 : NSETS][BDMA ? 1 : SUB][BDMA ? 1 : NS][BDMA ? 1 : B_PT];
  int posb = 0, pb_koff = 0;      // BDMA: the weight stream's own K position (one step ahead of the MFMAs, one behind the A loads)
  unsigned bwave = 0;             // BDMA: byte offset of this wave's 16 rows inside a 64-row... pass of a plane (wave-uniform)

  __device__ __forceinline__ StagerBF(const GemmArgs& a, long long m0, int n0, int t) {
    H = a.H; W = a.W; kh = a.kh; kw = a.kw; ph = a.kh >> 1; pw = a.kw >> 1; nsrc = a.nsrc;
    ld0 = a.ld0; ld1 = a.ld1; ld2 = a.ld2; ch0 = a.ch0; ch1 = a.ch1; ch2 = a.ch2;
    rs0 = make_rsrc(a.src0);
    rs1 = make_rsrc(a.src1);
    rs2 = make_rsrc(a.src2);
    rsw = make_rsrc(a.wbf);
    plane_bytes = (int)a.wbf_plane_bytes;
    total = a.sk_steps;
    c8 = (t & 3) * 8;
    r0 = t >> 2;
    sbyte = (unsigned)(r0 * ROWB + (((t & 3) ^ ((r0 >> 2) & 3)) << 4));
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      // multiplier arithmetic (launch_bf fills the multipliers; M < 2^31): three instructions per division instead of ~100
      const unsigned p = (unsigned)m0 + (unsigned)(r0 + RP * i);
      pok[i] = (long long)p < a.M;
      const unsigned prow_o = fastdiv_u32(p, a.wo_mul, a.wo_sh);            // b*Ho + yo
      const unsigned bimg = fastdiv_u32(prow_o, a.ho_mul, a.ho_sh);
      px[i] = (int)(p - prow_o * (unsigned)a.Wo) * a.stride;                // input coordinates of the centre tap
      py[i] = (int)(prow_o - bimg * (unsigned)a.Ho) * a.stride;
      prow[i] = (int)((bimg * (unsigned)a.H + (unsigned)py[i]) * (unsigned)a.W + (unsigned)px[i]);
    }
    // BDMA: lane (row r0, physical chunk t & 3) fetches the LOGICAL chunk that belongs there
    const int c8b = BDMA ? (((t & 3) ^ ((r0 >> 2) & 3)) * 8) : c8;
    bwave = (unsigned)__builtin_amdgcn_readfirstlane(t >> 6) * 1024u;
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int n = n0 + r0 + RP * i;
      wvoff[i] = n < a.b_rows ? (unsigned)(n * a.ktot + c8b) * 2u : OOB;
    }
    set_segment(0);
    set_tap();
  }

  __device__ __forceinline__ void set_segment(int s) {
    if (s == 0) { rs = rs0; cld = ld0; cch = ch0; }
    else if (s == 1) { rs = rs1; cld = ld1; cch = ch1; }
    else { rs = rs2; cld = ld2; cch = ch2; }
#pragma unroll
    for (int i = 0; i < A_PT; ++i) abase[i] = (unsigned)(prow[i] * cld + c8) * 4u;
  }

  __device__ __forceinline__ void set_tap() {
    const int dy = ky - ph, dx = kx - pw;
    const unsigned toff = (unsigned)((dy * W + dx) * cld * 4);
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const bool ok = pok[i] && (unsigned)(py[i] + dy) < (unsigned)H && (unsigned)(px[i] + dx) < (unsigned)W;
      aoff[i] = ok ? abase[i] + toff : OOB;
    }
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

  // ---- the staging work of one step, cut into small "units" that the kernel places between MFMAs ----------------
  // (a wave issues in order: whatever does not sit between two MFMAs in program order runs with the matrix pipe idle).
  // Register set S receives the loads of step j+2 while set 1-S (loaded a whole step ago) is split and stored.
  // All indices are template constants so every register array stays in registers.
  bool p_ok0, p_ok1;
  int p_coff, p_koff;
  u32x4 pa[NS];   // planes of the row chunk being split

  // Past the end of K every A lane is out of range (zeros) and B is parked on K-step 0 (valid memory, multiplied by
  // zeros): the K loop stays branch-free.
  __device__ __forceinline__ void load_setup() {
    const bool live = pos < total;
    const int lim = live ? cch - c0 : 0;
    p_ok0 = c8 < lim; p_ok1 = c8 + 4 < lim;   // sources have a multiple of 4 channels
    p_coff = c0 * 4;
    p_koff = live ? kofs * 2 : 0;
  }
  template <int S, int U, int I, int Hf>
  __device__ __forceinline__ void load_a() {
    const unsigned o = aoff[I];
    const unsigned off = Hf == 0 ? (p_ok0 ? o : OOB) : ((p_ok1 && o != OOB) ? o + 16u : OOB);
    ra[S][U][I][Hf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, p_coff, 0));
  }
  template <int S, int U, int PL, int I>
  __device__ __forceinline__ void load_b() {
    rb[S][U][PL][I] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[I], p_koff + PL * plane_bytes, 0);
  }
  __device__ __forceinline__ void load_done() {
    ++pos;
    if (pos < total) advance();
  }
  // plane PL of half Hf (4 floats) of row chunk (U, I) of set S -> dwords 2*Hf, 2*Hf+1 of pa[PL]; the running
  // residual lives in `sr` between the planes of one half
  f32x4 sr;
  template <int S, int U, int I, int Hf, int PL>
  __device__ __forceinline__ void split_piece() {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (PL == 0) sr = ra[S][U][I][Hf];
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (__bf16)sr[e];                           // round to nearest even (v_cvt_pk_bf16_f32)
      if constexpr (PL + 1 < NS) sr[e] = sr[e] - (float)h[e];   // exact
    }
    const u32x2 w = __builtin_bit_cast(u32x2, h);
    pa[PL][2 * Hf] = w[0];
    pa[PL][2 * Hf + 1] = w[1];
  }
  // BDMA: plane PL of the weight rows of pass I, K sub-step `posb`, into sub-step slot U of `stage`
  __device__ __forceinline__ void b_setup() { pb_koff = posb < total ? posb * BKB * 2 : 0; }
  template <int U, int PL, int I>
  __device__ __forceinline__ void dma_b(char* stage) const {
#if defined(__HIP_DEVICE_COMPILE__)       // (the host pass of hipcc does not know the LDS-DMA builtin)
    char* dst = stage + U * SUBSTAGE + NS * A_PLANE + PL * B_PLANE + I * RP * ROWB + bwave;       // wave-uniform: goes to M0
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)dst, 16, wvoff[I],
                                             pb_koff + PL * plane_bytes, 0, 0);
#endif
  }
  // stage layout: [sub-step][A planes 0..NS-1][B planes 0..NS-1], planes [rows][64 B]
  template <int U, int I, int PL>
  __device__ __forceinline__ void store_a(char* stage) const {
    *reinterpret_cast<u32x4*>(stage + U * SUBSTAGE + PL * A_PLANE + I * RP * ROWB + sbyte) = pa[PL];
  }
  template <int S, int U, int PL, int I>
  __device__ __forceinline__ void store_b(char* stage) const {
    *reinterpret_cast<u32x4*>(stage + U * SUBSTAGE + NS * A_PLANE + PL * B_PLANE + I * RP * ROWB + sbyte) = rb[S][U][PL][I];
  }

  // Unit X of a step, in order per sub-step: B stores | per row chunk: split pieces, A stores | load setup, A loads,
  // B loads, iterator advance.
  static constexpr int U_BST = NS * B_PT, U_ROW = 3 * NS, U_AST = A_PT * U_ROW, U_ALD = 2 * A_PT, U_BLD = BDMA ? 0 : NS * B_PT;
  static constexpr int A_LOADS_PER_SUB = U_ALD;        // BDMA: the loads that may stay in flight behind the step's last DMA
  static constexpr int UNITS_PER_SUB = U_BST + U_AST + 1 + U_ALD + U_BLD + 1;
  static constexpr int UNITS = SUB * UNITS_PER_SUB;
  template <int S, int X>
  __device__ __forceinline__ void unit(char* other) {
    constexpr int U = X / UNITS_PER_SUB, x = X % UNITS_PER_SUB;
    constexpr int SL = NSETS == 2 ? S : 0, SS = NSETS == 2 ? 1 - S : 0;     // set that receives the loads / set that is stored
    if constexpr (x < U_BST) {
      if constexpr (BDMA) {          // the step's first units: the DMA has the whole step to land
        if constexpr (x == 0) b_setup();
        dma_b<U, x / B_PT, x % B_PT>(other);
        if constexpr (x == U_BST - 1) ++posb;
      } else {
        store_b<SS, U, x / B_PT, x % B_PT>(other);
      }
    } else if constexpr (x < U_BST + U_AST) {
      constexpr int y = x - U_BST, I = y / U_ROW, z = y % U_ROW;
      if constexpr (z < 2 * NS) split_piece<SS, U, I, z / NS, z % NS>();
      else store_a<U, I, z - 2 * NS>(other);
    } else if constexpr (x == U_BST + U_AST) {
      load_setup();
    } else if constexpr (x < U_BST + U_AST + 1 + U_ALD) {
      constexpr int y = x - (U_BST + U_AST + 1);
      load_a<SL, U, y / 2, y % 2>();
    } else if constexpr (x < U_BST + U_AST + 1 + U_ALD + U_BLD) {
      constexpr int y = x - (U_BST + U_AST + 1 + U_ALD);
      load_b<SL, U, y / B_PT, y % B_PT>();
    } else {
      load_done();
    }
  }
  template <int S, int... X>
  __device__ __forceinline__ void units_plain(char* other, std::integer_sequence<int, X...>) {
    (unit<S, X>(other), ...);
  }
  // prologue helpers (no interleaving needed)
  template <int S>
  __device__ __forceinline__ void load_all() {   // the load units of a step, nothing else
    load_only<S>(std::make_integer_sequence<int, UNITS>{});
  }
  template <int S, int... X>
  __device__ __forceinline__ void load_only(std::integer_sequence<int, X...>) {
    ((X % UNITS_PER_SUB >= U_BST + U_AST ? unit<S, X>(nullptr) : (void)0), ...);
  }
  template <int S>
  __device__ __forceinline__ void store_all(char* stage) {   // the split + store units of set S
    store_only<1 - S>(stage, std::make_integer_sequence<int, UNITS>{});
  }
  template <int S, int... X>
  __device__ __forceinline__ void store_only(char* stage, std::integer_sequence<int, X...>) {
    ((X % UNITS_PER_SUB < U_BST + U_AST ? unit<S, X>(stage) : (void)0), ...);
  }
};

// Block tile BM x BN, WM x WN waves (2 x 2, or 2 x 4 with 512 threads), wave tile (BM/WM) x (BN/WN) = MT x NT 32x32 MFMA blocks;
// two LDS stages of SUB K-sub-steps each.  Big tiles amortise the split (VALU) and the LDS traffic over more MFMAs — per 16-channel
// K-block a wave reads (MT + NT) * NS fragments for MT * NT * NS(NS+1)/2 MFMAs.
// t-th kept product term a_i * b_j (i + j < NS), smallest first
template <int NS> constexpr int term_a(int t) {
  int c = 0;
  for (int o = NS - 1; o >= 0; --o)
    for (int i = 0; i <= o; ++i) { if (c == t) return i; ++c; }
  return 0;
}
template <int NS> constexpr int term_b(int t) {
  int c = 0;
  for (int o = NS - 1; o >= 0; --o)
    for (int i = 0; i <= o; ++i) { if (c == t) return o - i; ++c; }
  return 0;
}

// Term order of the single-fragment-buffer schedule (three planes): a1 b1 | a0 b2 | a2 b0 | a0 b1 | a1 b0 | a0 b0 — still the three
// 2^-16 terms first and a0 b0 last, but plane 2 retires after the third term, plane 1 after the fifth and plane 0 after the sixth,
// and the block opens with the one term that needs no plane-0 fragment: each plane's registers are re-read for the NEXT K-block
// as soon as they retire, and the plane-0 reads (issued behind the block's last MFMA) have the a1 b1 MFMAs of the next block to land.
constexpr int term1_a(int t) { constexpr int v[6] = {1, 0, 2, 0, 1, 0}; return v[t]; }
constexpr int term1_b(int t) { constexpr int v[6] = {1, 2, 0, 1, 0, 0}; return v[t]; }

// One K-step of a wave as a hand-placed instruction stream: MFMA, filler, MFMA, filler ...  (sched_barrier(0) after
// every item pins the order).  A step has NB = 2*SUB 16-channel K-blocks of MPB = T*MT*NT MFMAs; fragments are
// register double-buffered per K-block: the reads of block g+1 are the first fillers of block g; the staging units of
// StagerBF::unit are spread evenly over all MFMA slots.  Only the first block's fragment reads (right after the
// barrier) are exposed — the co-resident block's wave covers them.
template <int EPI, int NS, int BM, int BN, int SUB, int WM = 2, int WN = 2, int NSETS = 2, bool BDMA = false>
struct KStep {
  // fragment registers: double-buffered per K-block, except three planes on 64x64 wave tiles (96 registers would not fit next to
  // the 64 accumulators): there the next block's fragments are read behind the block's last MFMA and the SIMD's other wave covers
  // the LDS latency
  static constexpr int FD = (NS == 3 && WM * WN == 8 && BM / (32 * WM) * (BN / (32 * WN)) >= 4) ? 1 : 2;
  using St = StagerBF<NS, BM, BN, SUB, 64 * WM * WN, NSETS, BDMA>;
  static constexpr int MT = BM / (32 * WM), NT = BN / (32 * WN);
  static constexpr int T = NS * (NS + 1) / 2;
  static constexpr int NB = 2 * SUB, MPB = T * MT * NT, NM = NB * MPB;
  static constexpr int NFR = (MT + NT) * NS;          // fragment reads per K-block
  static constexpr int RU = (NFR + 1) / 2;            // ... as units of two
  static_assert(RU <= MPB, "not enough MFMA slots for the fragment reads");
  int a_row, b_row, ko0, ko1;
  bf16x8 fa[FD][MT][NS], fb[FD][NT][NS];

  template <int G, int E>
  __device__ __forceinline__ void read_one(const char* stage) {
    if constexpr (E < NFR) {
      const char* base = stage + (G >> 1) * St::SUBSTAGE + ((G & 1) ? ko1 : ko0);
      if constexpr (E < MT * NS)
        fa[G & (FD - 1)][E / NS][E % NS] = *reinterpret_cast<const bf16x8*>(base + a_row + (E % NS) * St::A_PLANE + (E / NS) * 32 * ROWB);
      else
        fb[G & (FD - 1)][(E - MT * NS) / NS][(E - MT * NS) % NS] =
            *reinterpret_cast<const bf16x8*>(base + b_row + ((E - MT * NS) % NS) * St::B_PLANE + ((E - MT * NS) / NS) * 32 * ROWB);
    }
  }
  template <int G, int... E>
  __device__ __forceinline__ void read_block(const char* stage, std::integer_sequence<int, E...>) {
    (read_one<G, E>(stage), ...);
  }
  // plane PL of every A and B fragment of K-block G
  template <int G, int PL, int... I>
  __device__ __forceinline__ void read_plane_seq(const char* stage, std::integer_sequence<int, I...>) {
    ((I < MT ? read_one<G, (I < MT ? I : 0) * NS + PL>(stage) : read_one<G, MT * NS + (I < MT ? 0 : I - MT) * NS + PL>(stage)), ...);
  }
  template <int G, int PL>
  __device__ __forceinline__ void read_plane(const char* stage) {
    read_plane_seq<G, PL>(stage, std::make_integer_sequence<int, MT + NT>{});
  }

  template <int S, int Q>
  __device__ __forceinline__ void slot(f32x16 (&acc)[MT][NT], St& st, const char* cur, char* other) {
    constexpr int G = Q / MPB, R = Q % MPB;
    constexpr int t = R / (MT * NT), mt = (R / NT) % MT, nt = R % NT;
    constexpr int ta = FD == 1 ? term1_a(t) : term_a<NS>(t), tb = FD == 1 ? term1_b(t) : term_b<NS>(t);
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[G & (FD - 1)][mt][ta], fb[G & (FD - 1)][nt][tb], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FD == 2 && G + 1 < NB && R < RU) {       // next K-block's fragments, >= MPB - RU MFMAs ahead of their first use
      read_one<G + 1, 2 * R>(cur);
      read_one<G + 1, 2 * R + 1>(cur);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FD == 1 && G + 1 < NB) {                 // single buffer: a plane is re-read for the next K-block when it retires
      constexpr int MN = MT * NT;
      if constexpr (R == 3 * MN - 1) { read_plane<G + 1, 2>(cur); __builtin_amdgcn_sched_barrier(0); }
      else if constexpr (R == 5 * MN - 1) { read_plane<G + 1, 1>(cur); __builtin_amdgcn_sched_barrier(0); }
      else if constexpr (R == 6 * MN - 1) { read_plane<G + 1, 0>(cur); __builtin_amdgcn_sched_barrier(0); }
    }
    constexpr int X0 = Q * St::UNITS / NM, X1 = (Q + 1) * St::UNITS / NM;
    unit_range<S, X0>(st, other, std::make_integer_sequence<int, X1 - X0>{});
  }
  template <int S, int X0, int... D>
  __device__ __forceinline__ void unit_range(St& st, char* other, std::integer_sequence<int, D...>) {
    ((st.template unit<S, X0 + D>(other), __builtin_amdgcn_sched_barrier(0)), ...);
  }
  template <int S, int... Q>
  __device__ __forceinline__ void slots(f32x16 (&acc)[MT][NT], St& st, const char* cur, char* other, std::integer_sequence<int, Q...>) {
    (slot<S, Q>(acc, st, cur, other), ...);
  }
  template <int S>
  __device__ __forceinline__ void run(f32x16 (&acc)[MT][NT], St& st, const char* cur, char* other) {
    if constexpr (FD == 1) {       // in the order the terms need them: a1 b1, then a0 (b2), then a2 b0
      read_plane<0, 1>(cur);
      __builtin_amdgcn_sched_barrier(0);
      read_plane<0, 0>(cur);
      read_plane<0, 2>(cur);
    } else {
      read_block<0>(cur, std::make_integer_sequence<int, NFR>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    slots<S>(acc, st, cur, other, std::make_integer_sequence<int, NM>{});
  }
};

// two resident blocks per CU (=> <= 256 registers) whenever two of them fit in the 160 KB of LDS; an eight-wave block stays alone
// (two of them would cap a lane at 128 registers)
constexpr int bf_blocks_per_cu(int ns, int bm, int bn, int sub, int waves, int nsets) {
  // (two resident eight-wave blocks of the two-plane 128x128 tile would need <= 128 registers: 32 bytes of scratch, not built)
  return (waves <= 4 && 2 * (2 * sub * ns * (bm + bn) * ROWB) <= 160 * 1024) ? 2 : 1;
}

// end of a K step.  BDMA: the weight planes of the next step were DMA'd into the other stage by the step's first units; the A loads
// issued behind them may stay in flight (counted vmcnt), the LDS stores of the A planes and this wave's fragment reads may not.
template <bool BDMA, int A_LOADS>
__device__ __forceinline__ void step_barrier() {
  if constexpr (BDMA) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(A_LOADS) : "memory");
  } else {
    __syncthreads();
  }
}

template <int EPI, int NS, int BM, int BN, int SUB, int WM = 2, int WN = 2, int NSETS = 2, bool BDMA = false>
// (hipcc reads the second launch-bounds argument as waves per SIMD: resident blocks x waves per block / 4)
__global__ __launch_bounds__(64 * WM * WN, bf_blocks_per_cu(NS, BM, BN, SUB, WM * WN, NSETS) * WM * WN / 4) void conv_gemm_bf_kernel(const GemmArgs a) {
  static_assert(!BDMA || SUB == 1, "the counted vmcnt of step_barrier assumes one K sub-step per barrier");
  using St = StagerBF<NS, BM, BN, SUB, 64 * WM * WN, NSETS, BDMA>;
  constexpr int MT = BM / (32 * WM), NT = BN / (32 * WN);
  constexpr int A_PLANE = St::A_PLANE, STAGE = SUB * St::SUBSTAGE;
  extern __shared__ __attribute__((aligned(16))) char smem_bf[];   // [2][STAGE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm0 = (wid / WN) * (BM / WM);
  const int wn0 = (wid % WN) * (BN / WN);

  const int bid = pfk_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n;
  const int tile_m = bid / a.tiles_n + a.m_tile_base;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  St st(a, m0, n0, tid);
  const int nsteps = (st.total + SUB - 1) / SUB;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int frow = lane & 31;
  const int hl = lane >> 5;
  const int key = (frow >> 2) & 3;   // wave / block row offsets are multiples of 32: same key
  const int ko0 = ((0 + hl) ^ key) << 4, ko1 = ((2 + hl) ^ key) << 4;
  const int a_row = (wm0 + frow) * ROWB;
  const int b_row = NS * A_PLANE + (wn0 + frow) * ROWB;

  // Pipeline: registers hold steps j+1 (loaded a whole step ago; split + stored into the other LDS stage during step j)
  // and j+2 (issued during step j); LDS holds steps j and j+1; one barrier per step.
  char* const stage0 = smem_bf;
  char* const stage1 = smem_bf + STAGE;
  KStep<EPI, NS, BM, BN, SUB, WM, WN, NSETS, BDMA> ks{a_row, b_row, ko0, ko1};
  if constexpr (NSETS == 2) {
    st.template load_all<0>();
    st.template load_all<1>();
    st.template store_all<0>(stage0);
  } else {                      // one register set: step 0 through the registers into stage 0, then step 1 into the registers
    st.template load_all<0>();
    st.template store_all<0>(stage0);
    st.template load_all<0>();
  }
  if constexpr (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // step 0's weight planes have landed in stage 0
  __syncthreads();
  for (int step = 0; step < nsteps; step += 2) {
    ks.template run<0>(acc, st, stage0, stage1);   // MFMAs on stage0, loads -> set 0, set 1 -> stage1
    step_barrier<BDMA, St::A_LOADS_PER_SUB>();
    if (step + 1 >= nsteps) break;
    ks.template run<1>(acc, st, stage1, stage0);
    step_barrier<BDMA, St::A_LOADS_PER_SUB>();
  }
  if constexpr (BDMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // nothing may land in the epilogue's LDS
  epilogue_lds<MT, NT, EPI>(a, acc, m0 + wm0, n0 + wn0, lane, 0, reinterpret_cast<float*>(smem_bf), wid);   // behind the loop's final barrier
}

template <int EPI, int NS, int BM, int BN, int SUB, int WM = 2, int WN = 2, int NSETS = 2, bool BDMA = false>
int launch_bf_one(const GemmArgs& a, hipStream_t st, long long row0, long long row1) {
  GemmArgs g = a;
  if (a.M >= 0x7fffffffLL || a.Wo <= 0 || a.Ho <= 0) return PFK_ERR_UNSUPPORTED;   // 32-bit pixel arithmetic in the stager
  fastdiv_make((unsigned)a.Wo, g.wo_mul, g.wo_sh);
  fastdiv_make((unsigned)a.Ho, g.ho_mul, g.ho_sh);
  // output rows [row0, row1) of M (row0 a multiple of BM; row1 < 0: to the end)
  if (row1 < 0 || row1 > a.M) row1 = a.M;
  if (row0 % BM || row0 >= row1) return row0 >= row1 && row0 <= a.M ? PFK_OK : PFK_ERR_BAD_ARG;
  g.m_tile_base = (int)(row0 / BM);
  const long long tiles_m = (row1 - row0 + BM - 1) / BM;
  g.tiles_n = (a.b_rows + BN - 1) / BN;
  const long long nblk = tiles_m * g.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  constexpr size_t smem = 2 * (size_t)SUB * NS * (BM + BN) * ROWB;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static_assert(WM * WN * 4096 <= (int)smem, "the LDS epilogue needs 4 KB per wave");
  auto kern = conv_gemm_bf_kernel<EPI, NS, BM, BN, SUB, WM, WN, NSETS, BDMA>;
  static pfk_device_once attr_once;   // one per template instantiation and device; safe with several host threads
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WM * WN), smem, st, g);
  return pfk_launch_status();
}

template <int NS, int BM, int BN, int SUB, int WM = 2, int WN = 2, int NSETS = 2, bool BDMA = false>
int launch_bf_epi(const GemmArgs& g, int epi, hipStream_t st, long long row0, long long row1) {
  switch (epi) {
    case PFK_EPI_LINEAR: return launch_bf_one<PFK_EPI_LINEAR, NS, BM, BN, SUB, WM, WN, NSETS, BDMA>(g, st, row0, row1);
    case PFK_EPI_GRU_ZR: return launch_bf_one<PFK_EPI_GRU_ZR, NS, BM, BN, SUB, WM, WN, NSETS, BDMA>(g, st, row0, row1);
    case PFK_EPI_GRU_Q:  return launch_bf_one<PFK_EPI_GRU_Q, NS, BM, BN, SUB, WM, WN, NSETS, BDMA>(g, st, row0, row1);
    default: return PFK_ERR_BAD_ARG;
  }
}

// tile configurations: 1 = 64x64 (two sub-steps per barrier), 2 = 128x64, 3 = 128x128, 4 = 128x128 with eight waves (2 x 4: every
// thread splits and stages half as much per MFMA)
template <int NS>
int launch_bf_ns(const GemmArgs& g, int epi, int cfg, hipStream_t st, long long row0 = 0, long long row1 = -1) {
  switch (cfg) {
    case 1: return launch_bf_epi<NS, 64, 64, 2>(g, epi, st, row0, row1);
    case 2: return launch_bf_epi<NS, 128, 64, 1>(g, epi, st, row0, row1);
    case 3: return launch_bf_epi<NS, 128, 128, 1>(g, epi, st, row0, row1);
    case 4: return launch_bf_epi<NS, 128, 128, 1, 2, 4>(g, epi, st, row0, row1);
    // 64x64 wave tiles (half the fragment bytes per MFMA of the configurations above) with eight waves: 256x128 / 128x256
    case 5: return launch_bf_epi<NS, 256, 128, 1, 4, 2, 1>(g, epi, st, row0, row1);
    case 6: return launch_bf_epi<NS, 128, 256, 1, 2, 4, 1>(g, epi, st, row0, row1);
    // the eight-wave tiles with the weight planes by LDS-DMA (two / three planes; `launch_bf` adds 10 to the tile number)
    case 14: if constexpr (NS >= 2) return launch_bf_epi<NS, 128, 128, 1, 2, 4, 2, true>(g, epi, st, row0, row1); else return PFK_ERR_BAD_ARG;
    case 15: if constexpr (NS >= 2) return launch_bf_epi<NS, 256, 128, 1, 4, 2, 1, true>(g, epi, st, row0, row1); else return PFK_ERR_BAD_ARG;
    case 16: if constexpr (NS >= 2) return launch_bf_epi<NS, 128, 256, 1, 2, 4, 1, true>(g, epi, st, row0, row1); else return PFK_ERR_BAD_ARG;
    default: return PFK_ERR_BAD_ARG;
  }
}

}  // namespace

namespace pfkg {

int g_bf_cfg = 0;   // debug/tuning knob (pfk_debug_set_tile(100 + cfg)); 0 = heuristic

int launch_bf(const GemmArgs& a0, int epi, int nsplit, hipStream_t st) {
  GemmArgs a = a0;
  a.vec_flags = gemm_vec_flags(a);
  int cfg = g_bf_cfg % 10;
  a.m_tile_base = 0;
  if (cfg == 0) {
    // Measured on MI355X (scripts/conv_bench.py, batch 1 and 8, RAFT update-block shapes).  The kernels are bound by LDS
    // traffic (ds_write_b128 moves ~80 B/clk/CU), so the biggest tile that still fills the chip wins; three planes only
    // fit two resident blocks per CU up to 128x64.
    const long long tm = (a.M + 127) / 128;
    const long long b2 = tm * ((a.b_rows + 63) / 64), b3 = tm * ((a.b_rows + 127) / 128);
    const int pad128 = (a.b_rows + 127) / 128 * 128;
    // three planes (round 3, scripts/conv_bench.py at batch 1 / 2 / 8): the eight-wave 128x128 tile wins wherever it has >= 200
    // blocks and pads <= 10 % of the columns — fh|mask conv1 397 -> 357 us (186 TFLOP/s of fp32-equivalent work), z|r 343 -> 331,
    // q 187 -> 161, conv 218 -> 193 at batch 8; fh|mask conv1 56.7 -> 50.2 us at batch 1 — and loses below (z|r at batch 1: 110
    // blocks, 63.6 -> 73.4 us) and on cout 192 / 576 / 64
    if (nsplit == 3) cfg = (b3 >= 200 && (pad128 - a.b_rows) * 10 <= a.b_rows) ? 4 : (b2 >= 160 ? 2 : 1);
    else if (b3 >= 384) cfg = (pad128 - a.b_rows) * 10 > a.b_rows ? 2 : 3;       // > 10 % padded columns at BN = 128
    else cfg = b2 >= 400 ? 2 : 1;
    // 64x64 wave tiles on eight waves (256x128 / 128x256: half the fragment bytes per MFMA — the split kernels are LDS-bound — and
    // half the split work): they win from 220 blocks up on K >= 10 steps, even with a third of the columns padded (cout 192 -> 256);
    // batch 8, three planes: z|r 335 -> 288 us, q 183 -> 144, conv 214 -> 172, fh|mask conv1 360 -> 340, convc2 313 -> 288;
    // 110 blocks lose everywhere (q at batch 4: 85 -> 122 us), mask conv2 (8 K-steps) ties.  Two planes: the same except the
    // largest grid (fh|mask conv1 at batch 8: 197 us on two resident 128x128 blocks vs 205).
    // One plane (plain bf16 operands, batch 8, gpurun_out/r3y_conv_b8.log): eight waves on the 128x128 tile beat two four-wave
    // blocks by 3-7 % (fh|mask conv1 113.9 -> 106.1 us = 626 TFLOP/s, q 69 -> 66); 128x256 wins on z|r (123 -> 112), convc2
    // (113 -> 93) and convc1, 256x128 nowhere.
    if (nsplit == 1 && cfg == 3) cfg = 4;
    if (a.sk_steps >= 10) {
      const int pad256 = (a.b_rows + 255) / 256 * 256;
      const long long nb6 = tm * (pad256 / 256), nb5 = ((a.M + 255) / 256) * (pad128 / 128);
      const bool skip = nsplit <= 2 && nb6 >= 800 && (cfg == 3 || cfg == 4);
      if (!skip && (pad256 - a.b_rows) * 3 <= a.b_rows && nb6 >= 220) cfg = 6;
      else if (!skip && nsplit >= 2 && (pad128 - a.b_rows) * 3 <= a.b_rows && nb5 >= 220) cfg = 5;
    }
  }
  // Tail split (round 3): the 64x64-wave-tile configurations run ONE block per CU, so a grid of 3.44 rounds of 256 blocks costs 4
  // (fh|mask conv1 at batch 8: 880 blocks: 344 -> 329 us with the split).  When the last round is at most 60 % full (at 72 % — the
  // z|r convolutions' 440 blocks — the second launch's own ramp and tail cost more than the round they save: 293 -> 326 us), the big tiles take the whole rounds and the
  // remaining rows go to the eight-wave 128x128 tile (half the work per block, twice the blocks) in a second launch on disjoint rows.
  long long split_row = -1;
  if ((cfg == 5 || cfg == 6) && g_bf_cfg / 10 != 9) {       // pfk_debug_set_tile(190 + t): tail split off (A/B timing)
    const int bm = cfg == 5 ? 256 : 128, bn = cfg == 5 ? 128 : 256;
    const long long tn = (a.b_rows + bn - 1) / bn, tmr = (a.M + bm - 1) / bm, nblk = tmr * tn;
    const long long full = nblk / 256 * 256, rem = nblk - full;
    if (full > 0 && rem > 0 && rem * 10 <= 256 * 6) {
      const long long rows = full / tn * bm;              // whole row tiles inside the full rounds
      if (rows > 0 && rows < a.M) split_row = rows;
    }
  }
  auto run = [&](int c, long long r0, long long r1) {
    switch (nsplit) {
      case 1: return launch_bf_ns<1>(a, epi, c, st, r0, r1);
      case 2: return launch_bf_ns<2>(a, epi, c, st, r0, r1);
      case 3: return launch_bf_ns<3>(a, epi, c, st, r0, r1);
      default: return (int)PFK_ERR_BAD_ARG;
    }
  };
  // Round 4: on the eight-wave tiles the weight planes (three quarters of a K sub-step's bytes, nothing to split) go global -> LDS
  // by `buffer_load_dwordx4 ... lds` (StagerBF<..., BDMA>): 6 of a thread's 9 ds_write_b128 and 24 staging registers per step
  // disappear.  Same results (the K order does not change); batch 8, three planes (gpurun_out/r4n_conv_x6.log, 3 rounds, us):
  // fh|mask conv1 319.3 -> 312.1, z|r 200.4 -> 195.8, convc2 287.4 -> 279.7 on 128x256; 323.7 -> 318.3, mask conv2 118.7 -> 117.0
  // on 256x128; the 128x128 tail tile 2-9 %.  Whole forward, batch 8 (gpurun_out/r4o_legs.log, interleaved): bf16x6 102.4 -> 104.8
  // pairs/s (+2.3 %).  Two planes gain 2-8 % per launch in the micro-benchmark but the bf16x3 forward loses 0.9 % (145.0 -> 143.7),
  // one plane is a wash: three planes only.  pfk_debug_set_tile(170 + t) keeps the register path (A/B), 180 + t forces the DMA
  // path (two planes too).
  const int dma = (((nsplit == 3 && g_bf_cfg / 10 != 7) || (nsplit >= 2 && g_bf_cfg / 10 == 8)) && cfg >= 4 && cfg <= 6) ? 10 : 0;
  if (split_row > 0) {
    const int rc = run(cfg + dma, 0, split_row);
    return rc != PFK_OK ? rc : run(4 + dma, split_row, -1);
  }
  return run(cfg + dma, 0, -1);
}

}  // namespace pfkg
