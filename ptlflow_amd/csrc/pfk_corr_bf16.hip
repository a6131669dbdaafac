// K1 in bf16: all-pairs correlation with bf16 operands and a bf16 volume (what the reference's matmul yields under
// torch.autocast(bfloat16)).  HBM-write-bound by construction — 2 B per volume element, 106 MB per 55x128
// pair against 25.4 GFLOP on a ~2.5 PFLOP/s pipe — so the kernel is organised around the store:
//
//   * 128 x 128 output tile per 256-thread block (four waves as 2 x 2, wave tile 64 x 64 = 2 x 2 MFMA 32x32x16 blocks, fp32 acc);
//   * K walked in 64-channel chunks: both operand panels (128 rows x 128 B each) go global -> VGPR -> LDS as 16-byte pieces,
//     LDS rows of 64 B with the 16-byte chunk index XOR-ed by (row >> 2) & 3 (the layout of pfk_gemm_bf.hip: conflict-free for
//     the ds_write_b128 row pairs and for the fragment ds_read_b128 lane groups);
//   * epilogue: acc * scale -> bf16 (round to nearest even) -> LDS as a [128][128] bf16 tile -> 16-byte row-contiguous global
//     stores (a wave writes 4 rows x 256 B per instruction), instead of 2-byte scattered stores from the MFMA register layout;
//   * 32 KB of LDS and ~100 VGPRs per block: four resident blocks per CU — while one stores, the others load and multiply;
//   * tiles walked in 16 x 16 supertiles so both operand panels of a supertile stay in the XCD's L2 (pfk_gemm.h tile_of).
#include "pfk_gemm.h"

using namespace pfkg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 128, TN = 128, KC = 64;          // tile and K chunk (channels)
constexpr int ROWB = 64;                            // LDS row: 32 bf16
constexpr int PLANE = 128 * ROWB;                   // one 32-channel sub-block of one operand: 8 KB
constexpr int STAGE = 4 * PLANE;                    // A sub 0, A sub 1, B sub 0, B sub 1: 32 KB (also holds the 128x128 bf16 output tile)

struct CorrBfArgs {
  const __bf16* a; const __bf16* b; __bf16* out;
  int N1, N2, D, lda, ldb;
  float scale;
  long long a_bs, b_bs, o_bs;
  int tiles_m, tiles_n, supertile;
};

__global__ __launch_bounds__(256, 4) void corr_bf16_kernel(const CorrBfArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[STAGE];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  const int wm0 = (wid_s >> 1) * 64, wn0 = (wid_s & 1) * 64;
  const int bid = pfk_xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  tile_of(bid, g.tiles_m, g.tiles_n, g.supertile, tile_m, tile_n);
  const int m0 = tile_m * TM, n0 = tile_n * TN;
  const long long batch = blockIdx.y;
  const __amdgpu_buffer_rsrc_t rsa = make_rsrc(g.a + batch * g.a_bs), rsb = make_rsrc(g.b + batch * g.b_bs);

  // staging role: 16-byte piece c16 (of the chunk's 128 B) of rows r0 + 32 * i, both operands
  const int c16 = t & 7, r0 = t >> 3;
  const int sub = c16 >> 2, ch = c16 & 3;
  // rows r0 + 32 i: offsets are the row-0 offset + i * (32 rows), the LDS swizzle key ((row >> 2) & 3) does not depend on i
  const unsigned aoff0 = (unsigned)(((m0 + r0) * g.lda + c16 * 8) * 2), a32 = (unsigned)(32 * g.lda * 2);
  const unsigned boff0 = (unsigned)(((n0 + r0) * g.ldb + c16 * 8) * 2), b32 = (unsigned)(32 * g.ldb * 2);
  const unsigned soff0 = (unsigned)(sub * PLANE + r0 * ROWB + ((ch ^ ((r0 >> 2) & 3)) << 4));
  const int frow = lane & 31, hl = lane >> 5;
  const int key = (frow >> 2) & 3;     // wave / MFMA-block row offsets are multiples of 32: same key
  const int ko0 = ((0 + hl) ^ key) << 4, ko1 = ((2 + hl) ^ key) << 4;
  const int a_row = (wm0 + frow) * ROWB, b_row = 2 * PLANE + (wn0 + frow) * ROWB;

  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int chunks = (g.D + KC - 1) / KC;
  u32x4 ra[4], rb[4];
  // chunk 0 loads; channels past D read as zeros (the buffer offset of a partial last chunk is masked per piece)
  auto load = [&](int c) {
    const bool ok = c * KC + c16 * 8 < g.D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsa, (ok && m0 + r0 + 32 * i < g.N1) ? aoff0 + i * a32 : OOB, c * KC * 2, 0);
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (ok && n0 + r0 + 32 * i < g.N2) ? boff0 + i * b32 : OOB, c * KC * 2, 0);
    }
  };
  load(0);
  for (int c = 0; c < chunks; ++c) {
    __syncthreads();                       // the previous chunk's fragments have been read
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(smem + soff0 + i * 32 * ROWB) = ra[i];
      *reinterpret_cast<u32x4*>(smem + 2 * PLANE + soff0 + i * 32 * ROWB) = rb[i];
    }
    if (c + 1 < chunks) load(c + 1);       // in flight during this chunk's MFMAs
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const char* base = smem + s * PLANE + (kb ? ko1 : ko0);
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(base + a_row + mt * 32 * ROWB);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) fb[nt] = *reinterpret_cast<const bf16x8*>(base + b_row + nt * 32 * ROWB);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
      }
  }
  __syncthreads();                         // operand stage is dead: reuse it for the output tile [128][128] bf16 (256-byte rows)
  // The thread's coordinates are re-derived here from the wave id (an SGPR since the top) and the hardware lane count rather than
  // kept live across the K loop: at 128 VGPRs (four blocks per CU) the loop otherwise spills them to scratch, and scratch stores
  // are HBM writes (12 B per thread showed up as +9 % on the write counter of a kernel that is write-bound).
  const int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int t2 = wid_s * 64 + lane2;
  const int frow2 = lane2 & 31, hl2 = lane2 >> 5;
  const int wm2 = (wid_s >> 1) * 64, wn2 = (wid_s & 1) * 64;
  // D layout of a 32x32 block: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  __bf16* tile = reinterpret_cast<__bf16*>(smem);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm2 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl2;
        const int col = wn2 + nt * 32 + frow2;
        // 16-byte chunk index XOR-ed with the row: the 32 lanes of a half-wave write one row's 64 contiguous bytes either way,
        // and the row-wise 16-byte reads below spread over all banks
        tile[row * 128 + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7))] = (__bf16)(acc[mt][nt][r] * g.scale);
      }
  __syncthreads();
  __bf16* outp = g.out + batch * g.o_bs;
  const bool vec = (g.N2 & 7) == 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int piece = t2 + 256 * i;            // 2048 pieces of 8 bf16: row = piece / 16, 16-byte column chunk = piece % 16
    const int row = piece >> 4, cc = piece & 15;
    const long long grow = (long long)m0 + row;
    const int gcol = n0 + cc * 8;
    if (grow >= g.N1 || gcol >= g.N2) continue;
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * 256 + ((cc ^ (row & 15)) << 4));
    __bf16* dst = outp + grow * g.N2 + gcol;
    if (vec && gcol + 8 <= g.N2) {
      *reinterpret_cast<u32x4*>(dst) = v;
    } else {
      unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);   // constant indices after unrolling: v stays in registers
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (gcol + k < g.N2) d16[k] = (unsigned short)(v[k >> 1] >> (16 * (k & 1)));
    }
  }
}

}  // namespace

extern "C" {

// a_bf16 [B][N1][lda], b_bf16 [B][N2][ldb] bf16 rows (lda, ldb multiples of 8 elements, >= D), out_bf16 [B][N1][N2]
int pfk_corr_volume_bf16(const void* f1_bf16, int ld1, const void* f2_bf16, int ld2, void* out_bf16, int B, int N1, int N2, int D,
                         float scale, pfk_stream_t stream) {
  if (!f1_bf16 || !f2_bf16 || !out_bf16 || B <= 0 || N1 <= 0 || N2 <= 0 || D <= 0) return PFK_ERR_BAD_ARG;
  if (ld1 < D || ld2 < D) return PFK_ERR_BAD_ARG;
  if (!pfk_aligned16(f1_bf16) || !pfk_aligned16(f2_bf16) || !pfk_aligned16(out_bf16) || (ld1 & 7) || (ld2 & 7) || (D & 7))
    return PFK_ERR_ALIGNMENT;
  if ((long long)N1 * ld1 * 2 >= 0x7fffffffLL || (long long)N2 * ld2 * 2 >= 0x7fffffffLL) return PFK_ERR_UNSUPPORTED;
  CorrBfArgs g{};
  g.a = static_cast<const __bf16*>(f1_bf16); g.b = static_cast<const __bf16*>(f2_bf16); g.out = static_cast<__bf16*>(out_bf16);
  g.N1 = N1; g.N2 = N2; g.D = D; g.lda = ld1; g.ldb = ld2; g.scale = scale;
  g.a_bs = (long long)N1 * ld1; g.b_bs = (long long)N2 * ld2; g.o_bs = (long long)N1 * N2;
  g.tiles_m = (N1 + TM - 1) / TM; g.tiles_n = (N2 + TN - 1) / TN;
  g.supertile = (g.tiles_m >= 32 && g.tiles_n >= 32) ? 16 : 0;
  const long long nblk = (long long)g.tiles_m * g.tiles_n;
  if (nblk > 0x7fffffffLL || B > 65535) return PFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(corr_bf16_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  return pfk_launch_status();
}

}  // extern "C"
