/*
 * pfk.h — C ABI of libpfk.so: hand-written gfx950 (MI355X / CDNA4) kernels for the RAFT-family
 * optical-flow hot path of hmorimitsu/ptlflow.
 *
 * Every entry point is `extern "C"`, takes raw *device* pointers + explicit sizes + a HIP stream,
 * allocates nothing, never synchronises, never throws; it returns 0 on success or a negative
 * pfk_status code (pfk_status_string() names it).  No torch types cross this boundary: the torch
 * extension (ptlflow_amd/csrc/pfk_torch.cpp), the ctypes tests and any other host language bind
 * exactly these symbols.
 *
 * Activation layout is *pixel-major* ("NHWC"): a feature map is a row-major matrix
 * [B*H*W pixels][ld floats] whose row p = (b*H + y)*W + x holds that pixel's channels
 * contiguously; `ld >= channels` lets several producers write channel slices of one buffer
 * (that is how the reference's torch.cat calls disappear).  All pointers must be 16-byte aligned
 * and every `ld` / channel count / channel offset a multiple of 4 floats.
 *
 * Reference interface each entry replaces (paths under hmorimitsu/ptlflow):
 *   pfk_corr_volume_f32   ptlflow/models/raft/corr.py:56-64   CorrBlock.corr (matmul / sqrt(D));
 *                         sea_raft/corr.py:109-117 (same op, per level)
 *   pfk_corr_pool2x2_f32  ptlflow/models/raft/corr.py:25-27   F.avg_pool2d(corr, 2, stride=2)
 *   pfk_corr_lookup_f32   ptlflow/models/raft/corr.py:29-54   CorrBlock.__call__ and
 *                         ptlflow/models/raft/utils.py:67-75  bilinear_sampler -> F.grid_sample
 *   pfk_conv2d_f32        ptlflow/models/raft/update.py:6-153 every nn.Conv2d of the update block
 *                         (+ the fused sigmoid/tanh/GRU-blend/relu epilogues of :58-73, :24-32,
 *                         :104-112, :144-153); torch.cat at :60,63,67,70,109,112,146 is replaced by
 *                         channel-slice addressing
 *   pfk_conv_cin2_f32     ptlflow/models/raft/update.py:100,107 (convf1: 7x7 conv on the 2-ch flow)
 *   pfk_flow_delta_f32    ptlflow/models/raft/update.py:10,14 (FlowHead.conv2) fused with
 *                         ptlflow/models/raft/raft.py:174,178 (flow = coords1-coords0; coords1 += d)
 *   pfk_convex_upsample_f32  ptlflow/models/raft/raft.py:112-123 RAFT.upsample_flow
 *   pfk_altcorr_forward_f32  ptlflow/utils/external/alt_cuda_corr/correlation_kernel.cu:258-285 (alt_cuda_corr.forward)
 *   pfk_corr_lookup_bwd_f32 / pfk_corr_volume_bwd_f32 / pfk_convex_upsample_bwd_f32
 *                         what torch.autograd runs for corr.py:29-64 and raft.py:112-123 in a training step (train.py;
 *                         gradients reach fmap1 / fmap2 only, `coords` is detached at raft.py:171)
 * The native plug-in precedent in the reference is alt_cuda_corr
 * (ptlflow/utils/external/alt_cuda_corr/correlation.cpp:23-54: pybind forward/backward on raw
 * contiguous CUDA tensors); this header is the same kind of boundary, minus torch.
 */
#ifndef PFK_H
#define PFK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pfk_stream_t; /* a hipStream_t */

enum pfk_status {
  PFK_OK = 0,
  PFK_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, unsupported combination */
  PFK_ERR_ALIGNMENT = -2,    /* pointer not 16-byte aligned or ld/channels not a multiple of 4 */
  PFK_ERR_UNSUPPORTED = -3,  /* shape outside what the kernels were built for */
  PFK_ERR_LAUNCH = -4,       /* hipLaunchKernel reported an error */
  PFK_ERR_DISABLED = -5      /* a pfk_debug_set_* knob called in a process that did not opt in with PFK_DEBUG_KNOBS=1 */
};

#define PFK_MAX_LEVELS 8
#define PFK_ABI_VERSION 7

int pfk_abi_version(void);
const char* pfk_status_string(int status);
/* sha256 (first 16 hex digits) of the sources this library was built from (every file of csrc/, include/pfk.h, the compiler flags), stamped by
 * ptlflow_amd/_build.py; ptlflow_amd.load_native() refuses a library whose stamp differs from the tree it sits in. */
const char* pfk_source_hash(void);
/* The pfk_debug_set_* knobs change PROCESS-GLOBAL kernel selection (tests and tuning scripts force every tile shape / kernel
 * variant through them).  They are inert — return PFK_ERR_DISABLED and change nothing — unless the process environment has
 * PFK_DEBUG_KNOBS=1 at the first call, so a production caller cannot flip tile state by accident.  Not thread-safe.
 * tuning/debug knob: force the implicit-GEMM tile configuration (-1 = heuristic). */
int pfk_debug_set_tile(int cfg);
/* test hook: n / d through the multiplier arithmetic the persistent convolution kernel decodes its tiles with (n < 2^31, d >= 1) */
unsigned pfk_debug_fastdiv(unsigned n, unsigned d);
/* tuning knob of the pyramid lookup: source pixels per workgroup — 4 (default: 4 on the row-major layout; on the blocked layout 4
 * below 28 160 source pixels and 8 from there up), 8, 104 (always 4), 14 (4 with cross-lane tap reads, row-major only). */
int pfk_debug_set_lookup_pix(int pix);
int pfk_debug_set_altcorr(int mode);       /* on-demand correlation forward: 0 = heuristic, 1 = per-pixel kernel, 2 / 3 = window-sharing MFMA kernel on 8x4 / 8x8 patches */
int pfk_debug_set_wgrad(int variant);      /* weight-gradient tile height: 0 = by padding waste, 1 / 2 / 4 = forced 32 / 64 / 128 rows (tuning knob) */

/* ---- K1: all-pairs correlation --------------------------------------------------------------
 * out[b][i][j] = scale * sum_d f1[b][i][d] * f2[b][j][d]        (fp32 MFMA, exact fp32 products)
 * f1 [B][N1][ld1], f2 [B][N2][ld2] pixel-major, D channels used; out [B][N1][N2] row-major,
 * i.e. for source pixel i a full [h2][w2] map — the reference's [B*N,1,h2,w2] volume. */
int pfk_corr_volume_f32(const float* f1, int ld1, const float* f2, int ld2, float* out,
                        int B, int N1, int N2, int D, float scale, pfk_stream_t stream);

/* bf16 operands, bf16 volume — what the reference's matmul produces under torch.autocast(bfloat16):
 * f1_bf16 [B][N1][ld1], f2_bf16 [B][N2][ld2] bf16 rows (ld % 8 == 0, D % 8 == 0), out_bf16 [B][N1][N2] bf16; fp32 accumulate
 * on the bf16 matrix cores, scaled, rounded to nearest even.  HBM-write-bound (2 B per volume element). */
int pfk_corr_volume_bf16(const void* f1_bf16, int ld1, const void* f2_bf16, int ld2, void* out_bf16, int B, int N1, int N2, int D,
                         float scale, pfk_stream_t stream);

/* ---- K2: 2x2/stride-2 average pool over the target dims (floor) ------------------------------
 * in [M][H][W] -> out [M][H/2][W/2], value ((a00+a01)+a10)+a11) * 0.25 (torch's CPU order). */
int pfk_corr_pool2x2_f32(const float* in, float* out, int64_t M, int H, int W,
                         pfk_stream_t stream);
/* same on bf16 maps: fp32 accumulation, one rounding to bf16 (torch's avg_pool2d on bf16 tensors) */
int pfk_corr_pool2x2_bf16(const void* in, void* out, int64_t M, int H, int W, pfk_stream_t stream);
/* 2x2 average of a pixel-major FEATURE map [B][H][W][in_ld] -> [B][H/2][W/2][out_ld] (C channels): bit-identical to
 * F.interpolate(fmap2, scale_factor=0.5, mode="bilinear", align_corners=False), SEA-RAFT's pyramid step
 * (sea_raft/corr.py:81-83), and to F.avg_pool2d(fmap2, 2, 2) (AlternateCorrBlock, raft/corr.py:72-74). */
int pfk_fmap_pool2x2_f32(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, pfk_stream_t stream);

/* ---- K3: radius-r bilinear lookup over all pyramid levels in one launch ----------------------
 * levels[l] is the level-l volume [B*N][lvl_h[l]][lvl_w[l]]; coords is [B][2][h][w] (x then y,
 * pixel units, fp32, NCHW as the reference passes it).  Output: out[p*out_ld + l*n*n + i*n + j]
 * with n = 2r+1, sample (i, j) taken at (x/2^l + i - r, y/2^l + j - r) — the reference's
 * x-offset-major window — through grid_sample's exact fp32 normalise/un-normalise round trip,
 * zero padding per tap, FMA accumulation order of torch's CPU kernel (bit-exact values for the
 * same pyramid).  NaN / inf coordinates and 1-pixel levels give NaN exactly like the reference. */
typedef struct {
  const void* levels[PFK_MAX_LEVELS];   /* float maps (pfk_corr_lookup_f32) or bf16 maps (pfk_corr_lookup_bf16) */
  int lvl_h[PFK_MAX_LEVELS];
  int lvl_w[PFK_MAX_LEVELS];
  int num_levels;
  int radius;      /* 1..4 */
  int B, h, w;     /* source grid: N = h*w pixels per batch element */
  const float* coords;
  void* out;       /* fp32 rows, or bf16 rows when out_bf16 != 0 */
  int out_ld;      /* in elements; >= num_levels*(2r+1)^2, multiple of 4 */
  int out_bf16;    /* ABI 7: 0 = `out` holds fp32, 1 = bf16 (each sample rounded to nearest even: what autocast hands the next
                      convolution; columns the lookup does not write — the pad up to out_ld — are left untouched) */
} pfk_lookup_desc;
int pfk_corr_lookup_f32(const pfk_lookup_desc* d, pfk_stream_t stream);
/* same with `levels` pointing to bf16 maps; arithmetic and output fp32 (grid_sample is an fp32 op under autocast) */
int pfk_corr_lookup_bf16(const pfk_lookup_desc* d, pfk_stream_t stream);

/* ---- K1-K3 on the BLOCKED volume layout (round 5; inference path) -------------------------------------------------------
 * A level's [H][W] map per source pixel stored as ceil(H/4) x ceil(W/8) tiles of 4 rows x 8 columns, 32 consecutive elements per
 * tile (one 128-byte line in fp32): element (y, x) of a map lives at ((y/4 * ceil(W/8) + x/8) * 32 + (y%4) * 8 + x%8);
 * pfk_blocked_map_elems(H, W) elements per source pixel (pad elements of edge tiles are zero).  The 12 x 12 window a lookup
 * stages covers ~9 lines instead of ~16 (raft/corr.py:29-54 reads the same taps either way).
 *   pfk_fmap_to_blocked_f32      permutes the rows of the target feature map [B][H*W][in_ld] into that order (zero rows for pad
 *                                elements): out [B][pfk_blocked_map_elems(H, W)][out_ld].  pfk_corr_volume_f32 / _bf16 run against
 *                                it (N2 = pfk_blocked_map_elems) write level 0 directly in the blocked layout (raft/corr.py:56-64).
 *   pfk_corr_pool2x2_blocked_*   K2 blocked -> blocked (raft/corr.py:25-27), the same ((a00+a01)+a10)+a11)*0.25 per element.
 *   pfk_corr_lookup_blocked_*    K3 on blocked levels: the same pfk_lookup_desc (lvl_h / lvl_w = the maps' LOGICAL sizes), the same
 *                                arithmetic, bit-identical output to pfk_corr_lookup_* on the row-major pyramid. */
int64_t pfk_blocked_map_elems(int H, int W);
int pfk_fmap_to_blocked_f32(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, pfk_stream_t stream);
int pfk_corr_pool2x2_blocked_f32(const float* in, float* out, int64_t M, int H, int W, pfk_stream_t stream);
int pfk_corr_pool2x2_blocked_bf16(const void* in, void* out, int64_t M, int H, int W, pfk_stream_t stream);
int pfk_corr_lookup_blocked_f32(const pfk_lookup_desc* d, pfk_stream_t stream);
int pfk_corr_lookup_blocked_bf16(const pfk_lookup_desc* d, pfk_stream_t stream);

/* ---- backward of K3 (training): scatter d(out) through the forward's four bilinear weights --------------------
 * grad_out [B*N][grad_out_ld] (the layout of pfk_corr_lookup_f32's `out`); grad_levels[l] is the gradient of the level-l
 * volume, one [lvl_h][lvl_w] map per source pixel at row stride lvl_ld[l] floats (>= lvl_h*lvl_w; a stride padded to a
 * multiple of 4 makes the buffer a legal operand of pfk_corr_volume_bwd_f32).  ACCUMULATES (+=): the caller zeroes the
 * buffers once per training step and every recurrent iteration's lookup adds into them.  Deterministic (no atomics).
 * Taps that fell outside the map in the forward (zero padding) receive nothing. */
typedef struct {
  float* grad_levels[PFK_MAX_LEVELS];
  int lvl_h[PFK_MAX_LEVELS];
  int lvl_w[PFK_MAX_LEVELS];
  long long lvl_ld[PFK_MAX_LEVELS];
  int num_levels;
  int radius;      /* 1..4 */
  int B, h, w;
  const float* coords;    /* [B][2][h][w], the coordinates the forward was called with */
  const float* grad_out;
  int grad_out_ld;
} pfk_lookup_bwd_desc;
int pfk_corr_lookup_bwd_f32(const pfk_lookup_bwd_desc* d, pfk_stream_t stream);

/* ---- backward of K1 for one batch element and one pyramid level (training) -----------------------------------
 * The level-l volume is C = scale * F1 . F2_l^T with F2_l the (avg-pooled / bilinearly halved) target feature map, so
 *   df1[i][d]  = (accumulate_df1 ? df1[i][d] : 0) + scale * sum_j dC[i][j] * f2[j][d]         i < N1   (fp32 MFMA GEMM)
 *   df2[j][d]  =                                       scale * sum_i dC[i][j] * f1[i][d]         j < ldc  (transposed product)
 * dC [N1][ldc]: ldc % 4 == 0, columns [N2, ldc) zero.  f1 [N1][ld1] pixel-major.  f2_cm: F2_l CHANNEL-major [D][ld2cm],
 * ld2cm == round_up(ldc, 32), columns >= N2 zero (pfk_pm_to_cm_f32 into a zeroed buffer).  df1 [N1][df1_ld];
 * df2 [ldc][round_up(D, 32)] pixel-major (rows >= N2 and columns >= D come out zero).
 * workspace: pfk_corr_volume_bwd_workspace_bytes(N1, ldc, D) bytes, 16-byte aligned (may be NULL when that is 0). */
long long pfk_corr_volume_bwd_workspace_bytes(int N1, int ldc, int D);
int pfk_corr_volume_bwd_f32(const float* dC, int ldc, int N1, int N2, const float* f1, int ld1, const float* f2_cm, int ld2cm,
                            int D, float scale, float* df1, int df1_ld, int accumulate_df1, float* df2, void* workspace,
                            long long workspace_bytes, pfk_stream_t stream);

/* ---- backward of the convex upsampling (raft/raft.py:112-123) -------------------------------------------------
 * flow: NCHW [B][2][H][W] (flow_ld == 0) or pixel-major rows flow[p*flow_ld + 0..1]; mask pixel-major [M][mask_ld] as in
 * pfk_convex_upsample_f32; grad_out [B][2][8H][8W] -> grad_mask pixel-major [M][grad_mask_ld] (576 channels written),
 * grad_flow [B][2][H][W].  workspace: pfk_convex_upsample_bwd_workspace_bytes(B, H, W).  Deterministic. */
long long pfk_convex_upsample_bwd_workspace_bytes(int B, int H, int W);
int pfk_convex_upsample_bwd_f32(const float* flow, int flow_ld, const float* mask, int mask_ld, const float* grad_out,
                                float* grad_mask, int grad_mask_ld, float* grad_flow, void* workspace, long long workspace_bytes,
                                int B, int H, int W, pfk_stream_t stream);

/* ---- K4-K6: "same"-padded convolution (stride 1, or s for the encoders) as an implicit GEMM on fp32 MFMA ----
 * out[p][co] = epilogue( bias[co] + sum_{s, ky, kx, c} src[s][p + (ky-kh/2)*W + (kx-kw/2)][c]
 *                                                    * weight[co][k(s,ky,kx,c)] )
 * Input channels may come from up to 3 pixel-major sources (the reference's torch.cat operands).
 * Packed weight: row-major [cout][ktot]; k enumerates, for each source s in order, each tap
 * (ky major, kx minor), the source's channels padded up to a multiple of 32 (pad weights = 0):
 * ktot = sum_s kh*kw*round_up(channels_s, 32).  pfk_conv_ktot() returns it. */
enum pfk_epilogue {
  PFK_EPI_LINEAR = 0,  /* v = (acc+bias) ; relu? ; v *= scale ; [v = residual[p*residual_ld+co] + v] ;
                          out[p*out_ld + out_coff + co] = v */
  PFK_EPI_GRU_ZR = 1,  /* cout = 2*Ch: v = acc + bias [+ residual[p*residual_ld+co]]; co<Ch: z=sigmoid(v) -> aux_z[p*Ch+co];
                          co>=Ch: r=sigmoid(v) -> aux_rh[p*Ch + co-Ch] = r * h[p*h_ld + co-Ch] */
  PFK_EPI_GRU_Q = 2    /* cout = Ch: q=tanh(acc + bias [+ residual]); h[p*h_ld+co] = (1-z)*h + z*q, z = aux_z[p*Ch+co] */
};

typedef struct {
  const float* ptr;
  int ld;        /* row stride in floats */
  int channels;  /* channels read from each row (multiple of 4) */
} pfk_conv_src;

typedef struct {
  pfk_conv_src src[3];
  int num_src;
  int B, H, W;
  int kh, kw;            /* odd; padding kh/2, kw/2 */
  const float* weight;   /* packed [cout][ktot] */
  const float* bias;     /* [cout] or NULL */
  int cout;
  int epilogue;          /* enum pfk_epilogue */
  int relu;              /* LINEAR only */
  float scale;           /* LINEAR only (1.0f for none) */
  float* out;            /* LINEAR only */
  int out_ld, out_coff;
  float* h;              /* GRU_ZR (read) / GRU_Q (read+write): hidden state slice */
  int h_ld;
  float* aux_z;          /* [M][Ch] */
  float* aux_rh;         /* [M][Ch] */
  const float* residual; /* optional [M][cout] rows.  LINEAR: added after relu/scale (GMA's `fmap + gamma*out`, gma_utils.py:111).
                            GRU_ZR / GRU_Q (ABI 5): added to the gate PRE-activation — the loop-invariant part of a gate's
                            convolution: in raft/update.py:60-71 the GRU input is cat([h, inp, motion]) and `inp` (the context
                            features, raft.py:158-160) does not change over the iterations, so conv(W[:, inp slice], inp) + bias is
                            computed once per forward and handed in here while the per-iteration convolution covers the h and
                            motion channels only (16-byte aligned, residual_ld % 4 == 0) */
  int residual_ld;
  int stride;            /* 0 or 1: "same" convolution; s > 1: output (yo, xo) reads input (yo*s + ky - kh/2, xo*s + kx - kw/2),
                            Ho = (H-1)/s + 1, Wo = (W-1)/s + 1 (= PyTorch's Conv2d(k, stride=s, padding=k/2)); H, W are the
                            INPUT dims, sources have B*H*W rows, out / h / aux / residual have B*Ho*Wo rows */
  int relu_after_residual; /* LINEAR only: relu once more after the residual add (ResidualBlock, raft/extractor.py:53-61) */
  void* workspace;       /* optional: >= pfk_conv_workspace_bytes() of device memory, 16-byte aligned, private to
                            the stream, ZERO-FILLED by the caller once after allocation (the kernels leave its flag region
                            zeroed); enables the stream-K schedule for small grids (deterministic).  NULL: tile grid */
  long long workspace_bytes;
  int cout_active;       /* ABI 6, LINEAR only: 0 or >= cout = every output channel; otherwise only output channels [0, cout_active)
                            (a multiple of 64) are computed and written, the rest of `out` is left untouched — with EXACTLY the bits
                            the full launch gives those channels: the tile schedule (tile grid or stream-K, and the stream-K split
                            points) is chosen for the full `cout` and the column tiles past cout_active are skipped.  The fused
                            flow-head | mask-head convolution (raft/update.py:13, :138-139) on the iterations whose mask nobody
                            reads (raft/raft.py:180-192 in eval). */
  int cout_split;        /* ABI 6, LINEAR only, 0 = none: the output channels form two groups [0, cout_split) and [cout_split, cout)
                            of equal width (a multiple of 64; 2 * cout_split == cout) and the stream-K schedule walks their column
                            tiles INTERLEAVED (group 0's first tile, group 1's first tile, group 0's second ...), so that a launch
                            with cout_active == cout_split still gives every block about half of its range to do.  Pass the same
                            value in the full launch and in the cout_active launch: the schedule, hence the bits, are then the
                            same.  Tile-grid schedules ignore it. */
} pfk_conv_desc;

long long pfk_conv_workspace_bytes(void);
/* Byte offset, inside a conv workspace, of a 32-bit counter of stream-K fix-ups that timed out (a partner block's partial
 * tile never became visible within the bounded spin).  The caller zero-initialises the workspace once; the kernels only
 * ever increment the word, and the affected output tile is written as NaN.  0 = every result so far is complete; after a
 * non-zero reading zero-fill the workspace again before the next launch (flags of the lost fix-up may be left set). */
long long pfk_conv_workspace_fault_offset(void);
int pfk_conv_ktot(const pfk_conv_desc* d);
int pfk_conv2d_f32(const pfk_conv_desc* d, pfk_stream_t stream);

/* Grouped launch (ABI 7): `n` (1..PFK_CONV_GROUP_MAX) INDEPENDENT convolutions — no output of one is a source, residual or weight of
 * another — in ONE grid of 64 x 64 tiles; every descriptor has the LINEAR epilogue.  Results are those of `n` calls of
 * pfk_conv2d_f32 without a workspace, bit for bit (a tile's K order does not depend on what runs beside it).  For the small-batch
 * regime, where the update block's launches leave most CUs idle: convc1 | convf2 | the previous iteration's mask conv2
 * (ptlflow/models/raft/update.py:105-108, :152 — `cor` and `flo` only meet in `conv`, the mask head reads `net` only).
 * `workspace`, `cout_active`, `cout_split` of the descriptors: ignored / unsupported (PFK_ERR_UNSUPPORTED). */
#define PFK_CONV_GROUP_MAX 4
int pfk_conv2d_group_f32(const pfk_conv_desc* descs, int n, pfk_stream_t stream);

/* Same convolution on the bf16 matrix cores with split operands (fp32 in HBM, fp32 accumulate, fp32 epilogue).
 * Every fp32 operand x is replaced by the sum of its first `nsplit` bf16 planes x0 = bf16(x), x1 = bf16(x - x0),
 * x2 = bf16(x - x0 - x1); a product keeps the terms a_i*b_j with i + j < nsplit:
 *   nsplit 1: plain bf16 operands (what the reference computes under `--fp16` autocast with bf16,
 *             model_benchmark.py:311 / base_model.py autocast), nsplit 2: ~2^-17 relative product error,
 *   nsplit 3: fp32-grade products at 6 bf16 MFMAs per 16 channels.
 * Activations are split by the kernel while staging; the caller pre-splits the weight:
 * weight_planes = bf16 [nsplit][cout][ktot], ktot = pfk_conv_ktot_bf16(d) (= pfk_conv_ktot(d): the fp32 packed
 * weight's k order and padding, split plane by plane).  d->weight, d->workspace are ignored. */
int pfk_conv_ktot_bf16(const pfk_conv_desc* d);
int pfk_conv2d_bf16s(const pfk_conv_desc* d, const void* weight_planes, int nsplit, pfk_stream_t stream);

/* ---- K8b (ABI 7): the same convolution with bf16 ACTIVATION STORAGE on the bf16 matrix cores ---------------------------
 * What the reference's convolutions compute under its reduced-precision switch (scripts/model_benchmark.py:317-319,
 * validate.py:243-244 / torch.autocast(bfloat16): every nn.Conv2d of raft/update.py:6-153 reads and writes 16-bit tensors).
 * Sources are bf16 pixel-major rows (`ld`, `channels` in ELEMENTS, multiples of 8; 16-byte aligned), the weight is bf16
 * [cout][ktot] with the K order of pfk_conv2d_f32's packed weight but 64-channel chunks: for each source, each tap (ky major), the
 * source's channels zero-padded to a multiple of 64 — ktot = pfk_conv_ktot_b16(d).  Both operands go global -> LDS by LDS-DMA
 * (no register round trip, no conversion); products on v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 bias / residual / gate
 * arithmetic.  Outputs:
 *   LINEAR   out = bf16 (out_bf16 != 0; the next convolution's operand) or fp32 rows, `out_ld` / `out_coff` in elements of that type
 *   GRU_ZR   z -> aux_z BF16 [M][Ch];  r * h (h read from its bf16 copy h_b16) -> aux_rh BF16 [M][Ch] (the q convolution's operand)
 *   GRU_Q    h fp32 updated in place (the recurrent state keeps full precision; z read from aux_z bf16) and, when h_b16 != NULL,
 *            its bf16 rounding -> h_b16[p * h_b16_ld + co] (the next convolutions' operand)
 * `residual`: LINEAR — fp32 rows added after relu / scale (pfk_conv_desc); GRU epilogues — BF16 rows [M][cout] added to the gate
 * pre-activation (the loop-invariant context term: at batch 8 these launches are bound by the HBM bytes of their epilogues, and the
 * term is a convolution output that autocast would hold in 16 bits anyway).
 * kh * kw <= 32.  No workspace, no stream-K: tile grids only.
 * Alignment: sources / weight 16 bytes.  The GRU epilogues move their bf16 operands and results as 16-byte row pieces: Ch % 8 == 0,
 * h_b16 / aux_z / aux_rh / the bf16 `residual` 16-byte aligned with row strides that are multiples of 8 elements (h: 16 bytes, h_ld % 4).
 * LINEAR: `out` 8 bytes (bf16) / 16 bytes (fp32), out_ld / out_coff multiples of 4; bf16 rows without a residual whose pieces are 16-byte
 * aligned (out 16 bytes, out_ld and out_coff multiples of 8) are stored with 16-byte accesses. */
typedef struct {
  const void* ptr;   /* bf16 rows */
  int ld;            /* row stride in elements (multiple of 8) */
  int channels;      /* channels read from each row (multiple of 8) */
} pfk_conv_src_b16;

typedef struct {
  pfk_conv_src_b16 src[3];
  int num_src;
  int B, H, W;
  int kh, kw;
  const void* weight;    /* bf16 [cout][ktot] */
  const float* bias;     /* fp32 [cout] or NULL */
  int cout;
  int epilogue;          /* enum pfk_epilogue */
  int relu;
  float scale;
  void* out;             /* LINEAR */
  int out_ld, out_coff;
  int out_bf16;          /* LINEAR: 1 = `out` holds bf16 elements, 0 = fp32 */
  float* h;              /* GRU_Q (read + write): fp32 hidden state */
  int h_ld;
  void* h_b16;           /* GRU_ZR (read, required): bf16 copy of the hidden state; GRU_Q (write, optional): bf16 copy of the new one */
  int h_b16_ld;
  void* aux_z;           /* bf16 [M][Ch]: written by GRU_ZR, read by GRU_Q */
  void* aux_rh;          /* bf16 [M][Ch] */
  const void* residual;  /* LINEAR: fp32 [M][cout] rows; GRU_ZR / GRU_Q: bf16 [M][cout] rows (see above) */
  int residual_ld;       /* in elements of that type */
  int stride;
  int relu_after_residual;
  int residual_bf16;     /* LINEAR: 1 = `residual` holds bf16 rows (cout % 4 == 0), 0 = fp32 */
  /* batched GEMM — `batches` > 1 independent problems of one shape in one grid (LINEAR, one source, 1 x 1, B == 1: M = H*W rows
   * each): problem b reads src[0].ptr + b*src_batch_stride, weight + b*weight_batch_stride, residual + b*residual_batch_stride and
   * writes out + b*out_batch_stride (strides in ELEMENTS of the respective type).  GMA's per-pair `attn @ v`
   * (gma/gma_utils.py:100-113) on a bf16 attention map: A = attn[b] [N][N], B = v[b]^T [C][N]. */
  int batches;
  long long src_batch_stride, weight_batch_stride, out_batch_stride, residual_batch_stride;
} pfk_conv_b16_desc;
int pfk_conv_ktot_b16(const pfk_conv_b16_desc* d);
int pfk_conv2d_b16(const pfk_conv_b16_desc* d, pfk_stream_t stream);
int pfk_debug_set_b16(int cfg);            /* K8b tile configuration: 0 = heuristic, 1..5 (pfk_gemm_b16.hip::launch_b16) */

/* ---- small direct kernels ------------------------------------------------------------------- */
/* k x k conv on a 2-channel map (the flow), relu optional: out[p*out_ld + out_coff + co].
 * in [M][in_ld] (channels 0,1 used); weight packed [k*k][2][cout]; bias [cout]. */
int pfk_conv_cin2_f32(const float* in, int in_ld, const float* weight, const float* bias,
                      float* out, int out_ld, int out_coff, int B, int H, int W, int k, int cout,
                      int relu, pfk_stream_t stream);

int pfk_debug_set_cin2_valu(int on);      /* 0 = by size; 1 = the tiled VALU kernel everywhere; 2 = the MFMA kernel (k = 7, cout 64 / 128) at every size */

/* same with a bf16 output (ABI 7; k = 3, 5 or 7): `out_bf16` holds bf16 elements, out_ld / out_coff in elements — the A operand of
 * convf2 on the K8b path (pfk_conv2d_b16) */
int pfk_conv_cin2_b16(const float* in, int in_ld, const float* weight, const float* bias,
                      void* out_bf16, int out_ld, int out_coff, int B, int H, int W, int k, int cout,
                      int relu, pfk_stream_t stream);

/* FlowHead.conv2 (3x3, cin -> 2) fused with the coordinate update of the RAFT loop:
 *   delta = conv(in) + bias ; coords1 += delta ; flow = coords1 - coords0
 * in [M][in_ld]; weight packed [9][2][cin]; coords0/coords1 [B][2][h][w] (NCHW, updated in place);
 * delta_out [B][2][h][w] (may be NULL); flow_out written as 2 channels at
 * flow_out[p*flow_ld + 0..1] (may be NULL). */
int pfk_flow_delta_f32(const float* in, int in_ld, int cin, const float* weight,
                       const float* bias, const float* coords0, float* coords1, float* delta_out,
                       float* flow_out, int flow_ld, int B, int H, int W, pfk_stream_t stream);

/* same with a bf16 input (ABI 7; the flow head's hidden activation as pfk_conv2d_b16 writes it: in_ld in elements, 8-byte aligned
 * rows) and an optional second, bf16 copy of the flow at flow_out_b16[p*flow_b16_ld + 0..1] (the 16-bit twin of the hx buffer that
 * the next iteration's convolutions read); weight / bias / coordinates fp32, fp32 accumulation. */
int pfk_flow_delta_b16(const void* in_bf16, int in_ld, int cin, const float* weight,
                       const float* bias, const float* coords0, float* coords1, float* delta_out,
                       float* flow_out, int flow_ld, void* flow_out_b16, int flow_b16_ld, int B, int H, int W, pfk_stream_t stream);

/* flow = coords1 - coords0 written pixel-major (2 channels) — loop prologue / drop-in mode. */
int pfk_flow_from_coords_f32(const float* coords0, const float* coords1, float* flow_out,
                             int flow_ld, int B, int H, int W, pfk_stream_t stream);

/* RAFT.upsample_flow: softmax over the 9 mask taps, convex combination of the 3x3 neighbourhood
 * of 8*flow, pixel-shuffle to 8x resolution.
 * flow [B][2][h][w] NCHW; mask pixel-major [M][mask_ld] with channel c = k*64 + sy*8 + sx
 * (already multiplied by 0.25); out [B][2][8h][8w] NCHW. */
int pfk_convex_upsample_f32(const float* flow, const float* mask, int mask_ld, float* out,
                            int B, int H, int W, pfk_stream_t stream);
/* upflow8 (raft/utils.py:94-96, the mask-less upsampling of raft_small): out [B][2][8H][8W] = 8 * bilinear(coords1 - coords0,
 * size 8x, align_corners = True); coords NCHW [B][2][H][W].  Index / weight arithmetic as torch's upsample_bilinear2d. */
int pfk_upflow8_f32(const float* coords0, const float* coords1, float* out, int B, int H, int W, pfk_stream_t stream);

/* ---- fused mask head conv2 + softmax + convex upsampling (round 5) ------------------------------------------------------
 * raft/update.py:138-142,152 (`0.25 * mask[2](relu(mask[0](net)))`: the 1x1 convolution to 9*64 channels) followed by
 * raft/raft.py:112-123 (`upsample_flow`) in ONE kernel: the [M][576] mask is never written.  x [M = B*H*W][x_ld] = the mask head's
 * hidden activation (cin channels, cin % 32 == 0); weight_perm [640][cin] / bias_perm [640] = the 1x1 convolution's weight and bias
 * with their ROWS in the order [quarter q (4)][tile j (5)][column c (32)]: row q*160 + j*32 + c = original channel k*64 + s with tap
 * k = 2j + (c >> 4) and sub-pixel s = q*16 + (c & 15); the rows with k == 9 (tile 4, c >= 16) are ZERO (a block's nine taps of 16
 * sub-pixels are then one contiguous range of 160 rows; ptlflow_amd/packing.py::permute_mask_head).  scale = 0.25; flow_pm
 * pixel-major (flow_pm[p*flow_ld + 0..1]); out [B][2][8H][8W] NCHW.  Bit-identical to pfk_conv2d_f32 (scale 0.25) +
 * pfk_convex_upsample_pm_f32. */
int pfk_mask_upsample_f32(const float* x, int x_ld, int cin, const float* weight_perm, const float* bias_perm, float scale,
                          const float* flow_pm, int flow_ld, float* out, int B, int H, int W, pfk_stream_t stream);

/* K13b — the same on the K8b path (ABI 7): x bf16 [M][x_ld] (cin % 64 == 0, x_ld % 8 == 0), weight_perm bf16 [640][cin] in the same
 * row order, bias / flow / out fp32.  Every logit is rounded to bf16 where pfk_conv2d_b16 (bf16 out) would store it: bit-identical
 * to that launch + pfk_convex_upsample_pm_b16. */
int pfk_mask_upsample_b16(const void* x_bf16, int x_ld, int cin, const void* weight_perm_bf16, const float* bias_perm, float scale,
                          const float* flow_pm, int flow_ld, float* out, int B, int H, int W, pfk_stream_t stream);

/* same, with the flow read pixel-major (flow_pm[p*flow_ld + 0..1], e.g. the update engine's hx slice) */
int pfk_convex_upsample_pm_f32(const float* flow_pm, int flow_ld, const float* mask, int mask_ld,
                               float* out, int B, int H, int W, pfk_stream_t stream);

/* same with a bf16 mask (ABI 7): mask_bf16 [M][mask_ld] bf16 logits (mask_ld in elements, rows 8-byte aligned) as the K8b mask head
 * writes them — what `0.25 * self.mask(net)` is under the reference's reduced-precision switch; softmax and combination in fp32 */
int pfk_convex_upsample_pm_b16(const float* flow_pm, int flow_ld, const void* mask_bf16, int mask_ld, float* out,
                               int B, int H, int W, pfk_stream_t stream);

/* ---- on-demand correlation behind the reference's alt_cuda_corr ABI --------------------------------
 * replaces corr_cuda_forward (ptlflow/utils/external/alt_cuda_corr/correlation_kernel.cu:258-285; pybind
 * `alt_cuda_corr.forward`, correlation.cpp:23-37, called from AlternateCorrBlock, raft/corr.py:76-101):
 * fmap1 [B][H1][W1][C], fmap2 [B][H2][W2][C] (NHWC fp32 contiguous), coords [B][H1][W1][2] (x, y) ->
 * out [B][(2r+1)^2][H1][W1], cell = oy + (2r+1)*ox, UNSCALED (the caller divides by sqrt(C)).
 * `workspace`: pfk_altcorr_workspace_bytes(B, H1, W1) bytes of device memory (4-byte aligned, no initialisation needed), or NULL.
 * With it the launch counts the 8x4-pixel patches whose windows share a bounding box small enough for the window-sharing MFMA
 * kernel and runs that kernel only when at least 5/8 of them qualify; otherwise (noise-like coordinate fields) every pixel
 * takes the per-pixel kernel.  NULL: the decision is per patch only. */
long long pfk_altcorr_workspace_bytes(int B, int H1, int W1);
int pfk_altcorr_forward_f32(const float* fmap1, const float* fmap2, const float* coords, float* out,
                            int B, int H1, int W1, int H2, int W2, int C, int radius,
                            void* workspace, pfk_stream_t stream);

/* The same with bf16 feature maps (NHWC bf16 contiguous; coords and out fp32): the maps are widened exactly, products and
 * accumulation are fp32 — the on-demand counterpart of the bf16 volume for bf16 / autocast callers, half the
 * gather bytes of an L2-bound kernel.  The reference's extension is float-only (correlation_kernel.cu:275) and its callers
 * up-cast half inputs (raft/corr.py:90-96); IterativeCorrBlock under autocast (ptlflow/utils/correlation.py:539-615) is the oracle. */
int pfk_altcorr_forward_bf16(const void* fmap1_bf16, const void* fmap2_bf16, const float* coords, float* out,
                             int B, int H1, int W1, int H2, int W2, int C, int radius, void* workspace, pfk_stream_t stream);

/* backward of the above w.r.t. the feature maps: corr_cuda_backward (correlation_kernel.cu:288-324, pybind
 * `alt_cuda_corr.backward`).  corr_grad [B][(2r+1)^2][H1][W1] -> fmap1_grad [B][H1][W1][C] (overwritten),
 * fmap2_grad [B][H2][W2][C] (zeroed here, then accumulated with fp32 atomics like the reference). */
int pfk_altcorr_backward_f32(const float* fmap1, const float* fmap2, const float* coords,
                             const float* corr_grad, float* fmap1_grad, float* fmap2_grad, int B, int H1,
                             int W1, int H2, int W2, int C, int radius, pfk_stream_t stream);

/* ---- weight gradient of the convolution above (torch.autograd of every nn.Conv2d in raft/update.py) ---
 * dw_packed[co][k(s,tap,c)] = sum_p dy[p][co] * src_s[p + tap][c]   — the packed [cout][ktot] layout of pfk_conv2d_f32's weight.
 * `d` describes the forward convolution (sources, B, H, W, kh, kw, cout, stride; weight/out/epilogue fields ignored);
 * dy [B*Ho*Wo][dy_ld] is the gradient w.r.t. the convolution output (Ho = (H-1)/stride + 1; stride 0 or 1: Ho = H), cout % 4 == 0;
 * with a stride the product reads src_s at (yo*stride + dy, xo*stride + dx) — no zero-upsampled gradient is needed.  The reduction runs over pixels, cut
 * into slices whose partial results go through `workspace` (pfk_conv_wgrad_workspace_bytes(d) bytes, 16-byte aligned; may
 * be NULL when that is 0) and are added in a fixed order: deterministic, no atomics.  with_bias != 0 appends 32 columns to
 * every row of dw_packed ([cout][ktot + 32]); column ktot is the bias gradient sum_p dy[p][co] (computed as one more chunk
 * whose A operand is a column of ones), the other 31 are zero.  (The data gradient needs no entry
 * point of its own: it is pfk_conv2d_f32 of dy with the spatially flipped, transposed weight.) */
long long pfk_conv_wgrad_workspace_bytes(const pfk_conv_desc* d, int with_bias);
int pfk_conv_wgrad_f32(const pfk_conv_desc* d, const float* dy, int dy_ld, float* dw_packed, int with_bias, void* workspace,
                       long long workspace_bytes, pfk_stream_t stream);

/* The same weight gradient delivered in PyTorch's layouts (what autograd accumulates): dw [cout_real][cin][kh][kw] with cin =
 * sum of real_channels[s] (the sources' real channels in order; real_channels[s] <= src[s].channels, the rest being the
 * buffers' zero padding) and db [cout_real] (NULL: no bias gradient).  d->cout is dy's channel count (a multiple of 4, >=
 * cout_real).  The slice reduction writes them directly — no packed intermediate, no un-packing copies.  accumulate != 0:
 * dw / db += the gradient (the recurrent iterations of a training step add into one buffer per parameter instead of leaving
 * eleven additions per parameter to the autograd engine); 0: overwrite.  workspace:
 * pfk_conv_wgrad_unpacked_workspace_bytes(d, db != NULL) bytes, always required. */
long long pfk_conv_wgrad_unpacked_workspace_bytes(const pfk_conv_desc* d, int with_bias);
int pfk_conv_wgrad_unpacked_f32(const pfk_conv_desc* d, const int* real_channels, const float* dy, int dy_ld, int cout_real,
                                float* dw, float* db, int accumulate, void* workspace, long long workspace_bytes, pfk_stream_t stream);

/* Gate arithmetic of one ConvGRU / SepConvGRU pass for the training path (raft/update.py:24-32, 58-73), pixel-major
 * [M][C] tensors, C % 4 == 0; z, r, rh, q, h_new, da_q, dh are contiguous [M][C], a_zr / da_zr contiguous [M][2C]
 * (z half first), h / dh_new have their own row stride.
 *   gates_zr:     z = sigmoid(a_zr[:, :C]), r = sigmoid(a_zr[:, C:]), rh = r * h
 *   gates_q:      q = tanh(a_q), h_new = (1 - z) * h + z * q
 *   backward_q:   da_q = dh_new * z * (1 - q^2);  da_zr[:, :C] = dh_new * (q - h) * z * (1 - z);  dh = dh_new * (1 - z)
 *   backward_zr:  da_zr[:, C:] = d_rh * h * r * (1 - r);  dh += d_rh * r */
int pfk_gru_gates_zr_f32(const float* a_zr, const float* h, int h_ld, float* z, float* r, float* rh, long long M, int C,
                         pfk_stream_t stream);
int pfk_gru_gates_q_f32(const float* a_q, const float* z, const float* h, int h_ld, float* q, float* h_new, long long M, int C,
                        pfk_stream_t stream);
int pfk_gru_backward_q_f32(const float* dh_new, int dh_new_ld, const float* z, const float* q, const float* h, int h_ld,
                           float* da_q, float* da_zr, float* dh, long long M, int C, pfk_stream_t stream);
int pfk_gru_backward_zr_f32(const float* d_rh, const float* h, int h_ld, const float* r, float* da_zr, float* dh, long long M,
                            int C, pfk_stream_t stream);

/* ---- warm start (second half) -------------------------------------------------------------------
 * forward_interpolate (ptlflow/utils/external/raft.py:155-185, batched by utils/utils.py:454-478): flow [B][2][H][W] ->
 * out [B][2][H][W]: every grid point takes the flow of the nearest forward-projected pixel that lands strictly inside the
 * image (float64 distances, exact ties to the lowest source index; all-invalid -> 0).  Replaces the reference's per-sample
 * host round trip through scipy.interpolate.griddata. */
int pfk_forward_interpolate_f32(const float* flow, float* out, int B, int H, int W, pfk_stream_t stream);

/* ---- encoder pieces (ptlflow/models/raft/extractor.py:122-194 BasicEncoder) -----------------
 * The residual blocks' 3x3 / 1x1 convolutions (stride 1 and 2) are pfk_conv2d_f32 / pfk_conv2d_bf16s with `stride`
 * and `relu_after_residual`; eval-mode BatchNorm is folded into their weights by the caller.  What is left:
 *
 * stem = Conv2d(3, cout, 7, stride=2, padding=3) (extractor.py:146) read from the NCHW image:
 * img [B][3][H][W]; weight packed [49][3][cout] (tap ky*7+kx major); out pixel-major [B*Ho*Wo][out_ld], Ho=(H-1)/2+1. */
int pfk_conv_stem_f32(const float* img, const float* weight, const float* bias, float* out, int out_ld, int B,
                      int H, int W, int cout, int relu, pfk_stream_t stream);

/* same with a bf16 output (ABI 7; cout = 32 or 64 only — the MFMA kernel): what the K8b convolutions (pfk_conv2d_b16) read next */
int pfk_conv_stem_b16(const float* img, const float* weight, const float* bias, void* out_bf16, int out_ld, int B,
                      int H, int W, int cout, int relu, pfk_stream_t stream);
int pfk_debug_set_stem_valu(int on);       /* 1 = pfk_conv_stem_f32 always takes the VALU kernel (A/B timing) */

/* InstanceNorm2d statistics (extractor.py:136-140; affine=False, biased variance, eps inside the sqrt) over
 * pixel-major x[B*HW][ld], channels [0, C): mean[b*C+c], rstd[b*C+c].  One pass over x, deterministic chunked reduction of sum(x) and
 * sum(x*x) in double (var = E[x^2] - mean^2 evaluated in double; results rounded to fp32); needs pfk_instnorm_workspace_bytes(B, C) bytes of 32-byte-aligned device scratch.
 * Training-mode batch norm statistics are the same call with B = 1 and HW = every pixel of the batch. */
long long pfk_instnorm_workspace_bytes(int B, int C);
int pfk_instnorm_stats_f32(const float* x, int ld, int B, int HW, int C, float eps, float* mean, float* rstd,
                           void* workspace, long long workspace_bytes, pfk_stream_t stream);

/* out = relu?((x - mean) * rstd) ; if residual: out = relu?(residual + out)   — norm + relu (+ the ResidualBlock's
 * `relu(x + y)`, extractor.py:53-61) in one pass.  out may alias x. */
int pfk_norm_apply_f32(const float* x, int x_ld, const float* mean, const float* rstd, const float* residual,
                       int residual_ld, float* out, int out_ld, int B, int HW, int C, int relu,
                       int relu_after_residual, pfk_stream_t stream);

/* same with bf16 residual rows and a bf16 output (ABI 7; ld in elements, 8-byte aligned rows).  x = the convolution output the
 * statistics were taken from: fp32 rows (x_bf16 = 0), or bf16 rows (x_bf16 = 1; `out_bf16` may be `x` itself: in place) — what
 * F.instance_norm reads under the reference's autocast switch, where the convolution in front of it returns a 16-bit tensor
 * (the arithmetic is fp32 either way: instance_norm is on autocast's fp32 list). */
int pfk_norm_apply_b16(const void* x, int x_bf16, int x_ld, const float* mean, const float* rstd, const void* residual_bf16,
                       int residual_ld, void* out_bf16, int out_ld, int B, int HW, int C, int relu,
                       int relu_after_residual, pfk_stream_t stream);
/* the statistics of bf16 rows (pfk_instnorm_stats_f32 otherwise: one pass, double partial sums, deterministic) */
int pfk_instnorm_stats_b16(const void* x_bf16, int ld, int B, int HW, int C, float eps, float* mean, float* rstd,
                           void* workspace, long long workspace_bytes, pfk_stream_t stream);

/* In-place softmax over each row of x [rows][ld] (cols entries used) — GMA's attention map (gma/gma_utils.py:75-76:
 * `sim.softmax(dim=-1)`, once per forward; the similarity itself is pfk_corr_volume_f32 of the q / k maps). */
int pfk_softmax_rows_f32(float* x, long long rows, int cols, long long ld, pfk_stream_t stream);

/* ---- encoder backward (training, BASELINE config 5) ---------------------------------------------------------------
 * Backward of y = relu?((x - mean) * rstd) with per-(image, channel) statistics (pfk_instnorm_stats_f32): instance norm,
 * and batch norm in training mode when called with B = 1 and HW = all pixels of the batch (torch.autograd of
 * nn.InstanceNorm2d / nn.BatchNorm2d + relu in raft/extractor.py:51-59, 172-181):
 *   g = dy * (y > 0);  sum_g[b*C+c] = sum_p g;  sum_gxhat[b*C+c] = sum_p g * x_hat;  dx = rstd * (g - sum_g/HW - x_hat * sum_gxhat/HW)
 * sum_g / sum_gxhat may be NULL (kept in the workspace); for an affine batch norm they are d(beta) / d(gamma) when dy is the
 * gradient w.r.t. the un-affine'd output.  The two sums and the final subtraction are carried in double (the residual of a
 * mean-dominated gradient is otherwise lost to fp32 cancellation).  workspace: pfk_norm_bwd_workspace_bytes(B, C) bytes,
 * 32-byte aligned.  Deterministic. */
long long pfk_norm_bwd_workspace_bytes(int B, int C);
int pfk_norm_bwd_f32(const float* x, int x_ld, const float* dy, int dy_ld, const float* mean, const float* rstd, float* dx,
                     int dx_ld, float* sum_g, float* sum_gxhat, int B, int HW, int C, int relu, void* workspace,
                     long long workspace_bytes, pfk_stream_t stream);
/* Weight / bias gradient of the stem (pfk_conv_stem_f32): dy pixel-major [B*Ho*Wo][dy_ld] -> dw [49][3][cout] (the packed
 * layout of the forward's weight), db [cout] (may be NULL).  cout <= 64.  workspace: pfk_conv_stem_wgrad_workspace_bytes(). */
long long pfk_conv_stem_wgrad_workspace_bytes(void);
int pfk_conv_stem_wgrad_f32(const float* img, const float* dy, int dy_ld, float* dw, float* db, int B, int H, int W, int cout,
                            void* workspace, long long workspace_bytes, pfk_stream_t stream);

/* NCHW [B][C][H][W] -> pixel-major [B*H*W][ld] (+ channel offset) and back. */
int pfk_nchw_to_pm_f32(const float* in, float* out, int out_ld, int out_coff, int B, int C,
                       int H, int W, pfk_stream_t stream);
int pfk_pm_to_nchw_f32(const float* in, int in_ld, int in_coff, float* out, int B, int C, int H,
                       int W, pfk_stream_t stream);
/* pixel-major [B*N][in_ld] -> channel-major out[b][c][n] with row stride out_ld >= N (a K-contiguous "weight"
 * operand for pfk_conv2d_f32, e.g. GMA's value matrix V^T, gma_utils.py:103-105) */
int pfk_pm_to_cm_f32(const float* in, int in_ld, float* out, int out_ld, int B, int C, int N,
                     pfk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PFK_H */
