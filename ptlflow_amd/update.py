"""Seam B3 — the iterative update block (ptlflow/models/raft/update.py:6-153) on gfx950 kernels.

`UpdateEngine` owns the packed weights and the pixel-major work buffers and runs one iteration as a
fixed chain of launches on torch's current stream:

    lookup (K3, by the caller)                                   -> corr   [M, Cc]
    convc1 1x1 (+relu)            MFMA implicit GEMM             -> cor1   [M, c1]
    convc2 3x3 (+relu)            MFMA                           -> corflo [M, :c2]
    convf1 7x7 on flow (+relu)    direct                         -> flo1   [M, f1]
    convf2 3x3 (+relu)            MFMA                           -> corflo [M, c2:]
    conv   3x3 (+relu)            MFMA                           -> hx     [M, Ch+Ci : Ch+Ci+co]
    GRU pass(es): z|r conv (+sigmoid, r*h)  MFMA, one GEMM for both gates -> z, rh
                  q conv (+tanh, blend)      MFMA                 -> hx[:, :Ch] (in place)
                  (the context part of both convolutions — `inp` x its weight slice + bias — is loop-invariant:
                   computed once per forward by `prepare_context`, added in the gate epilogues)
    flow-head conv1 | mask conv1 3x3 (+relu)  MFMA, one GEMM      -> fm     [M, 2*fh]
    flow-head conv2 3x3 -> delta; coords1 += delta; flow = coords1 - coords0   direct, fused
    mask conv2 1x1, *0.25         MFMA                           -> mask   [M, 576]

``hx`` is the reference's ``torch.cat([net, inp, motion_features])`` laid out once:
``[ h (Ch) | inp (Ci) | encoder out (co) | flow (2) | zero pad ]`` — every producer writes its channel
slice in place, so none of the six ``torch.cat`` calls of update.py exists here.

`PfkUpdateBlock` wraps a reference ``BasicUpdateBlock`` / ``SmallUpdateBlock`` module *in place of its
forward*: the original sub-modules (and therefore ``state_dict`` keys, checkpoints, optimizers) stay,
weights are re-packed whenever a parameter's ``_version`` changes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import load_native
from .packing import permute_mask_head, pack_cin2_weight, pack_conv_weight, pack_flow_head_weight, round_up, split_bf16_planes

EPI_LINEAR, EPI_GRU_ZR, EPI_GRU_Q = 0, 1, 2


@dataclass(frozen=True)
class UpdateSpec:
    hidden: int          # Ch
    context: int         # Ci
    corr_channels: int   # L*(2r+1)^2
    c1: int              # convc1 out
    c2: int              # convc2 out (0: SmallMotionEncoder has no convc2)
    f1: int              # convf1 out
    f2: int              # convf2 out
    enc_out: int         # encoder.conv out (126 / 80)
    gru_passes: Tuple[Tuple[int, int, str], ...]  # (kh, kw, key suffix)
    fh_hidden: int
    has_mask: bool
    aggregate: bool = False   # GMA: + attention-aggregated motion features (gma/update.py:148-152)
    mask_channels: int = 576  # mask head's output: 9 * scale^2 (raft / gma: scale 8; ccmr / ms_raft_plus: scale 2 -> 36)
    external_aggregate: bool = False   # CCMR: the second motion-feature slot is filled by the caller's own module
                                       # (an XCiT block per scale, ccmr/update.py:135-168), not by GMA's attn @ v

    @property
    def motion_channels(self) -> int:     # encoder out | flow
        return self.enc_out + 2

    @property
    def x_channels(self) -> int:          # inp | encoder out | flow [| aggregated motion features]
        return self.context + self.motion_channels * (2 if self.aggregate else 1)

    @property
    def hx_channels(self) -> int:         # padded to a multiple of 4
        return round_up(self.hidden + self.x_channels, 4)


def basic_spec(corr_levels: int = 4, corr_radius: int = 4) -> UpdateSpec:
    """BasicUpdateBlock (raft/update.py:131-142)."""
    return UpdateSpec(128, 128, corr_levels * (2 * corr_radius + 1) ** 2, 256, 192, 128, 64, 126,
                      ((1, 5, "1"), (5, 1, "2")), 256, True)


def gma_spec(corr_levels: int = 4, corr_radius: int = 4) -> UpdateSpec:
    """GMAUpdateBlock (gma/update.py:127-160): BasicUpdateBlock + Aggregate, SepConvGRU input 128+128+128."""
    return UpdateSpec(128, 128, corr_levels * (2 * corr_radius + 1) ** 2, 256, 192, 128, 64, 126,
                      ((1, 5, "1"), (5, 1, "2")), 256, True, True)


def ccmr_spec(corr_channels: int, mask_channels: int = 36) -> UpdateSpec:
    """CCMR's BasicUpdateBlock (ccmr/update.py:110-168): RAFT's motion encoder, SepConvGRU with input 128 + 128 + 128 (inp |
    motion features | globally aggregated motion features from an XCiT block), FlowHead, mask head for x2 convex upsampling."""
    return UpdateSpec(128, 128, corr_channels, 256, 192, 128, 64, 126, ((1, 5, "1"), (5, 1, "2")), 256, True, True,
                      mask_channels, True)


def ms_raft_plus_spec(corr_channels: int, mask_channels: int = 36) -> UpdateSpec:
    """MS-RAFT+'s BasicUpdateBlock with stack_coords=False (ms_raft_plus/update.py:119-153): RAFT's block with a x2 mask head."""
    return UpdateSpec(128, 128, corr_channels, 256, 192, 128, 64, 126, ((1, 5, "1"), (5, 1, "2")), 256, True, False,
                      mask_channels)


def small_spec(corr_levels: int = 4, corr_radius: int = 3) -> UpdateSpec:
    """SmallUpdateBlock (raft/update.py:115-120)."""
    return UpdateSpec(96, 64, corr_levels * (2 * corr_radius + 1) ** 2, 96, 0, 64, 32, 80,
                      ((3, 3, ""),), 128, False)


# conv_precision -> number of bf16 planes per operand (0 = the fp32 matrix-core path, the default and the parity path)
CONV_PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3}


class UpdateEngine:
    """``hoist_context`` (default on): the GRU convolutions of raft/update.py:60-71 / :27-30 run over
    ``cat([h, inp, motion...])``, and ``inp`` — the context features, raft.py:158-160 — is the same tensor in every one of
    the 12 / 32 iterations.  A convolution is linear in its input channels, so

        conv(W, cat([h, inp, m])) + b  =  conv(W[:, h | m slices], cat([h, m]))  +  ( conv(W[:, inp slice], inp) + b )

    and the second term is loop-invariant: `prepare_context()` computes it ONCE per forward (one launch per gate convolution,
    a third of one iteration's GRU work) and the per-iteration launches cover the h and motion channels only, adding the
    stored term to the gate pre-activation in their epilogue (`residual` of PFK_EPI_GRU_ZR / _Q, include/pfk.h).  A third of
    the SepConvGRU's multiply-adds (128 of 384 input channels; 15.8 % of the update block's) leave the loop.  Same sum, other
    association: results differ from the single-chain form by fp32 rounding only (the tests' per-convolution and EPE gates
    are unchanged).  ``hoist_context=False`` keeps the single-chain launches (A/B, tests)."""

    def __init__(self, params: Dict[str, torch.Tensor], spec: UpdateSpec, device: torch.device,
                 conv_precision: str = "fp32", hoist_context: bool = True):
        load_native()
        self.hoist_context = bool(hoist_context)
        if conv_precision not in CONV_PRECISIONS:
            raise ValueError(f"conv_precision must be one of {sorted(CONV_PRECISIONS)}, got {conv_precision!r}")
        self.conv_precision = conv_precision
        self.nsplit = CONV_PRECISIONS[conv_precision]
        # "bf16" = BASELINE config 3's precision as the reference's reduced-precision switch gives it (model_benchmark.py:317-319 /
        # torch.autocast: every convolution reads and writes 16-bit tensors): K8b, `pfk_conv2d_b16` — bf16 ACTIVATION STORAGE between
        # this engine's own kernels (producer epilogues emit bf16 once), both GEMM operands by LDS-DMA; the recurrent state h, the
        # coordinates / flow and every accumulator stay fp32.  GMA's aggregate branch runs on a bf16 copy of the attention map (what
        # `attn @ v` reads under autocast: half the bytes of the 198 MB-per-pair operand that every iteration streams).  CCMR (the
        # caller's torch module reads the motion features) keeps the split kernel with one plane (fp32 storage).
        self.b16 = conv_precision == "bf16" and not spec.external_aggregate
        self.ops = torch.ops.pfk
        self.spec = spec
        self.device = device
        self._shape: Optional[Tuple[int, int, int]] = None
        self._fault_host: Optional[torch.Tensor] = None
        self._fault_event = None
        # optional per-launch instrumentation: key -> [(start_event, end_event)], and key -> algorithmic FLOPs
        self.profile: Optional[Dict[str, list]] = None
        self.flops: Dict[str, float] = {}
        self.bytes: Dict[str, float] = {}
        self.pack(params)

    # ------------------------------------------------------------------ weights
    def pack(self, P: Dict[str, torch.Tensor]) -> None:
        s, dev = self.spec, self.device

        def g(name):
            return P[name].detach().to(device=dev, dtype=torch.float32)

        def seg1(c):
            return [(0, c, round_up(c, 8 if self.b16 else 4))]

        def pk(weight, segments):
            # the weight tensor's dtype selects the kernel in torch.ops.pfk.conv2d: fp32 [cout, ktot] or bf16 planes
            if self.b16:      # K8b: one bf16 matrix [cout, ktot], 64-channel K chunks
                return pack_conv_weight(weight, segments, kpad=64).to(torch.bfloat16).contiguous()
            if self.nsplit == 0:
                return pack_conv_weight(weight, segments)
            return split_bf16_planes(pack_conv_weight(weight, segments), self.nsplit)

        w: Dict[str, torch.Tensor] = {}
        real = s.hidden + s.x_channels
        self._real_cin = {"c1": s.corr_channels, "c2": s.c1, "f2": s.f1, "cv": (s.c2 if s.c2 else s.c1) + s.f2,
                          "fm": s.hidden, "mk": s.fh_hidden}
        for _, _, sfx in s.gru_passes:
            # the launches' own multiply-adds: with the context term hoisted they cover the h and motion channels only
            self._real_cin["zr" + sfx] = real - (s.context if self.hoist_context else 0)
            self._real_cin["q" + sfx] = real - (s.context if self.hoist_context else 0)
            self._real_cin["zrc" + sfx] = s.context
            self._real_cin["qc" + sfx] = s.context
        w["c1.w"] = pk(g("encoder.convc1.weight"), seg1(s.corr_channels))
        w["c1.b"] = g("encoder.convc1.bias").contiguous()
        if s.c2:
            w["c2.w"] = pk(g("encoder.convc2.weight"), seg1(s.c1))
            w["c2.b"] = g("encoder.convc2.bias").contiguous()
        w["f1.w"] = pack_cin2_weight(g("encoder.convf1.weight"))
        w["f1.b"] = g("encoder.convf1.bias").contiguous()
        w["f2.w"] = pk(g("encoder.convf2.weight"), seg1(s.f1))
        w["f2.b"] = g("encoder.convf2.bias").contiguous()
        cf = (s.c2 if s.c2 else s.c1) + s.f2
        w["cv.w"] = pk(g("encoder.conv.weight"), seg1(cf))
        w["cv.b"] = g("encoder.conv.bias").contiguous()
        Ch = s.hidden
        hxc = round_up(s.hidden + s.x_channels, 8) if self.b16 else s.hx_channels      # (the bf16 twin of hx has 16-byte rows)
        real = Ch + s.x_channels
        for kh, kw, sfx in s.gru_passes:
            wz, wr, wq = (g(f"gru.conv{k}{sfx}.weight") for k in "zrq")
            bz, br, bq = (g(f"gru.conv{k}{sfx}.bias") for k in "zrq")
            if self.hoist_context:
                # per iteration: sources h (hx[:, :Ch]) / r*h and the rest of x behind the context slice (motion features ...);
                # once per forward: the context slice with the biases (prepare_context)
                Ci = s.context
                rest = [(0, Ch, Ch), (Ch + Ci, real - Ch - Ci, hxc - Ch - Ci)]
                wzr = torch.cat([wz, wr], 0)
                w[f"zr{sfx}.w"] = pk(wzr, rest)
                w[f"q{sfx}.w"] = pk(wq, rest)
                w[f"zrc{sfx}.w"] = pk(wzr, [(Ch, Ci, Ci)])
                w[f"zrc{sfx}.b"] = torch.cat([bz, br]).contiguous()
                w[f"qc{sfx}.w"] = pk(wq, [(Ch, Ci, Ci)])
                w[f"qc{sfx}.b"] = bq.contiguous()
                continue
            # z|r: one source = the whole hx row (h | x), padded channels get zero weights
            w[f"zr{sfx}.w"] = pk(torch.cat([wz, wr], 0), [(0, real, hxc)])
            w[f"zr{sfx}.b"] = torch.cat([bz, br]).contiguous()
            # q: sources r*h (own buffer) and x (hx slice)
            w[f"q{sfx}.w"] = pk(wq, [(0, Ch, Ch), (Ch, real - Ch, hxc - Ch)])
            w[f"q{sfx}.b"] = bq.contiguous()
        if s.has_mask:
            w["fm.w"] = pk(torch.cat([g("flow_head.conv1.weight"), g("mask.0.weight")], 0), seg1(Ch))
            w["fm.b"] = torch.cat([g("flow_head.conv1.bias"), g("mask.0.bias")]).contiguous()
            # the mask head's hidden width = what `fm` holds behind the flow head's half (raft/update.py:131-135: both 256)
            mask_hidden = g("mask.2.weight").shape[1]
            assert mask_hidden == g("mask.0.weight").shape[0], "mask.2 reads mask.0's output"
            w["mk.w"] = pk(g("mask.2.weight"), seg1(mask_hidden))
            w["mk.b"] = g("mask.2.bias").contiguous()
            if self.nsplit == 0 and s.mask_channels == 576 and mask_hidden % 32 == 0 and mask_hidden == s.fh_hidden:
                # fused mask conv2 + softmax + convex upsampling (`pfk_mask_upsample_f32`): the 1x1 weight / bias with their rows in
                # the kernel's [quarter][tile][32] order (packing.permute_mask_head, include/pfk.h); `fm` is split at fh_hidden, so
                # the fused kernel is only armed where the two hidden widths agree
                w["mku.w"], w["mku.b"] = permute_mask_head(pack_conv_weight(g("mask.2.weight"), seg1(mask_hidden)), w["mk.b"])
            elif self.b16 and s.mask_channels == 576 and mask_hidden % 64 == 0 and mask_hidden == s.fh_hidden:
                # K13b (`pfk_mask_upsample_b16`): the same row order, bf16 (whole 64-channel K-steps: the packed K is the plain channel order)
                wp, bp = permute_mask_head(pack_conv_weight(g("mask.2.weight"), seg1(mask_hidden)), w["mk.b"])
                w["mku.w"], w["mku.b"] = wp.to(torch.bfloat16).contiguous(), bp
            # flow-head conv1 alone: the iterations whose mask is never looked at (`upsample_every_iter=False`) skip the mask half
            w["fh.w"] = pk(g("flow_head.conv1.weight"), seg1(Ch))
            w["fh.b"] = g("flow_head.conv1.bias").contiguous()
            self._real_cin["fh"] = s.hidden
        else:
            w["fm.w"] = pk(g("flow_head.conv1.weight"), seg1(Ch))
            w["fm.b"] = g("flow_head.conv1.bias").contiguous()
        if s.aggregate and not s.external_aggregate:
            w["tv.w"] = pk(g("aggregator.to_v.weight"), seg1(s.motion_channels))
            w["tv.b"] = None
            self.gamma = float(P["aggregator.gamma"].detach().float().cpu().item())   # one host read per (re)pack
            self._real_cin["tv"] = s.motion_channels
        w["fh2.w"] = pack_flow_head_weight(g("flow_head.conv2.weight"))
        w["fh2.b"] = g("flow_head.conv2.bias").contiguous()
        self.w = w

    # ------------------------------------------------------------------ buffers
    def bind(self, B: int, H: int, W: int) -> None:
        if self._shape == (B, H, W):
            return
        s, dev = self.spec, self.device
        M = B * H * W
        z = lambda c: torch.zeros(M, c, device=dev, dtype=torch.float32)  # noqa: E731
        # activations between this engine's kernels: fp32, or bf16 on the K8b path
        a = (lambda c: torch.zeros(M, c, device=dev, dtype=torch.bfloat16)) if self.b16 else z  # noqa: E731
        self.hx = z(s.hx_channels)
        # K8b: the 16-bit twin of hx the convolutions read — same channel layout; h is mirrored by the q epilogue, the motion
        # features are written here only, the flow by the flow-head kernel (both widths)
        self.hxb = a(round_up(s.hidden + s.x_channels, 8)) if self.b16 else None
        self.corr16 = a(round_up(s.corr_channels, 8)) if self.b16 else None      # the lookup's output (pad columns stay zero)
        self.cor1 = a(s.c1) if s.c2 else None   # SmallMotionEncoder: convc1 writes corflo[:, :c1] directly
        self.corflo = a((s.c2 if s.c2 else s.c1) + s.f2)
        self.flo1 = a(s.f1)
        self.zbuf = a(s.hidden)
        self.rh = a(s.hidden)
        # loop-invariant gate pre-activations (prepare_context): per GRU pass [M, 2 Ch] for z|r and [M, Ch] for q
        self.ctx = {("zr" + sfx): a(2 * s.hidden) for _, _, sfx in s.gru_passes} if self.hoist_context else {}
        if self.hoist_context:
            self.ctx.update({("q" + sfx): a(s.hidden) for _, _, sfx in s.gru_passes})
        self.fm = a(s.fh_hidden * (2 if s.has_mask else 1))
        # the mask logits: fp32, or bf16 on the K8b path where the consumer is this package's own upsampling kernel (mask_channels
        # 576; `mask_nchw()` hands the seam's torch consumer an fp32 copy)
        self.mask = (a if s.mask_channels == 576 else z)(s.mask_channels) if s.has_mask else None
        self._scratch_c0 = torch.zeros(B, 2, H, W, device=dev, dtype=torch.float32)
        self._scratch_c1 = torch.zeros(B, 2, H, W, device=dev, dtype=torch.float32)
        self._delta = torch.zeros(B, 2, H, W, device=dev, dtype=torch.float32)
        if s.aggregate and not s.external_aggregate:
            N = H * W
            self.vbuf = a(s.motion_channels)
            if self.b16:      # K-contiguous B operand of the batched attn @ v GEMM: [B][C][N padded to the kernel's 64-wide K-steps]
                self.vT = torch.zeros(B, s.motion_channels, round_up(round_up(N, 8), 64), device=dev, dtype=torch.bfloat16)
            else:
                self.vT = torch.zeros(B, s.motion_channels, round_up(N, 32), device=dev, dtype=torch.float32)
            self.attn = None
        # scratch for the stream-K conv schedules (partials + flags), one per stream that may run convolutions concurrently: the
        # main chain, the flow branch of the motion encoder and the mask branch (RAFT._iterate's forked mode) — with its own
        # workspace a branch's launches pick the same schedule as in the serial order (bit-identical results)
        self._empty = torch.empty(0, device=dev, dtype=torch.float32)      # "no bias" / "no residual" entry of a grouped launch
        nws = self.ops.conv_workspace_bytes()
        self.workspace = torch.zeros(nws, device=dev, dtype=torch.uint8)
        self.workspace_flow = torch.zeros(nws, device=dev, dtype=torch.uint8)
        self.workspace_mask = torch.zeros(nws, device=dev, dtype=torch.uint8)
        self._shape = (B, H, W)

    def faults(self) -> int:
        """Stream-K fix-ups that timed out since the workspace was created (pfk_conv_workspace_fault_offset): 0 means every
        convolution result so far is complete.  Reads one word back from the device (synchronises the stream); affected
        tiles are NaN in any case, so this is for telling *why* an output went NaN."""
        off = self.ops.conv_workspace_fault_offset()
        return sum(int(ws[off: off + 4].view(torch.int32).item()) for ws in (self.workspace, self.workspace_flow, self.workspace_mask))

    def watch_faults(self) -> None:
        """Called once per forward by the callers (RAFT mirror, PfkUpdateBlock): raise if an EARLIER forward's fault word came
        back non-zero, then queue a non-blocking read-back of the current one into pinned host memory.  No host
        synchronisation: a timed-out fix-up (whose tile is NaN already) is reported one forward later, and the workspace is
        zero-filled again before raising — a producer that published after its consumer gave up leaves its flag set, which the
        next launch on the same workspace would otherwise take for a fresh partial (stale data, not NaN)."""
        if self._fault_host is None:
            self._fault_host = torch.zeros(3, dtype=torch.int32).pin_memory()
            self._fault_event = None
        if self._fault_event is not None and self._fault_event.query():
            n = int(self._fault_host.sum())
            self._fault_event = None
            if n:
                for ws in (self.workspace, self.workspace_flow, self.workspace_mask):
                    ws.zero_()
                raise RuntimeError(f"libpfk: {n} stream-K fix-up(s) timed out in an earlier forward (its affected tiles are NaN); "
                                   "the workspace has been re-initialised")
        if self._fault_event is None and not torch.cuda.is_current_stream_capturing():
            off = self.ops.conv_workspace_fault_offset()
            for i, ws in enumerate((self.workspace, self.workspace_flow, self.workspace_mask)):
                self._fault_host[i: i + 1].copy_(ws[off: off + 4].view(torch.int32), non_blocking=True)
            self._fault_event = torch.cuda.Event()
            self._fault_event.record()

    # views into hx
    @property
    def act(self):
        """the buffer the convolutions read their [h | inp | motion | flow] operand from: hx, or its bf16 twin (K8b)"""
        return self.hxb if self.b16 else self.hx

    @property
    def lookup_out(self):
        """where the caller's lookup should write for this engine (None: the correlation block's own fp32 buffer)"""
        return self.corr16 if self.b16 else None

    def flow_changed(self) -> None:
        """the flow slice of hx was written by something other than `flow_delta` (loop prologue, the drop-in seam): mirror it"""
        if self.b16:
            s = self.spec
            o = s.hidden + s.context + s.enc_out
            self.hxb[:, o: o + 2].copy_(self.hx[:, o: o + 2])

    def state_changed(self) -> None:
        """h / inp slices of hx were (re)loaded: mirror them into the bf16 twin"""
        if self.b16:
            n = self.spec.hidden + self.spec.context
            self.hxb[:, :n].copy_(self.hx[:, :n])

    @property
    def h_view(self):
        return self.hx[:, : self.spec.hidden]

    @property
    def inp_view(self):
        s = self.spec
        return self.hx[:, s.hidden: s.hidden + s.context]

    @property
    def x_view(self):
        return self.hx[:, self.spec.hidden:]

    @property
    def flow_view(self):
        s = self.spec
        o = s.hidden + s.context + s.enc_out
        return self.hx[:, o: o + 2]

    def set_attention(self, attn: torch.Tensor) -> None:
        """GMA attention map [B, 1, N, N] (gma_utils.py:76-78), kept for the whole forward; rows are the K-contiguous
        A operand of the per-iteration `attn @ v` GEMM (row stride must be a multiple of 4 floats)."""
        B, H, W = self._shape
        N = H * W
        if attn.dim() != 4 or attn.shape[0] != B or attn.shape[1] != 1 or attn.shape[2] != N or attn.shape[3] != N:
            raise RuntimeError(f"attention must be [B,1,N,N] with one head, got {tuple(attn.shape)}")
        if self.b16:      # the bf16 copy the per-iteration GEMM streams (rows padded to 16 bytes, pad columns zero), once per forward
            a = torch.zeros(B, N, round_up(N, 8), device=attn.device, dtype=torch.bfloat16) if N % 8 else \
                torch.empty(B, N, N, device=attn.device, dtype=torch.bfloat16)
            a[:, :, :N].copy_(attn.reshape(B, N, N))
            self.attn = a
            return
        a = attn.float().reshape(B, N, N)
        if N % 4 or not a.is_contiguous():
            pad = torch.zeros(B, N, round_up(N, 4), device=a.device, dtype=torch.float32)
            pad[:, :, :N] = a
            a = pad
        self.attn = a

    def _aggregate(self) -> None:
        """Aggregate.forward (gma_utils.py:100-113): v = to_v(mf); mf_global = mf + gamma * (attn @ v)."""
        s = self.spec
        B, H, W = self._shape
        N = H * W
        o = s.hidden + s.context
        mc = s.motion_channels
        if self.b16:
            # K8b: v = to_v(mf) in bf16, V^T by one small strided copy, then ONE batched GEMM over the pairs —
            # out[b] = mf[b] + gamma * attn[b] @ v[b] with the bf16 attention map as the LDS-DMA'd A operand
            hxb = self.hxb.view(B, N, self.hxb.shape[1])
            self._conv([self.hxb[:, o: o + mc]], 1, 1, "tv", mc, relu=False, out=self.vbuf)
            self.vT[:, :, :N].copy_(self.vbuf.view(B, N, mc).transpose(1, 2))
            prof = self.profile
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.ops.conv2d_b16([self.attn], 1, H, W, 1, 1, self.vT, None, mc, EPI_LINEAR, False, self.gamma,
                                hxb[:, :, o + mc: o + 2 * mc], None, None, None, None, hxb[:, :, o: o + mc])
            if prof is not None:
                e1.record()
                prof.setdefault("ag", []).append((e0, e1))
                self.flops["ag"] = 2.0 * B * N * N * mc
            return
        mf = self.hx[:, o: o + mc]
        self._conv([mf], 1, 1, "tv", mc, relu=False, out=self.vbuf)
        self.ops.pm_to_cm(self.vbuf, self.vT)
        if self.nsplit == 0 and B > 1:
            # fp32: the pairs' `attn[b] @ v[b]` are INDEPENDENT GEMMs of 110 x 2 = 220 tiles each (one block per CU on 220 CUs, one
            # wave per SIMD: 0.45 of the matrix peak) — as ONE grouped grid (`pfk_conv2d_group_f32`, up to four problems per launch)
            # the 880 tiles of four pairs are resident together, three blocks per CU.  Same tiles, same K order: same bits.
            for b0 in range(0, B, 4):
                bs = list(range(b0, min(B, b0 + 4)))
                rows = [slice(b * N, (b + 1) * N) for b in bs]
                prof = self.profile
                if prof is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self.ops.conv2d_group([self.attn[b] for b in bs], 1, H, W, [1] * len(bs), [self.vT[b] for b in bs],
                                      [self._empty] * len(bs), [0] * len(bs), [self.gamma] * len(bs),
                                      [self.hx[r, o + mc: o + 2 * mc] for r in rows], [self.hx[r, o: o + mc] for r in rows])
                if prof is not None:
                    e1.record()
                    prof.setdefault("ag", []).append((e0, e1))
                    self.flops["ag"] = 2.0 * len(bs) * N * N * mc
            return
        for b in range(B):
            rows = slice(b * N, (b + 1) * N)
            src = self.attn[b]
            prof = self.profile
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            # split modes: V^T changes every iteration, so its bf16 planes are cut here (three tiny elementwise kernels)
            vt = self.vT[b] if self.nsplit == 0 else split_bf16_planes(self.vT[b], self.nsplit)
            self.ops.conv2d([src], 1, H, W, 1, 1, vt, None, mc, EPI_LINEAR, False, self.gamma,
                            self.hx[rows, o + mc: o + 2 * mc], None, None, None, self.workspace, self.hx[rows, o: o + mc])
            if prof is not None:
                e1.record()
                prof.setdefault("ag", []).append((e0, e1))
                self.flops["ag"] = 2.0 * N * N * mc

    def load_state(self, net: torch.Tensor, inp: torch.Tensor) -> None:
        """NCHW ``net`` / ``inp`` -> their hx slices (once per forward), then the loop-invariant context terms."""
        B, H, W = self._shape
        for src, dst in ((net, self.h_view), (inp, self.inp_view)):
            src = src.float()
            if src.is_contiguous():       # NCHW producer (torch encoders, the drop-in seam): tiled transpose kernel
                self.ops.nchw_to_pm(src, dst)
            else:                         # channels-last producer (the native encoders): rows are already pixel-major
                dst.view(B, H, W, dst.shape[1]).copy_(src.permute(0, 2, 3, 1))
        self.state_changed()
        self.prepare_context()

    def prepare_context(self) -> None:
        """Once per forward, after `inp` is in its hx slice: conv(W[:, inp slice], inp) + bias of every gate convolution
        (class docstring) into `self.ctx` — what the per-iteration launches add to their pre-activations."""
        if not self.hoist_context:
            return
        Ch, Ci = self.spec.hidden, self.spec.context
        inp = self.act[:, Ch: Ch + Ci]
        for kh, kw, sfx in self.spec.gru_passes:
            self._conv([inp], kh, kw, "zrc" + sfx, 2 * Ch, relu=False, out=self.ctx["zr" + sfx])
            self._conv([inp], kh, kw, "qc" + sfx, Ch, relu=False, out=self.ctx["q" + sfx])

    # ------------------------------------------------------------------ one iteration
    def _conv(self, srcs: List[torch.Tensor], kh, kw, key, cout, relu=True, scale=1.0, out=None,
              epi=EPI_LINEAR, h=None, z=None, rh=None, workspace=True, residual=None, cout_active=0, cout_split=0):
        B, H, W = self._shape
        prof = self.profile
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # a stream-K workspace belongs to ONE stream: `workspace` = True (the main chain's), a branch's own tensor, or False
        # (none: plain tile grid)
        ws = self.workspace if workspace is True else (workspace if isinstance(workspace, torch.Tensor) else None)
        if self.b16:
            # K8b: z|r reads h from the bf16 twin (r * h is rounded to bf16 anyway), q updates the fp32 h and mirrors it
            hb = self.hxb[:, : self.spec.hidden] if epi != EPI_LINEAR else None
            self.ops.conv2d_b16(srcs, B, H, W, kh, kw, self.w[key + ".w"], self.w.get(key + ".b"), cout, epi, relu, scale,
                                out, h if epi == EPI_GRU_Q else None, hb, z, rh, residual)
        else:
            self.ops.conv2d(srcs, B, H, W, kh, kw, self.w[key + ".w"], self.w.get(key + ".b"), cout, epi, relu, scale,
                            out, h, z, rh, ws, residual, 1, False, cout_active, cout_split)
        if prof is not None:
            e1.record()
            if cout_active:
                key, cout = key + "_half", cout_active
                self._real_cin[key] = self._real_cin[key[:-5]]
            prof.setdefault(key, []).append((e0, e1))
            # algorithmic work: 2 * pixels * cout * taps * real input channels (padding is not work)
            self.flops[key] = 2.0 * B * H * W * cout * kh * kw * self._real_cin[key]
            # algorithmic HBM bytes: every input channel read once, every output written once, the weight once (fp32)
            self.bytes[key] = (2.0 if self.b16 else 4.0) * (B * H * W * (self._real_cin[key] + cout) + cout * kh * kw * self._real_cin[key])

    def motion(self, corr_pm: torch.Tensor) -> None:
        """BasicMotionEncoder / SmallMotionEncoder (update.py:104-112 / :85-91) into its hx slice; `flow` must already be in hx."""
        self.motion_corr(corr_pm)
        self.motion_flow()
        self.motion_join()

    def motion_corr(self, corr_pm: torch.Tensor) -> None:
        """correlation branch: convc1 [-> convc2] into corflo[:, :c] (update.py:105-106 / :86)"""
        s = self.spec
        if s.c2:
            self._conv([corr_pm], 1, 1, "c1", s.c1, out=self.cor1)
            self._conv([self.cor1], 3, 3, "c2", s.c2, out=self.corflo[:, : s.c2])
        else:
            self._conv([corr_pm], 1, 1, "c1", s.c1, out=self.corflo[:, : s.c1])

    def motion_flow(self, branch: bool = False) -> None:
        """flow branch: convf1 -> convf2 into corflo[:, c:] (update.py:107-108 / :87-88); reads only the flow slice of hx, so it
        may run next to the lookup and the correlation branch (`branch=True`: on another stream, with its own workspace)."""
        s = self.spec
        B, H, W = self._shape
        cor_c = s.c2 if s.c2 else s.c1
        self.ops.conv_cin2(self.flow_view, self.w["f1.w"], self.w["f1.b"], self.flo1, B, H, W, 7, True)
        self._conv([self.flo1], 3, 3, "f2", s.f2, out=self.corflo[:, cor_c: cor_c + s.f2],
                   workspace=self.workspace_flow if branch else True)

    @property
    def can_group(self) -> bool:
        """the grouped launch exists for the fp32 tile kernel (`pfk_conv2d_group_f32`)"""
        return self.nsplit == 0 and not self.b16

    def motion_grouped(self, corr_pm: torch.Tensor, prev_mask: bool = False) -> None:
        """The motion encoder with convc1 | convf2 [| the PREVIOUS iteration's mask conv2] as ONE grouped launch
        (`pfk_conv2d_group_f32`; update.py:105-108: `cor` and `flo` only meet in `conv`; :152: the mask head reads `net` only, i.e. the
        mask half of `fm` that the previous `heads_conv1` left).  The small-batch schedule: at 7040 pixels these launches have 440 /
        110 / 990 tiles for 256 CUs and each pays its own tail.  Same tiles, same K order, same bits as `motion()` + `mask_head()`."""
        s = self.spec
        B, H, W = self._shape
        self.ops.conv_cin2(self.flow_view, self.w["f1.w"], self.w["f1.b"], self.flo1, B, H, W, 7, True)
        cor_c = s.c2 if s.c2 else s.c1
        srcs = [corr_pm, self.flo1]
        ks = [1, 3]
        keys = ["c1", "f2"]
        relu, scale = [1, 1], [1.0, 1.0]
        outs = [self.cor1 if s.c2 else self.corflo[:, : s.c1], self.corflo[:, cor_c: cor_c + s.f2]]
        if prev_mask:
            srcs.append(self.fm[:, s.fh_hidden:]); ks.append(1); keys.append("mk"); relu.append(0); scale.append(0.25); outs.append(self.mask)
        prof = self.profile
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.ops.conv2d_group(srcs, B, H, W, ks, [self.w[k + ".w"] for k in keys],
                              [self.w[k + ".b"] for k in keys], relu, scale, outs, [])
        if prof is not None:
            e1.record()
            key = "+".join(keys)
            prof.setdefault(key, []).append((e0, e1))
            M = B * H * W
            self.flops[key] = sum(2.0 * M * o.shape[1] * k * k * self._real_cin[kk] for o, k, kk in zip(outs, ks, keys))
            self.bytes[key] = sum(4.0 * (M * (self._real_cin[kk] + o.shape[1]) + o.shape[1] * k * k * self._real_cin[kk])
                                  for o, k, kk in zip(outs, ks, keys))
        if s.c2:
            self._conv([self.cor1], 3, 3, "c2", s.c2, out=self.corflo[:, : s.c2])
        self.motion_join()

    def motion_join(self) -> None:
        """conv over cat([cor, flo]) into the motion slice of hx (update.py:110-112 / :89-91)"""
        s = self.spec
        o = s.hidden + s.context
        self._conv([self.corflo], 3, 3, "cv", s.enc_out, out=self.act[:, o: o + s.enc_out])

    @property
    def motion_view(self):
        """encoder out | flow: the reference's `motion_features` (update.py:112), a slice of hx"""
        s = self.spec
        o = s.hidden + s.context
        return self.hx[:, o: o + s.motion_channels]

    @property
    def aggregate_view(self):
        """the second motion-feature slot (GMA: fmap + gamma * attn @ v; CCMR: the caller's XCiT output)"""
        s = self.spec
        o = s.hidden + s.context + s.motion_channels
        return self.hx[:, o: o + s.motion_channels]

    def gru(self) -> None:
        """SepConvGRU / ConvGRU passes (update.py:58-73 / :24-32) over hx = [h | x], h updated in place."""
        s = self.spec
        Ch = s.hidden
        act = self.act
        if self.hoist_context:
            rest = act[:, Ch + s.context:]          # x without the context slice: motion features [| aggregated ones] | pad
            for kh, kw, sfx in s.gru_passes:
                self._conv([act[:, :Ch], rest], kh, kw, "zr" + sfx, 2 * Ch, epi=EPI_GRU_ZR, h=self.h_view, z=self.zbuf, rh=self.rh,
                           residual=self.ctx["zr" + sfx])
                self._conv([self.rh, rest], kh, kw, "q" + sfx, Ch, epi=EPI_GRU_Q, h=self.h_view, z=self.zbuf,
                           residual=self.ctx["q" + sfx])
            return
        for kh, kw, sfx in s.gru_passes:
            self._conv([act], kh, kw, "zr" + sfx, 2 * Ch, epi=EPI_GRU_ZR, h=self.h_view, z=self.zbuf, rh=self.rh)
            self._conv([self.rh, act[:, Ch:]], kh, kw, "q" + sfx, Ch, epi=EPI_GRU_Q, h=self.h_view, z=self.zbuf)

    def motion_and_gru(self, corr_pm: torch.Tensor, grouped: bool = False, prev_mask: bool = False) -> None:
        """update.py:104-112 (encoder) + :58-73 / :24-32 (GRU); `flow` must already be in hx.  `grouped`: `motion_grouped`
        (with `prev_mask` the previous iteration's mask conv2 rides in its grouped launch: `self.mask` is valid once this returns)."""
        s = self.spec
        if grouped:
            self.motion_grouped(corr_pm, prev_mask)
        else:
            self.motion(corr_pm)
        if s.aggregate:
            if s.external_aggregate:
                raise RuntimeError("this block's aggregate slot is filled by the caller: use motion(), aggregate_view, gru()")
            if self.attn is None:
                raise RuntimeError("GMA update block needs set_attention() before the first iteration")
            self._aggregate()
        self.gru()

    def heads(self, coords0: torch.Tensor, coords1: torch.Tensor, delta_out: Optional[torch.Tensor],
              want_mask: bool = True, write_flow: bool = True, mask_conv2: bool = True) -> None:
        """update.py:13-14 + :152 and the coordinate bookkeeping of raft.py:174,178.  `mask_conv2=False`: the mask head's hidden
        activation is computed, its second convolution is left to the caller (`mask_upsample`, the fused kernel)."""
        s = self.spec
        self.heads_conv1(want_mask)
        self.flow_delta(coords0, coords1, delta_out, write_flow)
        if s.has_mask and want_mask and mask_conv2:
            self.mask_head()

    def heads_conv1(self, want_mask: bool = True) -> None:
        """flow-head conv1 | mask conv1 as ONE GEMM over h (update.py:13, :138-139) -> fm; without `want_mask` only the flow-head
        half is computed (`cout_active`, include/pfk.h): the launch keeps the full launch's tile schedule — at batch 1 the stream-K
        split points of the 880-tile grid — and skips the mask half's column tiles, so the flow half has the full launch's bits
        whatever the schedule (round 4 could only drop the mask half where neither form fell into the stream-K window)."""
        s = self.spec
        if s.has_mask and self.nsplit == 0 and s.fh_hidden % 64 == 0:
            # `cout_split`: a stream-K schedule (batch 1) walks the column tiles of the two halves interleaved — in BOTH forms of the
            # launch, so that they share their split points (same bits) and the half launch still balances its blocks
            self._conv([self.h_view], 3, 3, "fm", 2 * s.fh_hidden, out=self.fm, cout_active=0 if want_mask else s.fh_hidden,
                       cout_split=s.fh_hidden)
        elif s.has_mask and not want_mask and (self.b16 or self._shape[0] * self._shape[1] * self._shape[2] >= 28160):
            # split-bf16 / K8b kernels (no `cout_active`): the flow-head-only weights, where no launch is in the stream-K window
            self._conv([self.act[:, : s.hidden]], 3, 3, "fh", s.fh_hidden, out=self.fm[:, : s.fh_hidden])
        else:
            self._conv([self.act[:, : s.hidden]], 3, 3, "fm", s.fh_hidden * (2 if s.has_mask else 1), out=self.fm)

    def flow_delta(self, coords0: torch.Tensor, coords1: torch.Tensor, delta_out: Optional[torch.Tensor] = None,
                   write_flow: bool = True) -> None:
        """flow-head conv2 + `coords1 += delta`, `flow = coords1 - coords0` (update.py:14, raft.py:174,178) on the flow half of fm"""
        s = self.spec
        if self.b16:      # bf16 hidden activation in; the flow goes to hx (fp32: 7x7 convolution, upsampling) and to its bf16 twin
            o = s.hidden + s.context + s.enc_out
            self.ops.flow_delta(self.fm[:, : s.fh_hidden], self.w["fh2.w"], self.w["fh2.b"], coords0, coords1, delta_out,
                                self.flow_view if write_flow else None, self.hxb[:, o: o + 2] if write_flow else None)
            return
        self.ops.flow_delta(self.fm[:, : s.fh_hidden], self.w["fh2.w"], self.w["fh2.b"], coords0, coords1, delta_out,
                            self.flow_view if write_flow else None)

    def mask_head(self, side_stream: bool = False) -> None:
        """mask[1] + the 0.25 scale of raft/update.py:131-135,152 on the mask half of `fm` (written by `heads_conv1`); reads nothing
        else of the iteration's state, so it may run on a side stream next to the following iteration (raft.py `_iterate`)."""
        s = self.spec
        self._conv([self.fm[:, s.fh_hidden:]], 1, 1, "mk", s.mask_channels, relu=False, scale=0.25, out=self.mask,
                   workspace=self.workspace_mask if side_stream else True)

    @property
    def can_fuse_mask_upsample(self) -> bool:
        return "mku.w" in self.w

    def mask_upsample(self, flow_up: torch.Tensor) -> None:
        """raft/update.py:152 (`0.25 * mask(net)`, its second convolution) + raft/raft.py:112-123 (`upsample_flow`) in ONE launch:
        reads the mask half of `fm` (written by `heads_conv1`) and the flow slice of hx, writes the 8x flow; the [M, 576] mask is
        never materialised.  Bit-identical to `mask_head()` + `convex_upsample_pm` (tests/test_gpu_kernels.py)."""
        s = self.spec
        key = "mku"
        x = self.fm[:, s.fh_hidden:]
        if self.profile is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        self.ops.mask_upsample(x, self.w["mku.w"], self.w["mku.b"], 0.25, self.flow_view, flow_up)
        if self.profile is not None:
            b.record()
            self.profile.setdefault(key, []).append((a, b))
            M = x.shape[0]
            self.flops[key] = 2.0 * M * 576 * s.fh_hidden
            eb = 2.0 if self.b16 else 4.0       # activation / weight element size; flow in and 8x flow out are fp32 on both paths
            self.bytes[key] = eb * (M * s.fh_hidden + 576 * s.fh_hidden) + 4.0 * (2 * M + 2 * 64 * M)

    def step(self, corr_pm: torch.Tensor, coords0: torch.Tensor, coords1: torch.Tensor, want_mask: bool = True) -> None:
        """One full RAFT iteration body after the lookup; updates hx (net, flow) and coords1 in place."""
        self.motion_and_gru(corr_pm)
        self.heads(coords0, coords1, None, want_mask=want_mask)

    # ------------------------------------------------------------------ NCHW views of the state
    def nchw_view(self, pm: torch.Tensor) -> torch.Tensor:
        B, H, W = self._shape
        return pm.view(B, H, W, pm.shape[1]) if pm.is_contiguous() else pm.unflatten(0, (B, H, W))

    def net_nchw(self) -> torch.Tensor:
        B, H, W = self._shape
        return self.h_view.unflatten(0, (B, H, W)).permute(0, 3, 1, 2)

    def mask_nchw(self) -> torch.Tensor:
        B, H, W = self._shape
        return self.mask.float().view(B, H, W, self.spec.mask_channels).permute(0, 3, 1, 2)


class PfkUpdateBlock(torch.nn.Module):
    """Drop-in for ``model.update_block`` (seam B3): same ``forward(net, inp, corr, flow)`` ->
    ``(net, mask, delta_flow)`` contract as raft/update.py:144-153 / :122-128, same parameters.

    The tensors handed back are views of the engine's buffers (net: channels-last strides); they are
    overwritten by the next call, which is how the reference loop uses them (raft.py:169-187)."""

    def __init__(self, ref_block: torch.nn.Module, spec: UpdateSpec, conv_precision: str = "fp32"):
        super().__init__()
        self.conv_precision = conv_precision
        # keep the reference sub-modules so state_dict keys/checkpoints/optimizers are unchanged
        for name, child in ref_block.named_children():
            self.add_module(name, child)
        self._ref = [ref_block]  # in a list: not registered twice in the module tree
        self.training = ref_block.training
        self.native_backward = True   # False: calls that need a gradient graph go to the wrapped reference module
        self._train_cache: dict = {}  # packed weights of the training path, keyed by parameter versions
        self.spec = spec
        self._engine: Optional[UpdateEngine] = None
        self._versions = None
        self._inp_ref, self._inp_version = None, -1      # strong references: the address cannot be recycled while cached
        self._attn_ref, self._attn_version = None, -1
        self._skip = None      # patch._DeadWorkSkip when `accelerate(model)` found the loop's dead work provably dead (patch.py)
        self._last_out = None  # the `net` tensor handed back by the previous call: the same OBJECT coming in marks the next iteration

    def _param_versions(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _get_engine(self, device) -> UpdateEngine:
        v = self._param_versions()
        if self._engine is None or self._engine.device != device:
            self._engine = UpdateEngine(dict(self.named_parameters()), self.spec, device, self.conv_precision)
            self._versions = v
        elif v != self._versions:
            self._engine.pack(dict(self.named_parameters()))
            self._versions = v
        return self._engine

    def forward(self, net, inp, corr, flow, *extra, **kw):
        """raft: (net, inp, corr, flow); gma: + attention (gma/update.py:148); ccmr: + global_context, level_index
        (ccmr/update.py:152); ms_raft_plus: + coords_x, coords_y, only None supported (stack_coords=False)."""
        s = self.spec
        if s.external_aggregate:
            names = ("global_context", "level_index")
        elif s.aggregate:
            names = ("attention",)
        else:
            names = ("coords_x", "coords_y")
        args = dict(zip(names, extra))
        args.update(kw)
        if not s.aggregate and any(v is not None for v in args.values()):
            return self._ref[0](net, inp, corr, flow, *extra, **kw)     # a variant the kernels do not implement
        attention = args.get("attention")
        if torch.is_grad_enabled() and (net.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training graph: every convolution forward / dgrad / wgrad on the MFMA kernel through the
            # differentiable composition of ptlflow_amd/train.py; GMA's aggregate branch, CPU tensors and split-bf16
            # modes keep the reference module's own autograd
            if (self.native_backward and net.is_cuda and not self.spec.aggregate and self.conv_precision == "fp32"
                    and self.spec.mask_channels == 576):
                from .train import update_block_train
                h, mask, delta = update_block_train(dict(self.named_parameters()), self.spec, net, inp, corr, flow, self._train_cache)
                return h.to(net.dtype), (None if mask is None else mask.to(net.dtype)), delta.to(net.dtype)
            return self._ref[0](net, inp, corr, flow, *extra, **kw)
        with torch.no_grad():
            return self._forward_kernels(net, inp, corr, flow, attention, args.get("global_context"), args.get("level_index", 0))

    def _forward_kernels(self, net, inp, corr, flow, attention=None, global_context=None, level_index=0):
        if not net.is_cuda:
            raise RuntimeError("PfkUpdateBlock needs GPU tensors (no CPU fallback)")
        eng = self._get_engine(net.device)
        B, _, H, W = net.shape
        eng.bind(B, H, W)
        ops = eng.ops
        # state in.  `net` that already *is* our buffer (the view handed back by the previous call) marks the second and later
        # iterations of one forward: nothing to copy.  Anything else starts a new forward: `net`, `inp` (and GMA's attention)
        # are (re)loaded unconditionally — the reference builds them as fresh tensors per forward (raft.py:158-160) and the
        # caching allocator hands the next forward the same addresses, so neither data_ptr nor _version can tell two
        # forwards apart.  Within a forward the loop passes the same `inp` object every iteration: identity + _version.
        # (a half / bf16 model gets `net` back as a fresh cast of the buffer: object identity tells its iterations apart)
        new_forward = not (net is self._last_out or (net.dtype == torch.float32 and net.data_ptr() == eng.hx.data_ptr()))
        if new_forward:
            eng.watch_faults()
            ops.nchw_to_pm(net.float().contiguous(), eng.h_view)
            eng.state_changed()
            if self._skip is not None:
                self._skip.begin_forward(fp32=net.dtype == torch.float32)
        # §8 f2: on the non-final iterations of an eval forward nobody reads the mask (patch._DeadWorkSkip)
        dead_mask = self._skip is not None and eng.spec.has_mask and self._skip.next_call()
        if new_forward or inp is not self._inp_ref or inp._version != self._inp_version:
            ops.nchw_to_pm(inp.float().contiguous(), eng.inp_view)
            self._inp_ref, self._inp_version = inp, inp._version
            eng.state_changed()
            eng.prepare_context()      # the loop-invariant part of the GRU pre-activations, once per (forward, scale)
        # corr: a channels-last view of a [M, C] buffer (what ptlflow_amd.CorrBlock returns) is used as is
        cpm = corr.permute(0, 2, 3, 1)
        C = corr.shape[1]
        padded = getattr(corr, "_pfk_padded_pm", None)
        if padded is not None and padded.shape[0] == B * H * W:
            corr_pm = padded                                   # [M, round_up(C, 4)], pad columns zero (CorrBlock's own buffer)
        elif corr.dtype == torch.float32 and cpm.is_contiguous() and C % 4 == 0:
            corr_pm = cpm.reshape(B * H * W, C)
        else:
            corr_pm = torch.zeros(B * H * W, round_up(C, 4), device=net.device, dtype=torch.float32)
            ops.nchw_to_pm(corr.float().contiguous(), corr_pm[:, :C])
        ops.nchw_to_pm(flow.float().contiguous(), eng.flow_view)
        eng.flow_changed()
        if eng.b16:      # K8b reads a bf16 lookup result (a CorrBlock that writes it directly is the mirror's path, ptlflow_amd/raft.py)
            eng.corr16[:, :C].copy_(corr_pm[:, :C])
            corr_pm = eng.corr16
        if eng.spec.external_aggregate:
            # CCMR (ccmr/update.py:152-163): motion encoder on the kernels, the scale's XCiT block — the reference's own module,
            # torch — on the NCHW view of the motion features, its output into the aggregate slot of hx, then GRU + heads
            eng.motion(corr_pm)
            mf = eng.motion_view.unflatten(0, (B, H, W)).permute(0, 3, 1, 2)
            mfg = self._ref[0].aggregator[level_index](global_context, mf)
            ops.nchw_to_pm(mfg.float().contiguous(), eng.aggregate_view)
            eng.gru()
        else:
            if eng.spec.aggregate:
                if attention is None:
                    raise RuntimeError("GMAUpdateBlock.forward needs the attention map (gma/update.py:148)")
                if new_forward or eng.attn is None or attention is not self._attn_ref or attention._version != self._attn_version:
                    eng.set_attention(attention)
                    self._attn_ref, self._attn_version = attention, attention._version
            # small batches: convc1 | convf2 as one grouped launch (`motion_grouped`; same tiles, same bits)
            eng.motion_and_gru(corr_pm, grouped=eng.can_group and B * H * W < 28160)
        if new_forward:
            eng._scratch_c1.zero_()     # the kernel's `coords1 += delta` lands here; `delta` itself does not depend on it — once per forward keeps it bounded
        # a live call of an armed forward leaves mask conv2 to seam B5, which runs it fused with the softmax and the upsampling (K13)
        defer = (self._skip is not None and eng.spec.has_mask and not dead_mask and eng.can_fuse_mask_upsample
                 and net.dtype == torch.float32 and self._skip.may_defer())
        eng.heads(eng._scratch_c0, eng._scratch_c1, eng._delta, want_mask=not dead_mask, write_flow=False, mask_conv2=not defer)
        mask = eng.mask_nchw() if eng.spec.has_mask else None
        if defer:
            self._skip.defer(eng, mask)
        out_net = eng.net_nchw()
        delta = eng._delta
        if net.dtype != torch.float32:
            out_net, delta = out_net.to(net.dtype), delta.to(net.dtype)
            mask = mask.to(net.dtype) if mask is not None else None
        self._last_out = out_net
        return out_net, mask, delta
