"""Seam B1 — drop-in for ``get_corr_block`` / ``CorrBlock`` (ptlflow/models/raft/corr.py:12-64,104-118;
identical copies in gma/corr.py, ccmr/corr.py, ms_raft_plus/corr.py; sea_raft/corr.py:71-117 differs
only in how the pyramid is built).

Same call contract as the reference: built once per forward from ``fmap1, fmap2 [B,D,h,w]``, then
called ``iters`` times with ``coords [B,2,h,w]`` (x, y in pixels) and returns
``[B, L*(2r+1)^2, h, w]`` with channel ``l*(2r+1)^2 + (dx+r)*(2r+1) + (dy+r)``.

MI355X layout: the volume is built by one fp32-MFMA GEMM launch (K1) straight into per-source-pixel
maps, pooled by K2, and every lookup is ONE launch over all levels (K3) that
writes a pixel-major ``[B*N, C]`` buffer; the tensor handed back is a channels-last *view* of that
buffer (shape and values as the reference, strides NHWC) so the update block's first 1x1
convolution reads it without any transpose.

Inference keeps the maps in the BLOCKED layout (include/pfk.h): 4 x 8-element tiles, one 128-byte line each, so that the
12 x 12 window a lookup stages covers ~9 lines instead of ~16.  K1 writes it for free — the target feature map's rows are
permuted into the blocked order once per forward (`pfk_fmap_to_blocked_f32`) and a GEMM does not care in which order its B rows
come — K2 / K3 have blocked variants with the same arithmetic (bit-identical lookups).  The training graph keeps the row-major
maps its backward kernels address.  ``corr_pyramid`` always presents the reference's ``[B*N, h_l, w_l]`` maps.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import List, Optional

import torch
import torch.nn.functional as F

from . import load_native


def _ops():
    load_native()
    return torch.ops.pfk


def to_pixel_major(x: torch.Tensor) -> torch.Tensor:
    """NCHW (any strides) -> contiguous ``[B, H*W, C]``; free when ``x`` is already channels-last."""
    B, C, H, W = x.shape
    if x.dtype != torch.float32:
        x = x.float()
    nhwc = x.permute(0, 2, 3, 1)
    if nhwc.is_contiguous():
        return nhwc.reshape(B, H * W, C)
    x = x.contiguous()
    out = torch.empty(B * H * W, C, device=x.device, dtype=torch.float32)
    _ops().nchw_to_pm(x, out)
    return out.view(B, H * W, C)


class _PyramidToken(torch.autograd.Function):
    """Builds the pyramid (no graph inside) and returns a 1-element token every lookup of this block depends on.  Autograd
    therefore runs this node's backward after ALL lookup backwards of the step — which only scatter into the block's level
    gradient buffers — and turns those buffers into the two feature-map gradients with two GEMMs per level."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, block):
        block._build(fmap1.detach(), fmap2.detach())
        block._zero_grad_levels()
        # weak: block -> _token -> grad_fn (this ctx) -> block would be a reference cycle that keeps a step's N x N pyramid and
        # its equally large gradient buffers alive until the cyclic GC runs.  Every lookup node holds the block strongly, and
        # this backward only runs when a lookup's gradient reaches the token, so the block is alive whenever it is needed.
        ctx.block_ref = weakref.ref(block)
        ctx.save_for_backward(fmap1, fmap2)
        return fmap1.new_zeros(1, dtype=torch.float32)

    @staticmethod
    def backward(ctx, _g):
        fmap1, fmap2 = ctx.saved_tensors
        block = ctx.block_ref()
        if block is None:       # no lookup node of this block is left in the graph: nothing was scattered
            return torch.zeros_like(fmap1), torch.zeros_like(fmap2), None
        d1, d2 = block._volume_backward(fmap1, fmap2)
        for g in block.grad_levels:     # consumed: a second backward over a retained graph starts from zero again
            g.zero_()
        return d1, d2, None


class _LookupFn(torch.autograd.Function):
    """One pyramid lookup -> pixel-major [B*h*w, C]; backward adds d(out) into the block's level-gradient buffers through the
    forward's bilinear weights (`pfk_corr_lookup_bwd_f32`).  `coords` is detached by every caller (raft.py:171)."""

    @staticmethod
    def forward(ctx, token, coords, block):
        out = torch.empty(block.B * block.h * block.w, block.channels, device=coords.device, dtype=torch.float32)
        c = coords.detach().float().contiguous()
        _ops().corr_lookup(block._levels, c, block.radius, out)
        ctx.block, ctx.coords = block, c
        return out

    @staticmethod
    def backward(ctx, g):
        b = ctx.block
        g = g.float()
        if g.dim() != 2 or g.stride(1) != 1:
            g = g.reshape(-1, b.channels).contiguous()
        _ops().corr_lookup_bwd(b.grad_levels, b._lvl_h, b._lvl_w, ctx.coords, b.radius, g)
        return torch.zeros(1, device=g.device, dtype=torch.float32), None, None


class CorrBlock:
    """All-pairs correlation pyramid + radius-r lookup on the GPU.  ``coords`` is detached by every caller (raft.py:171), so
    no coordinate gradient exists; when a feature map requires grad (training) the block records an autograd graph whose
    backward runs on libpfk too (`_PyramidToken` / `_LookupFn`) and delivers the gradients of ``fmap1`` / ``fmap2``."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4,
                 pyramid: str = "avgpool", volume_dtype: Optional[torch.dtype] = None, channels_last: bool = True,
                 layout: Optional[str] = None):
        """``layout``: ``"blocked"`` (4 x 8-element tiles, module docstring; the default without a gradient graph) or ``"rowmajor"``
        (the reference's ``[B*N, h_l, w_l]`` maps; what the training graph's backward kernels address, forced there).

        ``channels_last``: the tensor a lookup returns is a channels-last VIEW of the pixel-major buffer (what this package's
        update block reads without a transpose).  ``False`` returns a plain contiguous NCHW tensor instead — for callers whose
        consumer is torch's own convolutions: a channels-last input makes PyTorch run (and propagate) the channels-last memory
        format through the caller's whole update block, which costs the reference's SKFlow 25 ms per forward on MIOpen
        (profiles/r04_f_dropin_speedup.md); one 9 MB transpose per lookup is cheap next to that.

        ``volume_dtype``: storage type of the pyramid — ``torch.float32`` (exact fp32 products on the fp32 matrix cores, the
        parity path) or ``torch.bfloat16`` (bf16 operands and a bf16 volume: what the reference's matmul yields under
        ``torch.autocast(bfloat16)``; half the HBM bytes).  Default: bf16 only when the feature maps arrive in bfloat16 (the
        caller runs under bf16 autocast), fp32 otherwise — in particular for float16 maps (the reference's own reduced-precision
        mode is ``model.half()``, validate.py:243-244 / model_benchmark.py:317-319): fp16 operands are exact in the fp32 volume,
        rounding them to bf16 would lose 3 mantissa bits the reference's fp16 matmul keeps.  The lookup's arithmetic is fp32
        either way; the result is cast to the maps' dtype like the reference's."""
        if not fmap1.is_cuda:
            raise RuntimeError("ptlflow_amd.CorrBlock needs GPU tensors (no CPU fallback)")
        if not 1 <= radius <= 4:
            raise RuntimeError("radius must be in 1..4")
        _ops()
        self.num_levels = num_levels
        self.radius = radius
        self.channels_last = channels_last
        self.pyramid_mode = pyramid
        if pyramid not in ("avgpool", "bilinear_f2"):
            raise ValueError(f"unknown pyramid mode {pyramid!r}")
        n = 2 * radius + 1
        self.channels = num_levels * n * n
        # row stride of the pixel-major lookup buffer: the convolution that consumes it wants a multiple of 4 floats
        # (L = 2 levels of radius 4 give 162 channels: ccmr / ms_raft_plus); the pad columns stay zero
        self.cpad = (self.channels + 3) // 4 * 4
        self._out: Optional[torch.Tensor] = None
        self._levels: List[torch.Tensor] = []      # storage of the pyramid levels in `self.layout`
        self._lvl_hw: List[tuple] = []             # logical (h_l, w_l) of every level
        self._shape = None
        self.grad_levels: List[torch.Tensor] = []
        self._token: Optional[torch.Tensor] = None
        needs_graph = torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad)
        if volume_dtype is None:
            volume_dtype = torch.bfloat16 if fmap1.dtype == torch.bfloat16 and not needs_graph else torch.float32
        if volume_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("volume_dtype must be torch.float32 or torch.bfloat16")
        if needs_graph and volume_dtype != torch.float32:
            raise RuntimeError("the training path keeps the pyramid in fp32 (the backward kernels are fp32)")
        self.volume_dtype = volume_dtype
        if layout is None:
            # (PFK_VOLUME_LAYOUT: A/B knob of bench.py / the tuning scripts; a gradient graph always gets row-major maps)
            layout = "rowmajor" if needs_graph else os.environ.get("PFK_VOLUME_LAYOUT", "blocked")
        if layout not in ("blocked", "rowmajor"):
            raise ValueError(f"unknown volume layout {layout!r}")
        if needs_graph and layout != "rowmajor":
            raise RuntimeError("the training path keeps the row-major pyramid (its backward kernels address [h_l][w_l] maps)")
        self.layout = layout
        if needs_graph:
            self._token = _PyramidToken.apply(fmap1, fmap2, self)     # calls _build
        else:
            self.update(fmap1, fmap2)

    def update(self, fmap1: torch.Tensor, fmap2: torch.Tensor) -> "CorrBlock":
        """(Re)build the pyramid for a new frame pair (inference).  Feature maps of the same shape as last time are written
        into the SAME level tensors (same addresses), which is what lets the iteration loop live in a captured hipGraph."""
        self._token = None
        with torch.no_grad():
            return self._build(fmap1, fmap2)

    def _build(self, fmap1: torch.Tensor, fmap2: torch.Tensor) -> "CorrBlock":
        ops = _ops()
        self._rowmajor_cache = None
        self.out_dtype = fmap1.dtype
        B, D, h, w = fmap1.shape
        self.B, self.h, self.w = B, h, w
        N = h * w
        f1 = to_pixel_major(fmap1)
        scale = 1.0 / math.sqrt(D)
        shape = (B, D, h, w, tuple(fmap2.shape), fmap1.device)
        fresh = shape != self._shape
        if fresh:
            self._levels, self._lvl_hw, self._out, self._shape = [], [], None, shape
        dev = f1.device
        vt = self.volume_dtype
        bf = vt == torch.bfloat16

        f1b = f1.to(torch.bfloat16) if bf else None    # exact when the maps already hold bf16 values (an autocast encoder)

        def volume(f2_pm: torch.Tensor, out: torch.Tensor) -> None:
            if bf:    # bf16 operands, fp32 accumulate, bf16 volume
                ops.corr_volume_bf16(f1b, f2_pm.to(torch.bfloat16), scale, out)
            else:
                ops.corr_volume(f1, f2_pm, scale, out)

        blocked = self.layout == "blocked"

        def level_storage(hl: int, wl: int) -> torch.Tensor:
            if blocked:
                return torch.empty(B * N, ops.blocked_map_elems(hl, wl), device=dev, dtype=vt)
            return torch.empty(B * N, hl, wl, device=dev, dtype=vt)

        def target_rows(f2_pm: torch.Tensor, h2: int, w2: int) -> torch.Tensor:
            """[B, rows, D] B operand of K1 for a (h2, w2) target map: the map's pixels, in the blocked order when the volume is"""
            if not blocked:
                return f2_pm.view(B, h2 * w2, D)
            out = torch.empty(B * ops.blocked_map_elems(h2, w2), D, device=dev, dtype=torch.float32)
            ops.fmap_to_blocked(f2_pm.reshape(B * h2 * w2, D), out, B, h2, w2)
            return out.view(B, -1, D)

        if self.pyramid_mode == "avgpool":  # raft/corr.py:19-27
            h2, w2 = fmap2.shape[-2:]
            f2 = target_rows(to_pixel_major(fmap2), h2, w2)
            if fresh:
                hl, wl = h2, w2
                for _ in range(self.num_levels):
                    self._levels.append(level_storage(hl, wl))
                    self._lvl_hw.append((hl, wl))
                    hl, wl = hl // 2, wl // 2
            volume(f2, self._levels[0].view(B, N, -1))
            for l in range(1, self.num_levels):
                if blocked:
                    ops.corr_pool2x2_blocked(self._levels[l - 1], self._levels[l], *self._lvl_hw[l - 1])
                else:
                    ops.corr_pool2x2(self._levels[l - 1], self._levels[l])
        else:  # sea_raft/corr.py:77-84: one GEMM per level against fmap2 halved bilinearly (== a 2x2 average, pfk_fmap_pool2x2_f32)
            f2 = to_pixel_major(fmap2).reshape(B * fmap2.shape[-2] * fmap2.shape[-1], D)
            h2, w2 = fmap2.shape[-2:]
            for l in range(self.num_levels):
                if l > 0:
                    nxt = torch.empty(B * (h2 // 2) * (w2 // 2), D, device=dev, dtype=torch.float32)
                    ops.fmap_pool2x2(f2, nxt, B, h2, w2)
                    f2, h2, w2 = nxt, h2 // 2, w2 // 2
                if fresh:
                    self._levels.append(level_storage(h2, w2))
                    self._lvl_hw.append((h2, w2))
                volume(target_rows(f2, h2, w2), self._levels[l].view(B, N, -1))
        return self

    @property
    def corr_pyramid(self) -> List[torch.Tensor]:
        """The pyramid as the reference holds it (raft/corr.py:21-27): ``[B*N, h_l, w_l]`` maps.  Row-major storage is returned as
        is; blocked storage is un-tiled ONCE per build into copies that later accesses share (tests / inspection — the lookups read
        the blocked storage, so writes into the returned tensors do not reach them: treat the result as read-only)."""
        if self.layout != "blocked":
            return self._levels
        if self._rowmajor_cache is None:       # one un-tiling per build (dropped by `_build`); the result is a READ-ONLY copy
            out = []
            for lv, (hl, wl) in zip(self._levels, self._lvl_hw):
                th, tw = (hl + 3) // 4, (wl + 7) // 8
                out.append(lv.view(-1, th, tw, 4, 8).permute(0, 1, 3, 2, 4).reshape(-1, th * 4, tw * 8)[:, :hl, :wl].contiguous())
            self._rowmajor_cache = out
        return self._rowmajor_cache

    # ------------------------------------------------------------------ training (autograd) side
    def _zero_grad_levels(self) -> None:
        """Gradient buffers of the pyramid levels: one [h_l][w_l] map per source pixel at a row stride padded to a multiple
        of 4 floats (so a buffer is directly the A operand / dY operand of the two backward GEMMs); zeroed once per step."""
        M = self.B * self.h * self.w
        sizes = list(self._lvl_hw)
        if len(self.grad_levels) != len(sizes) or any(g.shape[0] != M or g.shape[1] != (hl * wl + 3) // 4 * 4
                                                      for g, (hl, wl) in zip(self.grad_levels, sizes)):
            dev = self._levels[0].device
            self.grad_levels = [torch.zeros(M, max(4, (hl * wl + 3) // 4 * 4), device=dev, dtype=torch.float32) for hl, wl in sizes]
        else:
            for g in self.grad_levels:
                g.zero_()
        self._lvl_h, self._lvl_w = [s[0] for s in sizes], [s[1] for s in sizes]

    def _f2_levels(self, f2: torch.Tensor) -> List[torch.Tensor]:
        """The per-level target feature maps the volume levels are correlations with (raft/corr.py:25-27: pooling the volume's
        target dims == correlating with the pooled map; sea_raft/corr.py:81-83: bilinear halving)."""
        out = [f2]
        for _ in range(1, self.num_levels):
            prev = out[-1]
            if self.pyramid_mode == "avgpool":
                out.append(F.avg_pool2d(prev, 2, stride=2))
            else:
                out.append(F.interpolate(prev, scale_factor=0.5, mode="bilinear", align_corners=False))
        return out

    def _volume_backward(self, fmap1: torch.Tensor, fmap2: torch.Tensor):
        """grad_levels -> (d fmap1, d fmap2): per level  dF1 += s dC_l F2_l  (MFMA GEMM) and  dF2_l = s dC_l^T F1  (the
        transposed product on the weight-gradient kernel); the chain F2 -> F2_l (tiny feature-map pooling) by torch autograd."""
        ops = _ops()
        B, D, h, w = fmap1.shape
        N = h * w
        scale = 1.0 / math.sqrt(D)
        f1 = to_pixel_major(fmap1.detach()).contiguous()
        df1 = torch.empty(B, N, D, device=f1.device, dtype=torch.float32)
        leaf = fmap2.detach().float().requires_grad_(True)
        with torch.enable_grad():
            f2_levels = self._f2_levels(leaf)
        G = []
        first = True
        for l, f2l in enumerate(f2_levels):
            h2, w2 = f2l.shape[-2:]
            N2 = h2 * w2
            if N2 == 0 or l >= len(self.grad_levels):
                G.append(torch.zeros_like(f2l))
                continue
            ldc = self.grad_levels[l].shape[1]
            ld2cm = (ldc + 31) // 32 * 32
            f2cm = torch.zeros(B, D, ld2cm, device=f1.device, dtype=torch.float32)
            ops.pm_to_cm(to_pixel_major(f2l.detach()).reshape(B * N2, D), f2cm)
            df2 = torch.empty(B, ldc, (D + 31) // 32 * 32, device=f1.device, dtype=torch.float32)
            ops.corr_volume_bwd(self.grad_levels[l].view(B, N, ldc), N2, f1, f2cm, scale, df1, not first, df2)
            first = False
            G.append(df2[:, :N2, :D].permute(0, 2, 1).reshape(B, D, h2, w2))
        d2 = G[0]
        if len(f2_levels) > 1:
            d2 = d2 + torch.autograd.grad(f2_levels[1:], leaf, G[1:])[0]
        d1 = df1.view(B, h, w, D).permute(0, 3, 1, 2)
        return d1.to(fmap1.dtype), d2.to(fmap2.dtype)

    def lookup_pm(self, coords: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Lookup into a pixel-major ``[B*h*w, C]`` buffer (allocated once and reused unless given).  In a training graph
        every call returns a fresh tensor that autograd keeps for the backward."""
        if self._token is not None and torch.is_grad_enabled():
            return _LookupFn.apply(self._token, coords, self)
        if out is None:
            if self._out is None:
                self._out = torch.zeros(self.B * self.h * self.w, self.cpad, device=coords.device, dtype=torch.float32)
            out = self._out
        c = coords
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        if self.layout == "blocked":
            _ops().corr_lookup_blocked(self._levels, [s[0] for s in self._lvl_hw], [s[1] for s in self._lvl_hw], c, self.radius, out)
        else:
            _ops().corr_lookup(self._levels, c, self.radius, out)
        return out

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        out = self.lookup_pm(coords)
        res = out.view(self.B, self.h, self.w, out.shape[1])[..., : self.channels].permute(0, 3, 1, 2)
        if not self.channels_last:
            res = res.contiguous()
            return res if self.out_dtype == torch.float32 else res.to(self.out_dtype)
        if self.out_dtype != torch.float32:
            res = res.to(self.out_dtype)
        elif out.shape[1] != self.channels:
            res._pfk_padded_pm = out     # PfkUpdateBlock takes the zero-padded pixel-major buffer behind this view as is
        return res


class AlternateCorrBlock:
    """On-demand correlation (ptlflow/models/raft/corr.py:67-101): no N x N volume is materialised; every call computes,
    per pyramid level of ``fmap2`` (avg-pooled), the (2r+1)^2 window around ``coords / 2^l`` with the gfx950 kernel behind
    the ``alt_cuda_corr`` ABI (``pfk_altcorr_forward_f32``).  Same channel layout and ``/ sqrt(dim)`` as the reference."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4,
                 map_dtype: Optional[torch.dtype] = None):
        """``map_dtype``: element type of the feature maps the kernel gathers — ``torch.float32`` (default for fp32 / fp16 maps:
        the reference up-casts half inputs, raft/corr.py:90-96) or ``torch.bfloat16`` (default for bf16 maps, i.e. autocast
        callers; `RAFT(conv_precision="bf16", alternate_corr=True)` asks for it explicitly): `pfk_altcorr_forward_bf16`, bf16
        maps pooled in bf16 like the reference's own `F.avg_pool2d` on them, fp32 products / accumulation / output."""
        if not fmap1.is_cuda:
            raise RuntimeError("ptlflow_amd.AlternateCorrBlock needs GPU tensors (no CPU fallback)")
        _ops()
        self.num_levels, self.radius = num_levels, radius
        self.out_dtype = fmap1.dtype
        self.dim = fmap1.shape[1]
        if map_dtype is None:
            map_dtype = torch.bfloat16 if fmap1.dtype == torch.bfloat16 else torch.float32
        if map_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("map_dtype must be torch.float32 or torch.bfloat16")
        self.map_dtype = map_dtype
        self.f1 = fmap1.to(map_dtype).permute(0, 2, 3, 1).contiguous()            # NHWC, what the kernel reads
        self.f2 = []
        f2 = fmap2.to(map_dtype)
        for _ in range(num_levels):
            self.f2.append(f2.permute(0, 2, 3, 1).contiguous())
            f2 = F.avg_pool2d(f2, 2, stride=2)

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        corr = self._windows(coords)
        return corr if self.out_dtype == torch.float32 else corr.to(self.out_dtype)

    def _windows(self, coords: torch.Tensor) -> torch.Tensor:
        """[B, L (2r+1)^2, H, W] in fp32 (the kernels' output type whatever the maps' type)."""
        from . import altcorr
        c = coords.float().permute(0, 2, 3, 1)
        B, H, W, _ = c.shape
        out = []
        for l in range(self.num_levels):
            ci = (c / 2 ** l).reshape(B, 1, H, W, 2).contiguous()
            if self.map_dtype == torch.bfloat16:       # the op itself returns fp32 (altcorr.forward would round it to the maps' dtype)
                corr = torch.ops.pfk.altcorr_forward(self.f1, self.f2[l], ci, self.radius)
            else:
                (corr,) = altcorr.forward(self.f1, self.f2[l], ci, self.radius)
            out.append(corr.squeeze(1))
        return torch.stack(out, dim=1).reshape(B, -1, H, W) / math.sqrt(self.dim)

    def lookup_pm(self, coords: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Pixel-major ``[B*h*w, C]`` form for the update engine (same contract as ``CorrBlock.lookup_pm``); ``out``: a
        ``[B*h*w, >= C]`` buffer of any float dtype to fill instead (pad columns are left alone)."""
        corr = self._windows(coords)
        B, C, H, W = corr.shape
        pm = corr.permute(0, 2, 3, 1).reshape(B * H * W, C)
        if out is not None:
            out[:, :C].copy_(pm)
            return out
        return pm.contiguous()


def get_corr_block(fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4,
                   alternate_corr: bool = False, pyramid: str = "avgpool", volume_dtype: Optional[torch.dtype] = None,
                   channels_last: bool = True):
    """Same signature as ptlflow/models/raft/corr.py:104-118; ``alternate_corr=True`` selects the on-demand block."""
    if alternate_corr:
        return AlternateCorrBlock(fmap1, fmap2, num_levels=num_levels, radius=radius)
    return CorrBlock(fmap1, fmap2, num_levels=num_levels, radius=radius, pyramid=pyramid, volume_dtype=volume_dtype,
                     channels_last=channels_last)
