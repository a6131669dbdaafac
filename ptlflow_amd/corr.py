"""Seam B1 — drop-in for ``get_corr_block`` / ``CorrBlock`` (ptlflow/models/raft/corr.py:12-64,104-118;
identical copies in gma/corr.py, ccmr/corr.py, ms_raft_plus/corr.py; sea_raft/corr.py:71-117 differs
only in how the pyramid is built).

Same call contract as the reference: built once per forward from ``fmap1, fmap2 [B,D,h,w]``, then
called ``iters`` times with ``coords [B,2,h,w]`` (x, y in pixels) and returns
``[B, L*(2r+1)^2, h, w]`` with channel ``l*(2r+1)^2 + (dx+r)*(2r+1) + (dy+r)``.

MI355X layout: the volume is built by one fp32-MFMA GEMM launch (K1) straight into
``[B*N, h, w]`` maps, pooled by K2, and every lookup is ONE launch over all levels (K3) that
writes a pixel-major ``[B*N, C]`` buffer; the tensor handed back is a channels-last *view* of that
buffer (shape and values as the reference, strides NHWC) so the update block's first 1x1
convolution reads it without any transpose.
"""
from __future__ import annotations

import math
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


class CorrBlock:
    """All-pairs correlation pyramid + radius-r lookup on the GPU (inference path; ``coords`` is
    detached by every caller — raft.py:171 — so no coordinate gradient exists)."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4,
                 pyramid: str = "avgpool"):
        if not fmap1.is_cuda:
            raise RuntimeError("ptlflow_amd.CorrBlock needs GPU tensors (no CPU fallback)")
        if not 1 <= radius <= 4:
            raise RuntimeError("radius must be in 1..4")
        _ops()
        self.num_levels = num_levels
        self.radius = radius
        self.pyramid_mode = pyramid
        if pyramid not in ("avgpool", "bilinear_f2"):
            raise ValueError(f"unknown pyramid mode {pyramid!r}")
        n = 2 * radius + 1
        self.channels = num_levels * n * n
        self._out: Optional[torch.Tensor] = None
        self.corr_pyramid: List[torch.Tensor] = []
        self._shape = None
        self.update(fmap1, fmap2)

    def update(self, fmap1: torch.Tensor, fmap2: torch.Tensor) -> "CorrBlock":
        """(Re)build the pyramid for a new frame pair.  Feature maps of the same shape as last time are written into the
        SAME level tensors (same addresses), which is what lets the iteration loop live in a captured hipGraph."""
        ops = _ops()
        self.out_dtype = fmap1.dtype
        B, D, h, w = fmap1.shape
        self.B, self.h, self.w = B, h, w
        N = h * w
        f1 = to_pixel_major(fmap1)
        scale = 1.0 / math.sqrt(D)
        shape = (B, D, h, w, tuple(fmap2.shape), fmap1.device)
        fresh = shape != self._shape
        if fresh:
            self.corr_pyramid, self._out, self._shape = [], None, shape
        dev = f1.device
        if self.pyramid_mode == "avgpool":  # raft/corr.py:19-27
            f2 = to_pixel_major(fmap2)
            if fresh:
                hl, wl = h, w
                for _ in range(self.num_levels):
                    self.corr_pyramid.append(torch.empty(B * N, hl, wl, device=dev, dtype=torch.float32))
                    hl, wl = hl // 2, wl // 2
            ops.corr_volume(f1, f2, scale, self.corr_pyramid[0].view(B, N, N))
            for l in range(1, self.num_levels):
                ops.corr_pool2x2(self.corr_pyramid[l - 1], self.corr_pyramid[l])
        else:  # sea_raft/corr.py:77-84: one GEMM per level
            f2n = fmap2.float()
            for l in range(self.num_levels):
                if l > 0:
                    f2n = F.interpolate(f2n, scale_factor=0.5, mode="bilinear", align_corners=False)
                h2, w2 = f2n.shape[-2:]
                f2 = to_pixel_major(f2n)
                if fresh:
                    self.corr_pyramid.append(torch.empty(B * N, h2, w2, device=dev, dtype=torch.float32))
                ops.corr_volume(f1, f2, scale, self.corr_pyramid[l].view(B, N, h2 * w2))
        return self

    def lookup_pm(self, coords: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Lookup into a pixel-major ``[B*h*w, C]`` buffer (allocated once and reused unless given)."""
        if out is None:
            if self._out is None:
                self._out = torch.empty(self.B * self.h * self.w, self.channels, device=coords.device,
                                        dtype=torch.float32)
            out = self._out
        c = coords
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        _ops().corr_lookup(self.corr_pyramid, c, self.radius, out)
        return out

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        out = self.lookup_pm(coords)
        res = out.view(self.B, self.h, self.w, self.channels).permute(0, 3, 1, 2)
        if self.out_dtype != torch.float32:
            res = res.to(self.out_dtype)
        return res


class AlternateCorrBlock:
    """On-demand correlation (ptlflow/models/raft/corr.py:67-101): no N x N volume is materialised; every call computes,
    per pyramid level of ``fmap2`` (avg-pooled), the (2r+1)^2 window around ``coords / 2^l`` with the gfx950 kernel behind
    the ``alt_cuda_corr`` ABI (``pfk_altcorr_forward_f32``).  Same channel layout and ``/ sqrt(dim)`` as the reference."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4):
        if not fmap1.is_cuda:
            raise RuntimeError("ptlflow_amd.AlternateCorrBlock needs GPU tensors (no CPU fallback)")
        _ops()
        self.num_levels, self.radius = num_levels, radius
        self.out_dtype = fmap1.dtype
        self.dim = fmap1.shape[1]
        self.f1 = fmap1.float().permute(0, 2, 3, 1).contiguous()                 # NHWC, what the kernel reads
        self.f2 = []
        f2 = fmap2.float()
        for _ in range(num_levels):
            self.f2.append(f2.permute(0, 2, 3, 1).contiguous())
            f2 = F.avg_pool2d(f2, 2, stride=2)

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        from . import altcorr
        c = coords.float().permute(0, 2, 3, 1)
        B, H, W, _ = c.shape
        out = []
        for l in range(self.num_levels):
            ci = (c / 2 ** l).reshape(B, 1, H, W, 2).contiguous()
            (corr,) = altcorr.forward(self.f1, self.f2[l], ci, self.radius)
            out.append(corr.squeeze(1))
        corr = torch.stack(out, dim=1).reshape(B, -1, H, W) / math.sqrt(self.dim)
        return corr if self.out_dtype == torch.float32 else corr.to(self.out_dtype)

    def lookup_pm(self, coords: torch.Tensor) -> torch.Tensor:
        """Pixel-major ``[B*h*w, C]`` form for the update engine (same contract as ``CorrBlock.lookup_pm``)."""
        corr = self(coords).float()
        B, C, H, W = corr.shape
        return corr.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def get_corr_block(fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4,
                   alternate_corr: bool = False, pyramid: str = "avgpool"):
    """Same signature as ptlflow/models/raft/corr.py:104-118; ``alternate_corr=True`` selects the on-demand block."""
    if alternate_corr:
        return AlternateCorrBlock(fmap1, fmap2, num_levels=num_levels, radius=radius)
    return CorrBlock(fmap1, fmap2, num_levels=num_levels, radius=radius, pyramid=pyramid)
