"""Seam B2 — the reference's only native plug-in point on this path: a Python module literally importable as
``alt_cuda_corr`` with ``forward(fmap1, fmap2, coords, radius) -> [corr]``
(ptlflow/utils/external/alt_cuda_corr/correlation.cpp:23-54; imported in a bare try/except by every family's
corr.py, e.g. ptlflow/models/raft/corr.py:5-8, and called from ``AlternateCorrBlock.__call__``, :76-101).

``install()`` registers this module under that name in ``sys.modules`` (do it before ``import ptlflow``), after
which ccmr / ccmr_p / ms_raft_p — which default to ``alternate_corr=True`` — pick the gfx950 kernel up with zero
patching.  The repo root also carries a one-line ``alt_cuda_corr.py`` so that having the repo on ``sys.path``
is enough.

``forward`` keeps the reference's contract exactly: fp32 contiguous GPU tensors, ``fmap1 [B,H1,W1,C]``,
``fmap2 [B,H2,W2,C]``, ``coords [B,N,H1,W1,2]`` (x, y), returns ``[corr]`` with ``corr [B,N,(2r+1)^2,H1,W1]``,
cell index ``iy + (2r+1)*ix`` (x-major), unscaled.  ``backward`` mirrors correlation.cpp:39-49 (the reference itself never
wires it into autograd: AlternateCorrBlock calls ``forward`` directly).
"""
from __future__ import annotations

import sys
from typing import List

import torch

from . import load_native


def forward(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, radius: int) -> List[torch.Tensor]:
    load_native()
    for name, t in (("fmap1", fmap1), ("fmap2", fmap2), ("coords", coords)):
        if not t.is_cuda:                       # CHECK_CUDA, correlation.cpp:19
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():               # CHECK_CONTIGUOUS, correlation.cpp:20
            raise RuntimeError(f"{name} must be contiguous")
    if fmap1.dtype == torch.bfloat16 and fmap2.dtype == torch.bfloat16:
        # bf16 maps go to the kernel as they are (pfk_altcorr_forward_bf16: exact widening, fp32 products and accumulation): half
        # the gather traffic, no fp32 copies of the maps; the result takes the maps' dtype like the half path below
        out = torch.ops.pfk.altcorr_forward(fmap1, fmap2, coords.float(), int(radius))
        return [out.to(fmap1.dtype)]
    if fmap1.dtype != torch.float32:
        # the reference's extension is float-only (correlation_kernel.cu:275: corr_forward_kernel<float>); its callers up-cast
        # half inputs and cast the result back (raft/corr.py:90-96).  Accepting them here saves the caller that dance.
        out = torch.ops.pfk.altcorr_forward(fmap1.float(), fmap2.float(), coords.float(), int(radius))
        return [out.to(fmap1.dtype)]
    return [torch.ops.pfk.altcorr_forward(fmap1, fmap2, coords, int(radius))]


def backward(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, corr_grad: torch.Tensor,
             radius: int) -> List[torch.Tensor]:
    """`alt_cuda_corr.backward` (correlation.cpp:39-49): [fmap1_grad, fmap2_grad, coords_grad]; like the reference,
    coords_grad is all zeros (correlation_kernel.cu:307) and fmap2_grad is accumulated with fp32 atomics."""
    load_native()
    for name, t in (("fmap1", fmap1), ("fmap2", fmap2), ("coords", coords), ("corr_grad", corr_grad)):
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
    return list(torch.ops.pfk.altcorr_backward(fmap1, fmap2, coords, corr_grad, int(radius)))


def install() -> None:
    """Make ``import alt_cuda_corr`` resolve to this module."""
    sys.modules.setdefault("alt_cuda_corr", sys.modules[__name__])
