"""Training path of the update block on libpfk kernels — SURVEY.md §8 f4 (first half).

The inference path fuses gates and concatenations into convolution epilogues and keeps its state in place, which autograd
cannot see.  For training the update block (ptlflow/models/raft/update.py:6-153) is re-composed from ONE differentiable
primitive, ``conv_pm`` — a "same" convolution over pixel-major ``[B*H*W, C]`` tensors whose forward, data gradient and weight
gradient all run on fp32-MFMA implicit-GEMM kernels — plus torch elementwise ops for the gates
(sigmoid, tanh, the GRU blend), which autograd differentiates by itself:

* forward   ``out = conv(srcs, W) + b`` (relu fused)                              -> ``pfk_conv2d_f32``
* dgrad     ``dX_s = conv(dY, flip(W_s)^T)``  — a convolution again                -> ``pfk_conv2d_f32`` (re-packed weight)
* wgrad     ``dW[:, c, ky, kx] = sum_p dY[p, :] * X[p + tap, c]``  -> ``pfk_conv_wgrad_f32``: the transposed product with the
  PIXEL as reduction index on the same matrix cores, read straight from the pixel-major tensors (tap padding by
  out-of-range buffer offsets, operands fetched from LDS "transposed" with conflict-free ``ds_read_b32``), pixel range cut
  into slices that are summed in a fixed order (deterministic), result in the forward weight's packed layout.
* dbias     column sums (torch).

Every MIOpen convolution of the update block's forward and backward is gone; the encoders, the correlation volume and
the lookup keep the reference's autograd in training (their backward kernels are later rows).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import load_native
from .packing import pack_conv_weight, round_up

EPI_LINEAR = 0


def _segments(srcs: Sequence[torch.Tensor], real: Sequence[int]) -> List[Tuple[int, int, int]]:
    segs, first = [], 0
    for s, n in zip(srcs, real):
        segs.append((first, n, s.shape[1]))
        first += n
    return segs


class _Geometry:
    __slots__ = ("B", "H", "W", "kh", "kw")

    def __init__(self, B, H, W, kh, kw):
        self.B, self.H, self.W, self.kh, self.kw = B, H, W, kh, kw


def _workspace(device) -> torch.Tensor:
    ws = _workspace.cache.get(device)
    if ws is None:
        ws = torch.zeros(torch.ops.pfk.conv_workspace_bytes(), device=device, dtype=torch.uint8)
        _workspace.cache[device] = ws
    return ws


_workspace.cache = {}


class ConvPacks:
    """Packed forms of one convolution weight (forward and per-source dgrad), built lazily and reused for as long as the
    parameter versions in ``key`` do not change — i.e. for all recurrent iterations of a training step."""
    __slots__ = ("key", "fwd", "dgrad")

    def __init__(self, key=None):
        self.key, self.fwd, self.dgrad = key, None, {}


def packs_for(cache: Optional[dict], name: str, params: Sequence[torch.Tensor]) -> ConvPacks:
    key = tuple((p.data_ptr(), p._version) for p in params)
    if cache is None:
        return ConvPacks(key)
    entry = cache.get(name)
    if entry is None or entry.key != key:
        entry = cache[name] = ConvPacks(key)
    return entry


class _ConvPM(torch.autograd.Function):
    """out[M, cout] = relu?(conv(cat(srcs), weight) + bias) on pixel-major tensors; srcs[i] is [M, C_i] with C_i % 4 == 0
    (``real[i]`` <= C_i real channels, the rest zero padding)."""

    @staticmethod
    def forward(ctx, weight, bias, g: _Geometry, relu: bool, real: Tuple[int, ...], packs: ConvPacks, *srcs):
        ops = torch.ops.pfk
        srcs = [s.contiguous() if s.stride(1) != 1 else s for s in srcs]
        cout = weight.shape[0]
        M = g.B * g.H * g.W
        if packs.fwd is None:
            packs.fwd = pack_conv_weight(weight.detach().float(), _segments(srcs, real))
        out = torch.empty(M, cout, device=srcs[0].device, dtype=torch.float32)
        ops.conv2d(list(srcs), g.B, g.H, g.W, g.kh, g.kw, packs.fwd, None if bias is None else bias.detach().float().contiguous(),
                   cout, EPI_LINEAR, relu, 1.0, out, None, None, None, _workspace(out.device))
        ctx.g, ctx.relu, ctx.real, ctx.has_bias, ctx.packs = g, relu, real, bias is not None, packs
        ctx.save_for_backward(weight, out if relu else None, *srcs)
        return out

    @staticmethod
    def backward(ctx, dY):
        ops = torch.ops.pfk
        weight, out, *srcs = ctx.saved_tensors
        g, real, packs = ctx.g, ctx.real, ctx.packs
        cout = weight.shape[0]
        M = g.B * g.H * g.W
        dY = dY.float().contiguous()
        if ctx.relu:
            dY = dY * (out > 0)
        ws = _workspace(dY.device)
        need = ctx.needs_input_grad      # (weight, bias, g, relu, real, packs, *srcs)
        dW = db = None
        dsrcs: List[Optional[torch.Tensor]] = [None] * len(srcs)
        # channels of dY padded to a multiple of 4 for its role as a convolution source (flow head: cout = 2)
        dY_src = dY if cout % 4 == 0 else F.pad(dY, (0, round_up(cout, 4) - cout))
        segs = _segments(srcs, real)
        # ---- dgrad: dX_s = conv(dY, W_s^T flipped)
        for i, (first, n, n_buf) in enumerate(segs):
            if not need[6 + i]:
                continue
            if i not in packs.dgrad:
                wt = weight.detach().float()[:, first:first + n].permute(1, 0, 2, 3).flip(2, 3)   # [n, cout, kh, kw]
                if n_buf > n:
                    wt = F.pad(wt, (0, 0, 0, 0, 0, 0, 0, n_buf - n))                              # padded channels: zero gradient
                packs.dgrad[i] = pack_conv_weight(wt.contiguous(), [(0, cout, dY_src.shape[1])])
            dx = torch.empty(M, n_buf, device=dY.device, dtype=torch.float32)
            ops.conv2d([dY_src], g.B, g.H, g.W, g.kh, g.kw, packs.dgrad[i], None, n_buf, EPI_LINEAR, False, 1.0, dx, None, None, None, ws)
            dsrcs[i] = dx
        # ---- wgrad: one launch (+ a deterministic slice reduction) straight from the pixel-major tensors, written in the packed
        # [cout, ktot] layout of the forward weight and un-packed here (the inverse of pack_conv_weight)
        if need[0]:
            taps = g.kh * g.kw
            ktot = sum(taps * round_up(n_buf, 32) for _, _, n_buf in segs)
            packed = torch.empty(dY_src.shape[1], ktot, device=dY.device, dtype=torch.float32)
            ops.conv_wgrad(list(srcs), dY_src, g.B, g.H, g.W, g.kh, g.kw, packed)
            dW = torch.empty(weight.shape, device=dY.device, dtype=torch.float32)
            k0 = 0
            for first, n, n_buf in segs:
                cpad = round_up(n_buf, 32)
                blk = packed[:cout, k0:k0 + taps * cpad].view(cout, g.kh, g.kw, cpad)[..., :n]
                dW[:, first:first + n] = blk.permute(0, 3, 1, 2)
                k0 += taps * cpad
            dW = dW.to(weight.dtype)
        if ctx.has_bias and need[1]:
            db = dY.sum(0)
        return (dW, db, None, None, None, None, *dsrcs)


def conv_pm(srcs: Sequence[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor], B: int, H: int, W: int,
            relu: bool = False, real: Optional[Sequence[int]] = None, packs: Optional[ConvPacks] = None) -> torch.Tensor:
    """Differentiable "same" convolution (stride 1, odd kernel) of the channel-concatenation of ``srcs`` (pixel-major
    ``[B*H*W, C_i]``, ``C_i % 4 == 0``) with a PyTorch-layout weight ``[cout, sum(real), kh, kw]``.  ``packs``: a
    ``ConvPacks`` from ``packs_for`` to reuse the packed weights across calls."""
    load_native()
    real = tuple(int(s.shape[1]) for s in srcs) if real is None else tuple(real)
    assert sum(real) == weight.shape[1], (real, tuple(weight.shape))
    g = _Geometry(B, H, W, weight.shape[2], weight.shape[3])
    return _ConvPM.apply(weight, bias, g, relu, real, packs if packs is not None else ConvPacks(), *srcs)


# --------------------------------------------------------------------------------------------------------------------
# The update block as a differentiable composition (raft/update.py:94-153)
# --------------------------------------------------------------------------------------------------------------------
def _pm(x: torch.Tensor) -> torch.Tensor:
    B, C, H, W = x.shape
    return x.float().permute(0, 2, 3, 1).reshape(B * H * W, C)


def _nchw(x: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    return x.view(B, H, W, x.shape[1]).permute(0, 3, 1, 2)


def _pad4(x: torch.Tensor) -> torch.Tensor:
    c = x.shape[1]
    return x if c % 4 == 0 else F.pad(x, (0, round_up(c, 4) - c))


def update_block_train(P: Dict[str, torch.Tensor], spec, net, inp, corr, flow, cache: Optional[dict] = None):
    """BasicUpdateBlock.forward / SmallUpdateBlock.forward (update.py:144-153 / :122-128) with every convolution on
    ``conv_pm``.  ``P``: the block's named parameters; ``cache``: a dict that keeps the packed weights between the recurrent
    calls of one training step.  NCHW in, NCHW out: ``(net, mask | None, delta_flow)``."""
    B, _, H, W = net.shape
    h, i, c, f = _pm(net), _pm(inp), _pm(corr), _pm(flow)

    def conv(srcs, name, relu=False, real=None):
        w = P[name + ".weight"]
        return conv_pm(srcs, w, P.get(name + ".bias"), B, H, W, relu, real, packs_for(cache, name, [w]))

    # motion encoder (update.py:104-112 / :85-91)
    cor = conv([_pad4(c)], "encoder.convc1", True, [c.shape[1]])
    if spec.c2:
        cor = conv([cor], "encoder.convc2", True)
    flo = conv([_pad4(f)], "encoder.convf1", True, [2])
    flo = conv([flo], "encoder.convf2", True)
    out = conv([cor, flo], "encoder.conv", True)
    motion = torch.cat([out, f], 1)                      # enc_out + 2
    x = _pad4(torch.cat([i, motion], 1))
    x_real = spec.context + spec.motion_channels
    # GRU passes (update.py:58-73 / :24-32): z and r share one convolution (weights concatenated on the output axis)
    for _kh, _kw, sfx in spec.gru_passes:
        wzr = torch.cat([P[f"gru.convz{sfx}.weight"], P[f"gru.convr{sfx}.weight"]], 0)
        bzr = torch.cat([P[f"gru.convz{sfx}.bias"], P[f"gru.convr{sfx}.bias"]], 0)
        zr = torch.sigmoid(conv_pm([h, x], wzr, bzr, B, H, W, False, [spec.hidden, x_real],
                                   packs_for(cache, "zr" + sfx, [P[f"gru.convz{sfx}.weight"], P[f"gru.convr{sfx}.weight"]])))
        z, r = zr[:, : spec.hidden], zr[:, spec.hidden:]
        q = torch.tanh(conv_pm([(r * h).contiguous(), x], P[f"gru.convq{sfx}.weight"], P[f"gru.convq{sfx}.bias"], B, H, W, False,
                               [spec.hidden, x_real], packs_for(cache, "q" + sfx, [P[f"gru.convq{sfx}.weight"]])))
        h = (1 - z) * h + z * q
    # heads (update.py:6-14, 138-142, 152)
    delta = conv([conv([h], "flow_head.conv1", True)], "flow_head.conv2")
    mask = None
    if spec.has_mask:
        mask = 0.25 * conv([conv([h], "mask.0", True)], "mask.2")
        mask = _nchw(mask, B, H, W)
    return _nchw(h, B, H, W), mask, _nchw(delta, B, H, W)
