"""Training path of the update block on libpfk kernels — SURVEY.md §8 f4 (first half).

The inference path fuses gates and concatenations into convolution epilogues and keeps its state in place, which autograd
cannot see.  For training the update block (ptlflow/models/raft/update.py:6-153) is re-composed from ONE differentiable
primitive, ``conv_pm`` — a "same" convolution over pixel-major ``[B*H*W, C]`` tensors whose forward, data gradient and weight
gradient all run on fp32-MFMA implicit-GEMM kernels — plus one fused autograd node per GRU pass (``_GruPass``: gate
arithmetic and its derivatives in four small kernels, ``pfk_gru_*``):

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
    """Input grid (B, H, W), kernel and stride of one convolution; the output grid is (Ho, Wo) = ((H-1)//stride + 1, ...)."""
    __slots__ = ("B", "H", "W", "kh", "kw", "stride")

    def __init__(self, B, H, W, kh, kw, stride=1):
        self.B, self.H, self.W, self.kh, self.kw, self.stride = B, H, W, kh, kw, stride

    @property
    def Ho(self):
        return (self.H - 1) // self.stride + 1

    @property
    def Wo(self):
        return (self.W - 1) // self.stride + 1


def _workspace(device) -> torch.Tensor:
    ws = _workspace.cache.get(device)
    if ws is None:
        ws = torch.zeros(torch.ops.pfk.conv_workspace_bytes(), device=device, dtype=torch.uint8)
        _workspace.cache[device] = ws
    return ws


_workspace.cache = {}


class ConvPacks:
    """Packed forms of one convolution weight (forward and per-source dgrad), built lazily and reused for as long as the
    parameter versions in ``key`` do not change — i.e. for all recurrent iterations of a training step.

    ``acc``: gradient accumulators of a step-scoped entry (`packs_for(..., accumulate=True)`): the weight-gradient launches of
    all recurrent uses add into ONE buffer per parameter (`pfk_conv_wgrad_unpacked_f32(accumulate=1)`) and a `_Flush` node
    hands the total to autograd once — instead of every use returning its own gradient and the engine adding them (eleven
    small additions per parameter and step at 12 iterations)."""
    __slots__ = ("key", "fwd", "dgrad", "accumulate", "acc", "alias", "extra")

    def __init__(self, key=None, accumulate=False):
        self.key, self.fwd, self.dgrad = key, None, {}
        self.accumulate = accumulate
        self.acc = None          # {position among the parameters given to _Flush: accumulator}, None before the first use
        self.alias = None        # the parameters as seen by the uses of this step (outputs of _Flush)
        self.extra = {}          # per-step derived tensors (the z|r weight concatenation)


def packs_for(cache: Optional[dict], name: str, params: Sequence[torch.Tensor], accumulate: bool = False) -> ConvPacks:
    """``accumulate=True`` only for a ``cache`` that lives for ONE forward / backward (the mirror's training forward creates a
    fresh dict per step): the accumulators belong to that step's graph."""
    key = tuple((p.data_ptr(), p._version) for p in params)
    if cache is None:
        return ConvPacks(key)
    entry = cache.get(name)
    if entry is None or entry.key != key:
        entry = cache[name] = ConvPacks(key, accumulate)
    return entry


class _Flush(torch.autograd.Function):
    """Identity on a group of parameters whose uses accumulate their gradients in ``packs.acc`` (and return None): this node
    sits between the parameters and ALL their uses, so the engine runs its backward after the last use — whatever the order,
    however many uses — and it hands the totals on.  (Uses that never ran simply did not contribute; nothing is counted.)"""

    @staticmethod
    def forward(ctx, packs, *params):
        ctx.set_materialize_grads(False)
        ctx.packs = packs
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *grads):
        packs = ctx.packs
        acc, packs.acc = packs.acc, None
        # break the cycle ConvPacks -> alias (this node's outputs) -> grad_fn -> ctx -> ConvPacks, and drop the step's
        # accumulators: a later backward through the same graph returns its gradients the ordinary way (g below)
        packs.alias = None
        packs.extra.pop("dwzr", None)
        packs.extra.pop("dbzr", None)
        out = []
        for i, g in enumerate(grads):          # accumulated total, plus whatever a use returned the ordinary way
            a = None if acc is None else acc.get(i)
            out.append(a if g is None else (g if a is None else a + g))
        return (None, *out)


def step_params(packs: ConvPacks, *params):
    """The parameters as the uses of this step see them: aliases behind one `_Flush` node (accumulating entries) or themselves."""
    if not packs.accumulate or not any(p is not None and p.requires_grad for p in params) or not torch.is_grad_enabled():
        return params
    if packs.alias is None:
        real = [p for p in params if p is not None]
        out = iter(_Flush.apply(packs, *real))
        packs.alias = tuple(None if p is None else next(out) for p in params)
    return packs.alias


class _ConvPM(torch.autograd.Function):
    """out[M, cout] = relu?(conv(cat(srcs), weight) + bias) on pixel-major tensors; srcs[i] is [M, C_i] with C_i % 4 == 0
    (``real[i]`` <= C_i real channels, the rest zero padding)."""

    @staticmethod
    def forward(ctx, weight, bias, g: _Geometry, relu: bool, real: Tuple[int, ...], packs: ConvPacks, *srcs):
        ops = torch.ops.pfk
        srcs = [s.contiguous() if s.stride(1) != 1 else s for s in srcs]
        cout = weight.shape[0]
        out = torch.empty(g.B * g.Ho * g.Wo, cout, device=srcs[0].device, dtype=torch.float32)
        ops.conv2d(list(srcs), g.B, g.H, g.W, g.kh, g.kw, _fwd_pack(packs, weight, _segments(srcs, real)), None if bias is None else bias.detach().float().contiguous(),
                   cout, EPI_LINEAR, relu, 1.0, out, None, None, None, _workspace(out.device), None, g.stride)
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
            dY = torch.ops.aten.threshold_backward(dY, out, 0.0)     # dY * (out > 0) in one kernel (what torch's relu backward runs)
        ws = _workspace(dY.device)
        need = ctx.needs_input_grad      # (weight, bias, g, relu, real, packs, *srcs)
        dW = db = None
        dsrcs: List[Optional[torch.Tensor]] = [None] * len(srcs)
        # channels of dY padded to a multiple of 4 for its role as a convolution source (flow head: cout = 2)
        dY_src = dY if cout % 4 == 0 else F.pad(dY, (0, round_up(cout, 4) - cout))
        segs = _segments(srcs, real)
        # ---- dgrad: dX_s = conv(dY, W_s^T flipped); for a strided convolution the gradient is first spread over the input grid
        # (zeros between the samples: the transposed convolution written as a stride-1 one)
        dY_in = dY_src
        if g.stride != 1 and any(need[6:]):
            dY_in = torch.zeros(g.B, g.H, g.W, dY_src.shape[1], device=dY.device, dtype=torch.float32)
            dY_in[:, ::g.stride, ::g.stride] = dY_src.view(g.B, g.Ho, g.Wo, -1)
            dY_in = dY_in.view(M, -1)
        for i, (first, n, n_buf) in enumerate(segs):
            if not need[6 + i]:
                continue
            dx = torch.empty(M, n_buf, device=dY.device, dtype=torch.float32)
            ops.conv2d([dY_in], g.B, g.H, g.W, g.kh, g.kw, _dgrad_pack(packs, i, weight, (first, n, n_buf), dY_src.shape[1]), None, n_buf,
                       EPI_LINEAR, False, 1.0, dx, None, None, None, ws)
            dsrcs[i] = dx
        # ---- wgrad: one launch + a deterministic slice reduction straight from the pixel-major tensors; the reduction writes
        # PyTorch's [cout, cin, kh, kw] / [cout] layouts directly (pfk_conv_wgrad_unpacked_f32), the bias gradient being one more
        # K chunk whose A operand is a column of ones
        want_b = ctx.has_bias and need[1]
        if need[0] and packs.accumulate and packs.alias is not None:
            # step-scoped entry: add into the parameter's accumulators; `_Flush` returns the totals once (weight, [bias])
            first = packs.acc is None
            if first:
                packs.acc = {0: torch.empty(weight.shape, device=dY.device, dtype=torch.float32)}
                if want_b:
                    packs.acc[1] = torch.empty(cout, device=dY.device, dtype=torch.float32)
            ops.conv_wgrad_unpacked(list(srcs), dY_src, g.B, g.H, g.W, g.kh, g.kw, packs.acc[0], packs.acc.get(1),
                                    [n for _, n, _ in segs], g.stride, not first)
        elif need[0]:
            dW = torch.empty(weight.shape, device=dY.device, dtype=torch.float32)
            db = torch.empty(cout, device=dY.device, dtype=torch.float32) if want_b else None
            ops.conv_wgrad_unpacked(list(srcs), dY_src, g.B, g.H, g.W, g.kh, g.kw, dW, db, [n for _, n, _ in segs], g.stride)
            dW = dW.to(weight.dtype)
        elif want_b:
            db = dY.sum(0)
        return (dW, db, None, None, None, None, *dsrcs)


def _fwd_pack(packs: ConvPacks, weight: torch.Tensor, segs) -> torch.Tensor:
    if packs.fwd is None:
        packs.fwd = pack_conv_weight(weight.detach().float(), segs)
    return packs.fwd


def _dgrad_pack(packs: ConvPacks, i: int, weight: torch.Tensor, seg, dy_channels: int) -> torch.Tensor:
    """Packed weight of the data-gradient convolution w.r.t. source i: W[:, seg] with in/out swapped and taps flipped."""
    if i not in packs.dgrad:
        first, n, n_buf = seg
        cout = weight.shape[0]
        wt = weight.detach().float()[:, first:first + n].permute(1, 0, 2, 3).flip(2, 3)        # [n, cout, kh, kw]
        if n_buf > n:
            wt = F.pad(wt, (0, 0, 0, 0, 0, 0, 0, n_buf - n))                                   # padded channels: zero gradient
        packs.dgrad[i] = pack_conv_weight(wt.contiguous(), [(0, cout, dy_channels)])
    return packs.dgrad[i]


def _unpack_wgrad(packed: torch.Tensor, weight_shape, segs, g: _Geometry) -> torch.Tensor:
    """packed [cout(+pad), ktot] (the forward weight's layout) -> PyTorch layout [cout, cin, kh, kw]."""
    cout = weight_shape[0]
    taps = g.kh * g.kw
    dW = torch.empty(weight_shape, device=packed.device, dtype=torch.float32)
    k0 = 0
    for first, n, n_buf in segs:
        cpad = round_up(n_buf, 32)
        blk = packed[:cout, k0:k0 + taps * cpad].view(cout, g.kh, g.kw, cpad)[..., :n]
        dW[:, first:first + n] = blk.permute(0, 3, 1, 2)
        k0 += taps * cpad
    return dW


class _GruPass(torch.autograd.Function):
    """One ConvGRU / SepConvGRU pass (raft/update.py:24-32, 58-73) as a single autograd node:
        z, r = sigmoid(conv([h, x], Wz|Wr));  q = tanh(conv([r*h, x], Wq));  h' = (1 - z) h + z q
    two convolutions + two fused gate kernels forward; backward = two gate-derivative kernels, four data-gradient
    convolutions (the second pair accumulates into the first pair's results through the epilogue's residual input) and two
    weight-gradient launches."""

    @staticmethod
    def forward(ctx, h, x, wz, wr, wq, bz, br, bq, g: _Geometry, x_real: int, pk_zr: ConvPacks, pk_q: ConvPacks,
                pzr: Optional[torch.Tensor] = None, pq: Optional[torch.Tensor] = None):
        """``pzr`` [M, 2C] / ``pq`` [M, C]: additive terms of the z|r and q pre-activations — the loop-invariant context part of
        the gate convolutions, computed once per step by the caller (`update_block_train_pm`); ``x`` / the weights then cover
        the remaining input channels only.  Their gradient is the pre-activation gradient itself."""
        ops = torch.ops.pfk
        h = h.float().contiguous()
        x = x.float().contiguous()
        M, C = h.shape
        dev = h.device
        ws = _workspace(dev)
        if "wzr" not in pk_zr.extra:      # once per (parameter version): the z and r convolutions run as one
            pk_zr.extra["wzr"] = torch.cat([wz.detach(), wr.detach()], 0).float()
            pk_zr.extra["bzr"] = torch.cat([bz.detach(), br.detach()], 0).float().contiguous()
        wzr, bzr = pk_zr.extra["wzr"], pk_zr.extra["bzr"]
        segs = [(0, C, C), (C, x_real, x.shape[1])]
        a_zr = torch.empty(M, 2 * C, device=dev, dtype=torch.float32)
        ops.conv2d([h, x], g.B, g.H, g.W, g.kh, g.kw, _fwd_pack(pk_zr, wzr, segs), bzr, 2 * C, EPI_LINEAR, False, 1.0, a_zr,
                   None, None, None, ws, None if pzr is None else pzr.detach().float().contiguous())
        z, r, rh = (torch.empty(M, C, device=dev, dtype=torch.float32) for _ in range(3))
        ops.gru_gates_zr(a_zr, h, z, r, rh)
        a_q = torch.empty(M, C, device=dev, dtype=torch.float32)
        ops.conv2d([rh, x], g.B, g.H, g.W, g.kh, g.kw, _fwd_pack(pk_q, wq, segs), bq.detach().float().contiguous(), C, EPI_LINEAR,
                   False, 1.0, a_q, None, None, None, ws, None if pq is None else pq.detach().float().contiguous())
        q, hn = torch.empty_like(a_q), torch.empty_like(a_q)
        ops.gru_gates_q(a_q, z, h, q, hn)
        ctx.g, ctx.x_real, ctx.pk_zr, ctx.pk_q, ctx.wzr = g, x_real, pk_zr, pk_q, wzr
        ctx.has_p = (pzr is not None, pq is not None)
        ctx.save_for_backward(h, x, z, r, q, rh, wq)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        ops = torch.ops.pfk
        h, x, z, r, q, rh, wq = ctx.saved_tensors
        g, x_real, pk_zr, pk_q, wzr = ctx.g, ctx.x_real, ctx.pk_zr, ctx.pk_q, ctx.wzr
        M, C = h.shape
        Cx = x.shape[1]
        dev = h.device
        ws = _workspace(dev)
        segs = [(0, C, C), (C, x_real, Cx)]
        dhn = dhn.float()
        if dhn.stride(1) != 1:
            dhn = dhn.contiguous()
        da_q = torch.empty(M, C, device=dev, dtype=torch.float32)
        da_zr = torch.empty(M, 2 * C, device=dev, dtype=torch.float32)
        dh = torch.empty(M, C, device=dev, dtype=torch.float32)
        ops.gru_backward_q(dhn, z, q, h, da_q, da_zr, dh)

        def dgrad(dy, packs, weight, i, n_out, residual=None):
            out = torch.empty(M, n_out, device=dev, dtype=torch.float32)
            ops.conv2d([dy], g.B, g.H, g.W, g.kh, g.kw, _dgrad_pack(packs, i, weight, segs[i], dy.shape[1]), None, n_out, EPI_LINEAR,
                       False, 1.0, out, None, None, None, ws, residual)
            return out

        d_rh = dgrad(da_q, pk_q, wq, 0, C)
        dx = dgrad(da_q, pk_q, wq, 1, Cx)
        ops.gru_backward_zr(d_rh, h, r, da_zr, dh)
        dh = dgrad(da_zr, pk_zr, wzr, 0, C, residual=dh)          # + dh through the epilogue
        dx = dgrad(da_zr, pk_zr, wzr, 1, Cx, residual=dx)
        reals = [n for _, n, _ in segs]
        if pk_zr.accumulate and pk_zr.alias is not None and pk_q.alias is not None:
            # step-scoped entries: every pass of every iteration adds into one buffer per parameter; `_Flush` returns the totals
            first = pk_q.acc is None
            if first:
                dwq, dbq = torch.empty(wq.shape, device=dev, dtype=torch.float32), torch.empty(C, device=dev, dtype=torch.float32)
                dwzr, dbzr = torch.empty(wzr.shape, device=dev, dtype=torch.float32), torch.empty(2 * C, device=dev, dtype=torch.float32)
                pk_q.acc = {0: dwq, 1: dbq}
                pk_zr.extra["dwzr"], pk_zr.extra["dbzr"] = dwzr, dbzr
                pk_zr.acc = {0: dwzr[:C], 1: dwzr[C:], 2: dbzr[:C], 3: dbzr[C:]}      # (wz, wr, bz, br) as given to _Flush
            ops.conv_wgrad_unpacked([rh, x], da_q, g.B, g.H, g.W, g.kh, g.kw, pk_q.acc[0], pk_q.acc[1], reals, 1, not first)
            ops.conv_wgrad_unpacked([h, x], da_zr, g.B, g.H, g.W, g.kh, g.kw, pk_zr.extra["dwzr"], pk_zr.extra["dbzr"], reals, 1, not first)
            return (dh, dx, None, None, None, None, None, None, None, None, None, None,
                    da_zr if ctx.has_p[0] else None, da_q if ctx.has_p[1] else None)
        dwq, dbq = torch.empty(wq.shape, device=dev, dtype=torch.float32), torch.empty(C, device=dev, dtype=torch.float32)
        ops.conv_wgrad_unpacked([rh, x], da_q, g.B, g.H, g.W, g.kh, g.kw, dwq, dbq, reals, 1)
        dwzr, dbzr = torch.empty(wzr.shape, device=dev, dtype=torch.float32), torch.empty(2 * C, device=dev, dtype=torch.float32)
        ops.conv_wgrad_unpacked([h, x], da_zr, g.B, g.H, g.W, g.kh, g.kw, dwzr, dbzr, reals, 1)
        return (dh, dx, dwzr[:C], dwzr[C:], dwq, dbzr[:C], dbzr[C:], dbq, None, None, None, None,
                da_zr if ctx.has_p[0] else None, da_q if ctx.has_p[1] else None)


def conv_pm(srcs: Sequence[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor], B: int, H: int, W: int,
            relu: bool = False, real: Optional[Sequence[int]] = None, packs: Optional[ConvPacks] = None,
            stride: int = 1) -> torch.Tensor:
    """Differentiable convolution (odd kernel, padding k//2, stride s: PyTorch's Conv2d(k, stride=s, padding=k//2)) of the
    channel-concatenation of ``srcs`` (pixel-major ``[B*H*W, C_i]``, ``C_i % 4 == 0``) with a PyTorch-layout weight
    ``[cout, sum(real), kh, kw]``; returns ``[B*Ho*Wo, cout]``.  ``packs``: a ``ConvPacks`` from ``packs_for`` to reuse the
    packed weights across calls."""
    load_native()
    real = tuple(int(s.shape[1]) for s in srcs) if real is None else tuple(real)
    assert sum(real) == weight.shape[1], (real, tuple(weight.shape))
    g = _Geometry(B, H, W, weight.shape[2], weight.shape[3], int(stride))
    packs = packs if packs is not None else ConvPacks()
    weight, bias = step_params(packs, weight, bias)
    return _ConvPM.apply(weight, bias, g, relu, real, packs, *srcs)


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


def _context_terms(P: Dict[str, torch.Tensor], spec, i: torch.Tensor, B: int, H: int, W: int, cache: dict, owner: torch.Tensor,
                   acc: bool):
    """The loop-invariant part of the GRU (raft/update.py:60-71 / :27-30): the gate convolutions run over cat([h, inp, motion])
    and `inp` — the context features — is the same tensor in every iteration, so conv(W[:, inp slice], inp) is computed ONCE per
    training step (forward, data and weight gradient alike: autograd sums the twelve pre-activation gradients and runs one
    backward convolution) and the per-iteration nodes work on W without that slice.  Returns, per GRU pass,
    (wz_rest, wr_rest, wq_rest, pzr, pq); cached in `cache` for as long as `owner` (the caller's context tensor) is the same
    object at the same version — a new forward brings a new one."""
    import weakref
    ent = cache.get("ctx")
    if ent is not None and ent[0]() is owner and ent[1] == owner._version:
        return ent[2]
    Ch, Ci = spec.hidden, spec.context
    out = {}
    for kh, kw, sfx in spec.gru_passes:
        ws = [P[f"gru.conv{k}{sfx}.weight"] for k in "zrq"]
        rest = [torch.cat([w[:, :Ch], w[:, Ch + Ci:]], 1).contiguous() for w in ws]       # autograd routes the gradients back
        wc = [w[:, Ch:Ch + Ci].contiguous() for w in ws]
        wzr_c = torch.cat([wc[0], wc[1]], 0)
        pzr = conv_pm([i], wzr_c, None, B, H, W, False, None, packs_for(cache, "zrc" + sfx, [wzr_c], acc))
        pq = conv_pm([i], wc[2], None, B, H, W, False, None, packs_for(cache, "qc" + sfx, [wc[2]], acc))
        out[sfx] = (rest[0], rest[1], rest[2], pzr, pq)
    cache["ctx"] = (weakref.ref(owner), owner._version, out)
    return out


def update_block_train_pm(P: Dict[str, torch.Tensor], spec, h, i, c, f, B: int, H: int, W: int, cache: Optional[dict] = None,
                          accumulate_wgrad: bool = False, attention: Optional[torch.Tensor] = None, hoist_context: bool = True,
                          context_owner: Optional[torch.Tensor] = None):
    """BasicUpdateBlock.forward / SmallUpdateBlock.forward (update.py:144-153 / :122-128) with every convolution on
    ``conv_pm``, on pixel-major tensors: ``h`` [M, Ch], ``i`` [M, Ci], ``c`` [M, corr channels], ``f`` [M, 2] ->
    ``(h', mask [M, 576] | None, delta [M, 2])``.  ``P``: the block's named parameters; ``cache``: a dict that keeps the packed
    weights between the recurrent calls of one training step.  ``accumulate_wgrad`` (only with a ``cache`` that is created per
    step): the recurrent calls add their weight gradients into one buffer per parameter (`ConvPacks.acc`, `_Flush`)."""
    acc = accumulate_wgrad and cache is not None

    def conv(srcs, name, relu=False, real=None):
        w = P[name + ".weight"]
        return conv_pm(srcs, w, P.get(name + ".bias"), B, H, W, relu, real, packs_for(cache, name, [w], acc))

    # motion encoder (update.py:104-112 / :85-91)
    cor = conv([_pad4(c)], "encoder.convc1", True, [c.shape[1]])
    if spec.c2:
        cor = conv([cor], "encoder.convc2", True)
    flo = conv([_pad4(f)], "encoder.convf1", True, [2])
    flo = conv([flo], "encoder.convf2", True)
    out = conv([cor, flo], "encoder.conv", True)
    # x = [inp | motion features (encoder out | flow) | zero pad to a multiple of 4] in ONE concatenation (update.py:112 + :146);
    # with the context term hoisted (`_context_terms`, needs a cache to live in) `inp` stays out of it
    # (step-scoped caches only — `accumulate_wgrad`, the mirror's training forward: a cache that outlives the step would hand the
    #  next forward derived tensors whose autograd graph the previous backward has already freed)
    hoist = hoist_context and acc
    ctx_terms = _context_terms(P, spec, i, B, H, W, cache, i if context_owner is None else context_owner, acc) if hoist else None
    x_real = spec.x_channels - (spec.context if hoist else 0)
    parts = [out, f] if hoist else [i, out, f]
    if spec.aggregate:
        mf = torch.cat([out, f], 1)
        v = conv([mf], "aggregator.to_v")                                        # 1x1, no bias
        N = H * W
        agg = torch.bmm(attention.reshape(B, N, N).float(), v.view(B, N, v.shape[1])).reshape(B * N, v.shape[1])
        parts = [mf, mf + P["aggregator.gamma"] * agg] if hoist else [i, mf, mf + P["aggregator.gamma"] * agg]
    if x_real % 4:
        zkey = ("zpad", i.shape[0], round_up(x_real, 4) - x_real, i.device)
        z = cache.get(zkey) if cache is not None else None
        if z is None:
            z = torch.zeros(i.shape[0], round_up(x_real, 4) - x_real, device=i.device, dtype=torch.float32)
            if cache is not None:
                cache[zkey] = z
        parts.append(z)
    x = torch.cat(parts, 1)
    # GRU passes (update.py:58-73 / :24-32): one fused autograd node per pass (z and r share a convolution)
    for kh, kw, sfx in spec.gru_passes:
        names = [f"gru.conv{k}{sfx}" for k in "zrq"]
        wz, wr, wq = (P[n + ".weight"] for n in names)
        bz, br, bq = (P[n + ".bias"] for n in names)
        pzr = pq = None
        if hoist:
            wz, wr, wq, pzr, pq = ctx_terms[sfx]      # the weights without their context slice; that slice's products, once per step
        pk_zr = packs_for(cache, "zr" + sfx, [wz, wr, bz, br], acc)      # the cached z|r concatenation includes the biases
        pk_q = packs_for(cache, "q" + sfx, [wq], acc)
        wz, wr, bz, br = step_params(pk_zr, wz, wr, bz, br)
        wq, bq = step_params(pk_q, wq, bq)
        h = _GruPass.apply(h, x, wz, wr, wq, bz, br, bq, _Geometry(B, H, W, kh, kw), x_real, pk_zr, pk_q, pzr, pq)
    # heads (update.py:6-14, 138-142, 152)
    delta = conv([conv([h], "flow_head.conv1", True)], "flow_head.conv2")
    mask = None
    if spec.has_mask:
        mask = 0.25 * conv([conv([h], "mask.0", True)], "mask.2")
    return h, mask, delta


def update_block_train(P: Dict[str, torch.Tensor], spec, net, inp, corr, flow, cache: Optional[dict] = None):
    """NCHW face of `update_block_train_pm` (what `PfkUpdateBlock` calls from the reference's loop):
    NCHW in, NCHW out: ``(net, mask | None, delta_flow)``."""
    B, _, H, W = net.shape
    h, mask, delta = update_block_train_pm(P, spec, _pm(net), _pm(inp), _pm(corr), _pm(flow), B, H, W, cache, context_owner=inp)
    return _nchw(h, B, H, W), (None if mask is None else _nchw(mask, B, H, W)), _nchw(delta, B, H, W)


# --------------------------------------------------------------------------------------------------------------------
# Convex upsampling (raft/raft.py:112-123) and the sequence loss (raft/raft.py:20-45) for the training step
# --------------------------------------------------------------------------------------------------------------------
class _ConvexUpsample(torch.autograd.Function):
    """flow [B,2,H,W] (NCHW), mask pixel-major [B*H*W, 576] (already x0.25) -> [B,2,8H,8W]; forward and both gradients on
    libpfk (`pfk_convex_upsample_f32`, `pfk_convex_upsample_bwd_f32`)."""

    @staticmethod
    def forward(ctx, flow, mask):
        flow = flow.float().contiguous()
        mask = mask.float()
        if mask.stride(1) != 1:
            mask = mask.contiguous()
        B, _, H, W = flow.shape
        out = torch.empty(B, 2, 8 * H, 8 * W, device=flow.device, dtype=torch.float32)
        torch.ops.pfk.convex_upsample(flow, mask, out)
        ctx.save_for_backward(flow, mask)
        return out

    @staticmethod
    def backward(ctx, g):
        flow, mask = ctx.saved_tensors
        gmask = torch.empty(mask.shape[0], 576, device=mask.device, dtype=torch.float32)
        gflow = torch.empty_like(flow)
        torch.ops.pfk.convex_upsample_bwd(flow, mask, g.float().contiguous(), gmask, gflow)
        return gflow, gmask


def convex_upsample(flow: torch.Tensor, mask_pm: torch.Tensor) -> torch.Tensor:
    load_native()
    return _ConvexUpsample.apply(flow, mask_pm)


def sequence_loss(flow_preds: Sequence[torch.Tensor], flow_gt: torch.Tensor, valid: torch.Tensor, gamma: float = 0.8,
                  max_flow: float = 400.0) -> torch.Tensor:
    """SequenceLoss.forward (raft/raft.py:31-45): gamma-weighted L1 over the prediction sequence; pixels that are invalid
    or whose ground-truth displacement is >= max_flow are excluded by zeroing (the mean still runs over all pixels, as in
    the reference).  ``flow_gt`` [B,2,H,W], ``valid`` [B,1,H,W]."""
    n = len(flow_preds)
    mag = torch.sum(flow_gt ** 2, dim=1, keepdim=True).sqrt()
    keep = (valid >= 0.5) & (mag < max_flow)
    loss = 0.0
    for k, pred in enumerate(flow_preds):
        loss = loss + gamma ** (n - k - 1) * (keep * (pred - flow_gt).abs()).mean()
    return loss
