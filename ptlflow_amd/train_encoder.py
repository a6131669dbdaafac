"""Training path of the encoders on libpfk kernels — the last piece of BASELINE config 5 (a full RAFT training step with no
MIOpen in it).

Reference: torch.autograd through ptlflow/models/raft/extractor.py:122-267 (`BasicEncoder` / `SmallEncoder` in
`model.train()`: InstanceNorm for fnet, BatchNorm with BATCH statistics for cnet, no norm for raft_small's cnet).

Everything is pixel-major ``[B*H*W, C]`` between kernels, as in the inference engine (ptlflow_amd/encoder.py):

* stem 7x7/2 from the NCHW image   forward `pfk_conv_stem_f32`; weight / bias gradient `pfk_conv_stem_wgrad_f32` (the image needs
  no gradient);
* 3x3 / 1x1 convolutions           `train.conv_pm` (forward, data gradient and weight gradient on the fp32-MFMA implicit-GEMM
  kernels).  The four stride-2 convolutions run strided in the forward kernel and in the weight-gradient kernel (which reads
  the input at (2 yo + dy, 2 xo + dx)); only their data gradient is the stride-1 convolution of the zero-upsampled gradient;
* InstanceNorm / BatchNorm(train)  statistics `pfk_instnorm_stats_f32` (a training-mode batch norm is the same reduction with the
  whole batch as one "image"), normalise `pfk_norm_apply_f32`, backward `pfk_norm_bwd_f32` (which also yields d gamma / d beta);
* relu, residual add, the affine of BatchNorm   elementwise torch ops.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import load_native
from .train import ConvPacks, conv_pm, packs_for

EPS = 1e-5
_DEBUG_F64_STATS = False
_DEBUG_BN = ""


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(16, nbytes), device=device, dtype=torch.uint8)


class _Stem(torch.autograd.Function):
    """Conv2d(3, C, 7, stride 2, padding 3) from the NCHW image -> pixel-major [B*Ho*Wo, C] (no activation)."""

    @staticmethod
    def forward(ctx, img, weight, bias):
        ops = torch.ops.pfk
        img = img.float().contiguous()
        B, _, H, W = img.shape
        cout = weight.shape[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        packed = weight.detach().float().permute(2, 3, 1, 0).reshape(49, 3, cout).contiguous()
        out = torch.empty(B * Ho * Wo, cout, device=img.device, dtype=torch.float32)
        ops.conv_stem(img, packed, bias.detach().float().contiguous(), out, False)
        ctx.save_for_backward(img)
        ctx.cout = cout
        return out

    @staticmethod
    def backward(ctx, dy):
        (img,) = ctx.saved_tensors
        cout = ctx.cout
        dy = dy.float()
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        dw = torch.empty(49, 3, cout, device=dy.device, dtype=torch.float32)
        db = torch.empty(cout, device=dy.device, dtype=torch.float32)
        torch.ops.pfk.conv_stem_wgrad(img, dy, dw, db)
        return None, dw.view(7, 7, 3, cout).permute(3, 2, 0, 1).contiguous(), db


class _Norm(torch.autograd.Function):
    """x [G*HW, C] -> x_hat = (x - mean) * rstd with statistics per (group g, channel) over HW rows, optional fused relu.
    Instance norm: G = images.  Batch norm in training mode: G = 1, HW = every pixel of the batch.  Also returns mean and
    the biased variance (for the running statistics), not differentiated."""

    @staticmethod
    def forward(ctx, x, G: int, HW: int, relu: bool):
        ops = torch.ops.pfk
        x = x.float()
        if x.stride(1) != 1:
            x = x.contiguous()
        C = x.shape[1]
        if _DEBUG_F64_STATS:      # diagnostic only (scripts/enc_grad_check.py): statistics in float64 by torch
            xd = x.double().view(G, HW, C)
            mean = xd.mean(1).reshape(-1).float()
            rstd = (1.0 / torch.sqrt(xd.var(1, unbiased=False) + EPS)).reshape(-1).float()
        else:
            mean = torch.empty(G * C, device=x.device, dtype=torch.float32)
            rstd = torch.empty(G * C, device=x.device, dtype=torch.float32)
            ops.instnorm_stats(x, G, HW, EPS, mean, rstd, _ws(ops.instnorm_workspace_bytes(G, C), x.device))
        out = torch.empty(x.shape[0], C, device=x.device, dtype=torch.float32)
        ops.norm_apply(x, mean, rstd, None, out, G, HW, relu, False)
        ctx.save_for_backward(x, mean, rstd)
        ctx.G, ctx.HW, ctx.relu = G, HW, relu
        ctx.mark_non_differentiable(mean, rstd)
        return out, mean, rstd

    @staticmethod
    def backward(ctx, dy, _dm, _dr):
        x, mean, rstd = ctx.saved_tensors
        dy = dy.float()
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        C = x.shape[1]
        dx = torch.empty_like(x)
        s1 = torch.empty(ctx.G * C, device=x.device, dtype=torch.float32)
        s2 = torch.empty(ctx.G * C, device=x.device, dtype=torch.float32)
        torch.ops.pfk.norm_bwd(x, dy, mean, rstd, dx, s1, s2, ctx.G, ctx.HW, ctx.relu)
        return dx, None, None, None


def _norm(kind: str, x: torch.Tensor, B: int, HW: int, bn: Optional[torch.nn.Module], relu: bool) -> torch.Tensor:
    """`relu?(norm(x))` for norm_fn in {instance, batch (training statistics), none}."""
    if kind == "none":
        return torch.relu(x) if relu else x
    if kind == "instance":
        return _Norm.apply(x, B, HW, relu)[0]
    # nn.BatchNorm2d in training mode: batch statistics, affine, running buffers updated with momentum (unbiased variance)
    if _DEBUG_BN == "miopen":      # diagnostics only (scripts/enc_grad_check.py)
        import torch.nn.functional as F
        C = x.shape[1]
        y = F.batch_norm(x.view(B, HW, C).permute(0, 2, 1).reshape(B, C, HW, 1), None, None, bn.weight, bn.bias, True, 0.1, EPS)
        y = y.reshape(B, C, HW).permute(0, 2, 1).reshape(B * HW, C)
        return torch.relu(y) if relu else y
    if _DEBUG_BN == "torch":
        m = x.mean(0, keepdim=True)
        v = (x - m).square().mean(0, keepdim=True)
        y = (x - m) * torch.rsqrt(v + EPS) * bn.weight + bn.bias
        return torch.relu(y) if relu else y
    if not bn.training and bn.running_mean is not None:
        # a BatchNorm put in eval inside a training model (the reference's `freeze_bn`, raft.py:93-101): running statistics as
        # constants, no buffer update — an affine map of x, differentiable w.r.t. x, weight and bias through plain torch ops
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        y = (x - bn.running_mean) * scale + bn.bias
        return torch.relu(y) if relu else y
    xh, mean, rstd = _Norm.apply(x, 1, B * HW, False)
    y = xh * bn.weight + bn.bias
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            n = B * HW
            var = (1.0 / (rstd * rstd) - EPS) * (n / max(n - 1, 1))
            m = bn.momentum if bn.momentum is not None else 0.1
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(var, alpha=m)
            bn.num_batches_tracked.add_(1)
    return torch.relu(y) if relu else y


def encoder_train(enc: torch.nn.Module, img: torch.Tensor, cache: Optional[dict] = None) -> torch.Tensor:
    """`BasicEncoder.forward` / `SmallEncoder.forward` in training mode for the mirror's `Encoder` module (or any module
    with the reference's attribute names): ``img`` [B, 3, H, W] -> NCHW-shaped channels-last view [B, out_dim, H/8, W/8].
    ``cache``: packed-weight cache shared by the calls of one step."""
    load_native()
    kind = enc.norm_fn
    if kind not in ("instance", "batch", "none"):
        raise RuntimeError(f"encoder_train: norm_fn {kind!r} has no kernels")
    B, _, H, W = img.shape
    cache = {} if cache is None else cache

    def conv(x, mod, name, h, w, stride=1):
        wgt = mod.weight
        return conv_pm([x], wgt, mod.bias, B, h, w, False, None, packs_for(cache, name, [wgt]), stride)

    h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = _Stem.apply(img, enc.conv1.weight, enc.conv1.bias)
    x = _norm(kind, x, B, h * w, enc.norm1 if kind == "batch" else None, True)
    for li in (1, 2, 3):
        layer = getattr(enc, f"layer{li}")
        for bi in (0, 1):
            blk = layer[bi]
            name = f"layer{li}.{bi}"
            strided = blk.downsample is not None
            stride = 2 if strided else 1
            ho, wo = ((h - 1) // stride + 1, (w - 1) // stride + 1)
            bn = kind == "batch"
            if getattr(blk, "bottleneck", hasattr(blk, "conv3")):
                y = _norm(kind, conv(x, blk.conv1, name + ".conv1", h, w), B, h * w, blk.norm1 if bn else None, True)
                y = _norm(kind, conv(y, blk.conv2, name + ".conv2", h, w, stride), B, ho * wo, blk.norm2 if bn else None, True)
                y = _norm(kind, conv(y, blk.conv3, name + ".conv3", ho, wo), B, ho * wo, blk.norm3 if bn else None, True)
            else:
                y = _norm(kind, conv(x, blk.conv1, name + ".conv1", h, w, stride), B, ho * wo, blk.norm1 if bn else None, True)
                y = _norm(kind, conv(y, blk.conv2, name + ".conv2", ho, wo), B, ho * wo, blk.norm2 if bn else None, True)
            if strided:
                x = _norm(kind, conv(x, blk.downsample[0], name + ".ds", h, w, stride), B, ho * wo,
                          blk.downsample[1] if bn else None, False)
            x = torch.relu(x + y)
            h, w = ho, wo
    y = conv(x, enc.conv2, "conv2", h, w)
    return y.view(B, h, w, y.shape[1]).permute(0, 3, 1, 2)
