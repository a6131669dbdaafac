"""Deterministic synthetic weights and inputs.

There are no checkpoints offline (the reference's URLs, ptlflow/models/raft/raft.py:49-54, need a
network), so benchmarks, golden fixtures and parity tests use weights drawn from a seeded generator
— the same values wherever they are regenerated, independent of module construction order.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Sequence

import torch


def synth_tensor(name: str, shape: Sequence[int], gen: torch.Generator, fan_in: int = 0) -> torch.Tensor:
    """Same *statistics* as the reference's initialisation — encoders: kaiming-normal, fan_out, relu
    (raft/extractor.py:160-168); everything else: PyTorch's Conv2d default, U(+-1/sqrt(fan_in)) for
    weight and bias — which keeps the 32-iteration recurrence well conditioned (fp32-vs-fp64 EPE ~1e-5);
    BatchNorm running statistics are perturbed so that they are actually exercised."""
    shape = tuple(shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if name.endswith("running_var"):
        return 1.0 + 0.2 * torch.rand(shape, generator=gen)
    if name.endswith("running_mean"):
        return 0.05 * torch.randn(shape, generator=gen)
    if name.endswith(".gamma"):   # GMA's aggregate gate (init 0 in the reference; non-zero here so the path is exercised)
        return 0.5 + 0.1 * torch.randn(shape, generator=gen)
    if len(shape) == 4:
        if name.startswith(("fnet.", "cnet.")):
            fan_out = shape[0] * shape[2] * shape[3]
            return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_out)
        bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound
    if fan_in:  # conv bias
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound
    if name.endswith("weight"):  # norm scale
        return 1.0 + 0.1 * torch.randn(shape, generator=gen)
    return 0.05 * torch.randn(shape, generator=gen)  # norm shift


def synth_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 1234) -> Dict[str, torch.Tensor]:
    """One tensor per (sorted) name; every tensor has its own sub-seed so adding keys never shifts others."""
    out = {}
    for i, name in enumerate(sorted(shapes)):
        gen = torch.Generator().manual_seed(seed * 100003 + i)
        fan_in = 0
        if name.endswith(".bias"):
            wshape = shapes.get(name[:-5] + ".weight")
            if wshape is not None and len(wshape) == 4:
                fan_in = wshape[1] * wshape[2] * wshape[3]
        out[name] = synth_tensor(name, shapes[name], gen, fan_in)
    return out


def update_block_shapes(spec) -> Dict[str, tuple]:
    """Parameter shapes of BasicUpdateBlock / SmallUpdateBlock."""
    s = spec
    sh: Dict[str, tuple] = {}

    def conv(name, co, ci, kh, kw):
        sh[name + ".weight"] = (co, ci, kh, kw)
        sh[name + ".bias"] = (co,)

    conv("encoder.convc1", s.c1, s.corr_channels, 1, 1)
    if s.c2:
        conv("encoder.convc2", s.c2, s.c1, 3, 3)
    conv("encoder.convf1", s.f1, 2, 7, 7)
    conv("encoder.convf2", s.f2, s.f1, 3, 3)
    conv("encoder.conv", s.enc_out, (s.c2 if s.c2 else s.c1) + s.f2, 3, 3)
    cin = s.hidden + s.x_channels
    if getattr(s, "aggregate", False) and not getattr(s, "external_aggregate", False):
        sh["aggregator.to_v.weight"] = (s.motion_channels, s.motion_channels, 1, 1)
        sh["aggregator.gamma"] = (1,)
    for kh, kw, sfx in s.gru_passes:
        for k in "zrq":
            conv(f"gru.conv{k}{sfx}", s.hidden, cin, kh, kw)
    conv("flow_head.conv1", s.fh_hidden, s.hidden, 3, 3)
    conv("flow_head.conv2", 2, s.fh_hidden, 3, 3)
    if s.has_mask:
        conv("mask.0", 256, s.hidden, 3, 3)
        conv("mask.2", getattr(s, "mask_channels", 576), 256, 1, 1)
    return sh


def synth_update_block_params(spec, seed: int = 1234) -> Dict[str, torch.Tensor]:
    return synth_state_dict(update_block_shapes(spec), seed)


def rand_pair(B: int, H: int, W: int, seed: int = 1234) -> torch.Tensor:
    """iid uniform [0,1) frame pair [B,2,3,H,W] — what the reference's model_benchmark.py feeds
    (model_benchmark.py:445-453)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 2, 3, H, W, generator=g)


def smooth_pair(B: int, H: int, W: int, seed: int = 1234, shift=(5, -4)) -> torch.Tensor:
    """A smooth random texture and a copy shifted by `shift` px: [B,2,3,H,W] in [0,1] — flow-like input
    for EPE checks (distribution ii).  Same construction as oracle.raft_oracle.smooth_pair."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(seed)
    m = 8
    base = torch.rand(B, 3, H // 8 + 4 + m, W // 8 + 4 + m, generator=g)
    big = F.interpolate(base, scale_factor=8, mode="bicubic", align_corners=False).clamp(0, 1)
    oy, ox = 4 * 8, 4 * 8
    im1 = big[..., oy: oy + H, ox: ox + W]
    im2 = big[..., oy - shift[1]: oy - shift[1] + H, ox - shift[0]: ox - shift[0] + W]
    return torch.stack([im1, im2], dim=1).contiguous()
