"""TEST INFRASTRUCTURE (moved out of the product package in round 6: the staged reference superseded it everywhere else).
A torch-only RAFT laid out like ptlflow's — the *caller side* of seams B1 / B3 / B4 for machines without ptlflow.

`ptlflow_amd.RAFT` (raft.py) is the fast mirror: its own loop, pixel-major state, fused coordinate update.  What a ptlflow
user gets after `patch.accelerate(model)` is something else: the reference's OWN loop (ptlflow/models/raft/raft.py:125-194)
calling the wrapped seams — `get_corr_block(...)` looked up as a global of the model's module once per forward, then per
iteration `corr_fn(coords1)`, `flow = coords1 - coords0`, `update_block(net, inp, corr, flow)`, `coords1 + delta_flow` and
`upsample_flow` in torch ops (raft.py:112-123), NCHW tensors in between.  This module is that caller, written against torch
only, so the seam path can be checked where neither the reference tree nor its staged archive exists
(tests/test_seam_model.py, tests/test_gpu_seam_model.py):

* un-patched it is a plain PyTorch RAFT (convolutions on MIOpen, `matmul` / `avg_pool2d` / `grid_sample` correlation block):
  what ptlflow runs on an MI355X today;
* `ptlflow_amd.patch.accelerate(model)` treats it exactly like a ptlflow model: the module global `get_corr_block` below is
  rebound (B1), `model.update_block` wrapped by `PfkUpdateBlock` (B3), `model.fnet` / `model.cnet` by `PfkEncoder` (B4) —
  dispatch by (module, class) + state_dict shapes, registered at the bottom of this file through the same
  `register_update_block` / `register_encoder` calls INTEGRATION.md documents for further families.

state_dict keys and shapes equal the reference's (`update_block.encoder.convc1.weight`, `update_block.gru.convz1.weight`,
`update_block.mask.2.bias`, `fnet.layer2.0.downsample.0.weight`, ...), so `ptlflow_amd.RAFT.state_dict()` and ptlflow
checkpoints load with strict=True.
"""
from __future__ import annotations

import math
import sys
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from ptlflow_amd.raft import Encoder


# ----------------------------------------------------------------------------- seam B1: the torch correlation block
class TorchCorrBlock:
    """All-pairs volume, average-pooled pyramid and windowed bilinear lookup in torch ops — the contract of
    raft/corr.py:12-64: built once per forward, called with `coords [B,2,h,w]` (x, y), returns `[B, L*(2r+1)^2, h, w]` with
    channel `l*(2r+1)^2 + i*(2r+1) + j` for the sample at (x/2^l + i - r, y/2^l + j - r)."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4):
        B, D, h, w = fmap1.shape
        self.radius, self.shape = radius, (B, h, w)
        vol = torch.matmul(fmap1.flatten(2).transpose(1, 2), fmap2.flatten(2)) / math.sqrt(D)
        level = vol.reshape(B * h * w, 1, h, w)
        self.levels: List[torch.Tensor] = [level]
        for _ in range(num_levels - 1):
            level = F.avg_pool2d(level, 2, stride=2)
            self.levels.append(level)

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        B, h, w = self.shape
        r = self.radius
        n = 2 * r + 1
        centre = coords.permute(0, 2, 3, 1).reshape(B * h * w, 1, 1, 2)
        off = torch.linspace(-r, r, n, device=coords.device, dtype=coords.dtype)
        # window[i, j] = (off[i], off[j]) added to (x, y): the first window index moves x (the reference's layout)
        window = torch.stack(torch.meshgrid(off, off, indexing="ij"), dim=-1).view(1, n, n, 2)
        out = []
        for l, lvl in enumerate(self.levels):
            hl, wl = lvl.shape[-2:]
            pts = centre / 2 ** l + window
            gx = 2 * pts[..., 0] / (wl - 1) - 1
            gy = 2 * pts[..., 1] / (hl - 1) - 1
            s = F.grid_sample(lvl, torch.stack([gx, gy], dim=-1), align_corners=True)
            out.append(s.view(B, h, w, n * n))
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().to(coords.dtype)


def get_corr_block(fmap1, fmap2, num_levels: int = 4, radius: int = 4, alternate_corr: bool = False):
    """The module global `patch.accelerate` rebinds (the reference binds its own into every model module: raft.py:10)."""
    if alternate_corr:
        raise NotImplementedError("the torch stand-in has no on-demand block; use ptlflow_amd.get_corr_block(alternate_corr=True)")
    return TorchCorrBlock(fmap1, fmap2, num_levels=num_levels, radius=radius)


# ----------------------------------------------------------------------------- seam B3: torch update blocks
def _conv(cin: int, cout: int, k, pad=None) -> nn.Conv2d:
    k = (k, k) if isinstance(k, int) else k
    return nn.Conv2d(cin, cout, k, padding=(k[0] // 2, k[1] // 2) if pad is None else pad)


class _MotionEncoder(nn.Module):
    """raft/update.py:76-112: correlation branch (convc1 [, convc2]) and flow branch (convf1, convf2) -> conv, flow appended."""

    def __init__(self, corr_ch: int, c1: int, c2: int, f1: int, f2: int, out: int):
        super().__init__()
        self.convc1 = _conv(corr_ch, c1, 1)
        if c2:
            self.convc2 = _conv(c1, c2, 3)
        self.convf1 = _conv(2, f1, 7)
        self.convf2 = _conv(f1, f2, 3)
        self.conv = _conv((c2 or c1) + f2, out, 3)

    def forward(self, flow, corr):
        c = F.relu(self.convc1(corr))
        if hasattr(self, "convc2"):
            c = F.relu(self.convc2(c))
        f = F.relu(self.convf2(F.relu(self.convf1(flow))))
        return torch.cat([F.relu(self.conv(torch.cat([c, f], 1))), flow], 1)


class _GRU(nn.Module):
    """ConvGRU (one 3x3 pass, raft/update.py:17-32) or SepConvGRU (1x5 then 5x1, :35-73); gate names as the reference's."""

    def __init__(self, hidden: int, xin: int, passes):
        super().__init__()
        self.passes = passes
        for kh, kw, sfx in passes:
            for g in "zrq":
                setattr(self, f"conv{g}{sfx}", _conv(hidden + xin, hidden, (kh, kw)))

    def forward(self, h, x):
        for _, _, sfx in self.passes:
            hx = torch.cat([h, x], 1)
            z = torch.sigmoid(getattr(self, "convz" + sfx)(hx))
            r = torch.sigmoid(getattr(self, "convr" + sfx)(hx))
            q = torch.tanh(getattr(self, "convq" + sfx)(torch.cat([r * h, x], 1)))
            h = (1 - z) * h + z * q
        return h


class _FlowHead(nn.Module):
    def __init__(self, cin: int, hidden: int):
        super().__init__()
        self.conv1, self.conv2 = _conv(cin, hidden, 3), _conv(hidden, 2, 3)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(x)))


class BasicUpdateBlock(nn.Module):
    """forward(net, inp, corr, flow) -> (net, mask, delta_flow), raft/update.py:131-153."""

    def __init__(self, corr_channels: int = 324, hidden: int = 128):
        super().__init__()
        self.encoder = _MotionEncoder(corr_channels, 256, 192, 128, 64, 126)
        self.gru = _GRU(hidden, 128 + hidden, ((1, 5, "1"), (5, 1, "2")))
        self.flow_head = _FlowHead(hidden, 256)
        self.mask = nn.Sequential(_conv(hidden, 256, 3), nn.ReLU(inplace=True), _conv(256, 64 * 9, 1))

    def forward(self, net, inp, corr, flow):
        net = self.gru(net, torch.cat([inp, self.encoder(flow, corr)], 1))
        return net, 0.25 * self.mask(net), self.flow_head(net)


class SmallUpdateBlock(nn.Module):
    """raft/update.py:115-128: no mask head (the caller upsamples bilinearly)."""

    def __init__(self, corr_channels: int = 196, hidden: int = 96):
        super().__init__()
        self.encoder = _MotionEncoder(corr_channels, 96, 0, 64, 32, 80)
        self.gru = _GRU(hidden, 82 + 64, ((3, 3, ""),))
        self.flow_head = _FlowHead(hidden, 128)

    def forward(self, net, inp, corr, flow):
        net = self.gru(net, torch.cat([inp, self.encoder(flow, corr)], 1))
        return net, None, self.flow_head(net)


# ----------------------------------------------------------------------------- seam B4: encoders (classes under THIS module)
class BasicEncoder(Encoder):
    def __init__(self, output_dim: int, norm_fn: str):
        super().__init__(output_dim, norm_fn, False)


class SmallEncoder(Encoder):
    def __init__(self, output_dim: int, norm_fn: str):
        super().__init__(output_dim, norm_fn, True)


# ----------------------------------------------------------------------------- the caller
def convex_upsample_torch(flow: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """8x convex upsampling in torch ops (what raft.py:112-123 computes): softmax over the 9 neighbours, weighted sum of the
    3x3 neighbourhood of 8*flow, sub-pixel shuffle."""
    B, _, h, w = flow.shape
    wgt = torch.softmax(mask.view(B, 1, 9, 8, 8, h, w), dim=2)      # (a view also of PfkUpdateBlock's channels-last mask: only dim 1 is split)
    nb = F.unfold(8 * flow, 3, padding=1).view(B, 2, 9, 1, 1, h, w)
    up = (wgt * nb).sum(2)                                   # [B, 2, 8, 8, h, w]
    return up.permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * h, 8 * w)


class SeamRAFT(nn.Module):
    """`forward({"images": [B,2,3,H,W] BGR in [0,1]}) -> {"flows": [B,1,2,H,W], "flow_small"}` through the three seams."""

    def __init__(self, small: bool = False, iters: int = 32, corr_levels: int = 4):
        super().__init__()
        self.small, self.iters, self.corr_levels = small, iters, corr_levels
        self.corr_radius = 3 if small else 4
        cc = corr_levels * (2 * self.corr_radius + 1) ** 2
        if small:
            self.hidden_dim, self.context_dim = 96, 64
            self.fnet, self.cnet = SmallEncoder(128, "instance"), SmallEncoder(160, "none")
            self.update_block = SmallUpdateBlock(cc, 96)
        else:
            self.hidden_dim, self.context_dim = 128, 128
            self.fnet, self.cnet = BasicEncoder(256, "instance"), BasicEncoder(256, "batch")
            self.update_block = BasicUpdateBlock(cc, 128)

    def upsample_flow(self, flow: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """A METHOD of the model, as in the reference (raft.py:112-123) — seam B5: `patch.accelerate` shadows it on the instance
        with the convex-upsampling kernel once a behaviour probe has shown the two agree."""
        return convex_upsample_torch(flow, mask)

    @staticmethod
    def _pad_amounts(H: int, W: int):
        ph, pw = (-H) % 8, (-W) % 8                          # two-sided replicate padding to a multiple of 8
        return (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)

    @torch.no_grad()
    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x = torch.flip(2.0 * (inputs["images"] - 0.5), dims=[2])                 # BGR -> RGB, [-1, 1]
        B, _, _, H, W = x.shape
        pads = self._pad_amounts(H, W)
        x = F.pad(x.flatten(0, 1), pads, mode="replicate").unflatten(0, (B, 2))
        image1, image2 = x[:, 0], x[:, 1]
        fmap1, fmap2 = self.fnet([image1, image2])
        corr_fn = sys.modules[type(self).__module__].get_corr_block(          # the module GLOBAL, resolved per forward
            fmap1=fmap1, fmap2=fmap2, radius=self.corr_radius, num_levels=self.corr_levels, alternate_corr=False)
        net, inp = torch.split(self.cnet(image1), [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        h, w = image1.shape[-2] // 8, image1.shape[-1] // 8
        ys, xs = torch.meshgrid(torch.arange(h, device=x.device, dtype=x.dtype), torch.arange(w, device=x.device, dtype=x.dtype),
                                indexing="ij")
        coords0 = torch.stack([xs, ys], 0).expand(B, 2, h, w).contiguous()
        coords1 = coords0.clone()
        flow_up = None
        for _ in range(self.iters):
            coords1 = coords1.detach()
            corr = corr_fn(coords1)
            flow = coords1 - coords0
            net, up_mask, delta_flow = self.update_block(net, inp, corr, flow)
            coords1 = coords1 + delta_flow
            if up_mask is None:
                flow_up = 8 * F.interpolate(coords1 - coords0, size=(8 * h, 8 * w), mode="bilinear", align_corners=True)
            else:
                flow_up = self.upsample_flow(coords1 - coords0, up_mask)
            flow_up = flow_up[..., pads[2]: 8 * h - pads[3], pads[0]: 8 * w - pads[1]]
        return {"flows": flow_up[:, None], "flow_small": coords1 - coords0}


def _register() -> None:
    """Declare this module's classes to `patch.accelerate` as the raft implementations they are (the shape check still applies)."""
    from ptlflow_amd import patch
    from ptlflow_amd.update import basic_spec, small_spec
    patch.register_update_block(__name__, "BasicUpdateBlock", lambda cc: patch._with_corr_channels(basic_spec(), cc))
    patch.register_update_block(__name__, "SmallUpdateBlock", lambda cc: patch._with_corr_channels(small_spec(), cc))
    patch.register_encoder(__name__, "BasicEncoder")
    patch.register_encoder(__name__, "SmallEncoder")
    patch.register_last_flow_only_forward(__name__, "SeamRAFT")      # its eval forward keeps the last prediction only, like raft.py:186-192


_register()
