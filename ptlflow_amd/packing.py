"""Weight re-layout for the implicit-GEMM kernels (one-off host glue, pure tensor reshapes).

A PyTorch conv weight ``[cout, cin, kh, kw]`` becomes the K-contiguous matrix ``[cout, ktot]`` that
``pfk_conv2d_f32`` reads as its B operand (include/pfk.h): for every input *segment* (a channel
range of the reference's ``torch.cat`` operand that lives in its own pixel-major buffer slice), for
every tap (ky major, kx minor), the segment's channels zero-padded to a multiple of 32.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch
import torch.nn.functional as F


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_conv_weight(weight: torch.Tensor, segments: Sequence[Tuple[int, int, int]], kpad: int = 32) -> torch.Tensor:
    """``segments`` = [(first_channel, n_channels, n_channels_in_buffer)], in source order.
    ``kpad`` = channels per K-step of the kernel that will read it (32: fp32 MFMA, 64: split-bf16 MFMA).

    ``n_channels_in_buffer`` >= ``n_channels`` is how many channels the kernel is told the source
    has (a multiple of 4; extra ones are zero padding in the buffer and get zero weights here)."""
    cout, cin, kh, kw = weight.shape
    parts = []
    for first, n, n_buf in segments:
        assert first + n <= cin and n_buf >= n and n_buf % 4 == 0
        w = weight[:, first:first + n].permute(0, 2, 3, 1)  # [cout, kh, kw, n]
        w = F.pad(w, (0, round_up(n_buf, kpad) - n))
        parts.append(w.reshape(cout, -1))
    return torch.cat(parts, dim=1).contiguous()


def split_bf16_planes(packed: torch.Tensor, nsplit: int) -> torch.Tensor:
    """fp32 ``[cout, ktot]`` -> bf16 ``[nsplit, cout, ktot]`` with plane 0 = bf16(w), plane 1 = bf16(w - plane 0), ...
    (round to nearest even; every residual is exact in fp32) — the weight operand of ``pfk_conv2d_bf16s``."""
    assert 1 <= nsplit <= 3 and packed.dtype == torch.float32
    planes, r = [], packed
    for _ in range(nsplit):
        p = r.to(torch.bfloat16)
        planes.append(p)
        r = r - p.to(torch.float32)
    return torch.stack(planes, 0).contiguous()


def pack_cin2_weight(weight: torch.Tensor) -> torch.Tensor:
    """[cout, 2, k, k] -> [k*k, 2, cout] for pfk_conv_cin2_f32."""
    cout, cin, kh, kw = weight.shape
    assert cin == 2 and kh == kw
    return weight.permute(2, 3, 1, 0).reshape(kh * kw, 2, cout).contiguous()


def pack_flow_head_weight(weight: torch.Tensor, cin_buf: int = 0) -> torch.Tensor:
    """[2, cin, 3, 3] -> [9, 2, cin_buf] for pfk_flow_delta_f32."""
    cout, cin, kh, kw = weight.shape
    assert cout == 2 and kh == 3 and kw == 3
    w = weight.permute(2, 3, 0, 1).reshape(9, 2, cin)
    if cin_buf > cin:
        w = F.pad(w, (0, cin_buf - cin))
    return w.contiguous()


def mask_upsample_perm(device=None) -> Tuple["torch.Tensor", "torch.Tensor"]:
    """Row order of `pfk_mask_upsample_f32`'s weight / bias (include/pfk.h): 640 rows = [quarter (4)][tile j (5)][32 columns c], row
    `q*160 + j*32 + c` = mask channel `k*64 + s` with tap `k = 2j + (c >> 4)` and sub-pixel `s = q*16 + (c & 15)`; the rows with
    `k == 9` (tile 4, c >= 16) are zero.  Returns `(index [640] into the 576 channels, valid [640] bool)`."""
    q = torch.arange(4, device=device).view(4, 1, 1)
    j = torch.arange(5, device=device).view(1, 5, 1)
    c = torch.arange(32, device=device).view(1, 1, 32)
    k = 2 * j + (c >> 4)
    s = q * 16 + (c & 15)
    valid = (k < 9).expand(4, 5, 32).reshape(-1)
    idx = (k.clamp(max=8) * 64 + s).expand(4, 5, 32).reshape(-1)
    return idx, valid


def permute_mask_head(weight_packed: "torch.Tensor", bias: "torch.Tensor") -> Tuple["torch.Tensor", "torch.Tensor"]:
    """[576, ktot] packed 1x1 weight and [576] bias of the mask head's second convolution -> the [640, ktot] / [640] operands of
    `pfk_mask_upsample_f32`."""
    idx, valid = mask_upsample_perm(weight_packed.device)
    w = weight_packed[idx] * valid[:, None].to(weight_packed.dtype)
    b = bias[idx] * valid.to(bias.dtype)
    return w.contiguous(), b.contiguous()
