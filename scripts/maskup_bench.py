#!/usr/bin/env python3
"""Fused mask conv2 + softmax + convex upsampling (pfk_mask_upsample_f32) against the two launches it replaces (GPU box)."""
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
from ptlflow_amd.packing import pack_conv_weight, permute_mask_head
ptlflow_amd.load_native()
ops = torch.ops.pfk
dev = torch.device("cuda")
torch.manual_seed(0)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for B in (1, 2, 4, 8, 16):
    H, W, cin = 55, 128, 256
    M = B * H * W
    fm = torch.randn(M, 512, device=dev)
    hx = torch.randn(M, 388, device=dev)
    wt = torch.randn(576, cin, 1, 1, device=dev) / math.sqrt(cin)
    bias = torch.randn(576, device=dev) * 0.1
    packed = pack_conv_weight(wt, [(0, cin, cin)])
    wp, bp = permute_mask_head(packed, bias)
    x, flow = fm[:, 256:], hx[:, 386:388]
    mask = torch.empty(M, 576, device=dev)
    a = torch.empty(B, 2, 8 * H, 8 * W, device=dev); b = torch.empty_like(a)
    ws = torch.zeros(ops.conv_workspace_bytes(), device=dev, dtype=torch.uint8)
    t_mk = timeit(lambda: ops.conv2d([x], B, H, W, 1, 1, packed, bias, 576, 0, False, 0.25, mask, None, None, None, ws))
    t_up = timeit(lambda: ops.convex_upsample_pm(flow, mask, a))
    t_fu = timeit(lambda: ops.mask_upsample(x, wp, bp, 0.25, flow, b))
    gf = 2.0 * M * 576 * cin / 1e9
    print(f"B={B:2d}: mask conv2 {t_mk:6.1f} us + upsample {t_up:5.1f} us = {t_mk + t_up:6.1f} us | fused {t_fu:6.1f} us "
          f"({gf / t_fu / 1e-3:5.1f} TFLOP/s = {gf / t_fu / 1e-3 / 157.3:.2f} of 157.3) | same={bool(torch.equal(a, b))}", flush=True)
