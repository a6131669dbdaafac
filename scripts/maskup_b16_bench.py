#!/usr/bin/env python3
"""K13b (pfk_mask_upsample_b16) against the two K8b launches it replaces, standalone at batch 1 / 8 (GPU box)."""
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd  # noqa: E402
from ptlflow_amd.packing import pack_conv_weight, permute_mask_head  # noqa: E402
ptlflow_amd.load_native()
ops = torch.ops.pfk
dev = torch.device("cuda")
BF = torch.bfloat16


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for B in (1, 8):
    H, W, cin = 55, 128, 256
    M = B * H * W
    torch.manual_seed(0)
    fm = torch.randn(M, 512, device=dev).to(BF)
    wt = torch.randn(576, cin, 1, 1) / math.sqrt(cin)
    bias = torch.randn(576, device=dev) * 0.1
    hx = torch.randn(M, 8, device=dev)
    packed = pack_conv_weight(wt, [(0, cin, cin)], kpad=64).to(BF).to(dev)
    wp, bp = permute_mask_head(pack_conv_weight(wt, [(0, cin, cin)]), bias.cpu())
    wp, bp = wp.to(dev, BF), bp.to(dev)
    x, flow = fm[:, 256:], hx[:, 4:6]
    mask = torch.empty(M, 576, device=dev, dtype=BF)
    a, b = torch.empty(B, 2, 8 * H, 8 * W, device=dev), torch.empty(B, 2, 8 * H, 8 * W, device=dev)
    t_mk = timeit(lambda: ops.conv2d_b16([x], B, H, W, 1, 1, packed, bias, 576, 0, False, 0.25, mask))
    t_up = timeit(lambda: ops.convex_upsample_pm(flow, mask, a))
    t_f = timeit(lambda: ops.mask_upsample(x, wp, bp, 0.25, flow, b))
    print(f"batch {B}: mask conv2 {t_mk:.1f} us + upsample {t_up:.1f} us = {t_mk + t_up:.1f} us | fused K13b {t_f:.1f} us | identical {bool(torch.equal(a, b))}", flush=True)
