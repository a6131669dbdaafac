#!/usr/bin/env python3
"""Time the pyramid lookup (K3) and the on-demand correlation (K7) on the north-star shape (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
ptlflow_amd.load_native()
ops = torch.ops.pfk
dev = torch.device("cuda")
torch.manual_seed(0)
for B in (1, 8):
    h, w, L, r = 55, 128, 4, 4
    N = h * w
    lv, hh, ww = [], h, w
    for l in range(L):
        lv.append(torch.randn(B * N, hh, ww, device=dev)); hh //= 2; ww //= 2
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xs, ys], 0)[None] + torch.randn(B, 2, h, w, device=dev) * 6).contiguous()
    out = torch.empty(B * N, 324, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ref = None
    for variant in (0,):
        for _ in range(3):
            ops.corr_lookup(lv, coords, r, out)
        e0.record()
        for _ in range(50):
            ops.corr_lookup(lv, coords, r, out)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 50
        alg = B * (N * L * (100 + 81) * 4 + 8 * N)
        same = True if ref is None else bool(torch.equal(ref, out))
        ref = out.clone() if ref is None else ref
        print(f"lookup B={B} variant {variant}: {us:.1f} us, algorithmic {alg/1e6:.1f} MB -> {alg/us/1e6:.2f} TB/s ({100*alg/us/1e6/8:.1f}% of 8 TB/s) same={same}")
    f1 = torch.randn(B, h, w, 256, device=dev); f2 = torch.randn(B, h, w, 256, device=dev)
    c5 = coords.permute(0, 2, 3, 1).reshape(B, 1, h, w, 2).contiguous()
    for _ in range(3):
        ops.altcorr_forward(f1, f2, c5, r)
    e0.record()
    for _ in range(20):
        ops.altcorr_forward(f1, f2, c5, r)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print(f"altcorr level-0 B={B}: {us:.1f} us ({B*N*100*256*2/us/1e6:.2f} TFLOP/s, {B*N*100*1024/us/1e6:.2f} TB/s of L2 gathers)")
