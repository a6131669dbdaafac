#!/usr/bin/env python3
"""Micro-benchmark + cross-check of pfk::conv_wgrad (K12) on the training shapes of BASELINE config 5 (GPU box).
    python scripts/wgrad_bench.py [--variants 0,4,2,1] [--reps 20]
variant 0 = the library's tile height (least channel padding), 4 / 2 / 1 = forced 128 / 64 / 32 output channels per tile
(pfk_debug_set_wgrad).  The check is the last variant against a float64
torch matmul of the unfolded input on the small shapes, and variant-vs-variant everywhere (different split counts => the sums
differ in the last bits; the gate is relative to the gradient's scale); the parity gate proper is tests/test_gpu_train.py."""
import argparse
import os
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")   # tuning script: uses the pfk_debug_set_* knobs
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd  # noqa: E402

ptlflow_amd.load_native()
ops = torch.ops.pfk

# name, B, H, W, cin segments, cout, kh, kw, with_bias
UB = (10, 46, 62)
SHAPES = [
    ("c1", *UB, [324], 256, 1, 1, True), ("c2", *UB, [256], 192, 3, 3, True), ("f2", *UB, [128], 64, 3, 3, True),
    ("cv", *UB, [256], 128, 3, 3, True), ("zr1", *UB, [128, 256], 256, 1, 5, True), ("q1", *UB, [128, 256], 128, 1, 5, True),
    ("zr2", *UB, [128, 256], 256, 5, 1, True), ("fh1", *UB, [128], 256, 3, 3, True), ("mk1", *UB, [128], 256, 3, 3, True),
    ("mk2", *UB, [256], 576, 1, 1, True),
    ("enc1", 20, 184, 248, [64], 64, 3, 3, True), ("enc2", 20, 92, 124, [96], 96, 3, 3, True),
    ("enc3", 20, 46, 62, [128], 128, 3, 3, True),
]


def reference(xs, dy, B, H, W, kh, kw):
    """dW[co, c, ky, kx] in float64 through unfold (small shapes only)."""
    x = torch.cat(xs, 1).double().view(B, H, W, -1).permute(0, 3, 1, 2)
    cols = F.unfold(x, (kh, kw), padding=(kh // 2, kw // 2))            # [B, C*kh*kw, H*W]
    g = dy.double().view(B, H * W, -1)
    return torch.einsum("bkp,bpo->ok", cols, g)                          # [cout, C*kh*kw] (channel-major, tap-minor)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,4,2,1")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tot = {v: 0.0 for v in variants}
    for name, B, H, W, segs, cout, kh, kw, wb in SHAPES:
        M = B * H * W
        xs = [torch.randn(M, c, device=dev) for c in segs]
        dy = torch.randn(M, cout, device=dev)
        ktot = sum(kh * kw * ((c + 31) // 32 * 32) for c in segs)
        flops = 2.0 * M * cout * kh * kw * sum(segs)
        outs = {}
        line = f"{name:5s} M={M:7d} cout={cout:3d} ktot={ktot:5d} {flops/1e9:6.1f} GF |"
        for v in variants:
            ops.debug_set_wgrad(v)
            out = torch.zeros(cout, ktot + (32 if wb else 0), device=dev)
            ops.conv_wgrad(xs, dy, B, H, W, kh, kw, out, wb)
            torch.cuda.synchronize()
            outs[v] = out.clone()
            for _ in range(2):
                ops.conv_wgrad(xs, dy, B, H, W, kh, kw, out, wb)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                ops.conv_wgrad(xs, dy, B, H, W, kh, kw, out, wb)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.reps
            tot[v] += us
            line += f" v{v}: {us:7.1f} us {flops/us/1e6:6.1f} TF |"
        scale = outs[variants[0]].abs().max().item()
        for v in variants[1:]:
            line += f" |v{v}-v{variants[0]}|/scale {(outs[v] - outs[variants[0]]).abs().max().item()/scale:.1e}"
        if M <= 30000 and len(segs) == 1 and segs[0] % 32 == 0:
            ref = reference(xs, dy, B, H, W, kh, kw)                     # [cout, C*taps] channel-major
            C = segs[0]
            got = outs[variants[-1]][:, :ktot].view(cout, kh * kw, C).permute(0, 2, 1).reshape(cout, -1).double()
            line += f" | vs f64 {(got - ref).abs().max().item()/ref.abs().max().item():.1e}"
            if wb:
                line += f" bias {(outs[variants[-1]][:, ktot].double() - dy.double().sum(0)).abs().max().item()/dy.double().sum(0).abs().max().item():.1e}"
        print(line, flush=True)
    ops.debug_set_wgrad(0)
    print("sum us:", {v: round(t, 1) for v, t in tot.items()})


if __name__ == "__main__":
    main()
