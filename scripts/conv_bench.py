#!/usr/bin/env python3
"""Micro-benchmark + correctness of pfk::conv2d on the update-block shapes (GPU box).
    python scripts/conv_bench.py [--batch 1] [--cfgs -1,1,5] [--reps 30]
cfg -1 = the library's own choice, 0..26 = a forced fp32 tile configuration; 100*n + t = split-bf16 with n planes (1..3) and
tile configuration t (0 = heuristic, 1 = 64x64, 2 = 128x64, 3 = 128x128), e.g. 200, 203, 302;
+1000*d = timing ablation d of the split kernels (1 no global loads, 2 no split/LDS stores, 4 no MFMAs; results are garbage).
Reference = torch conv2d on the same GPU (MIOpen fp32), only to catch wrong results quickly; the parity gate
proper is tests/ against the CPU oracle."""
import argparse
import math
import os
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")   # tuning script: uses the pfk_debug_set_* knobs
os.environ.setdefault("PFK_BENCH_VARIANTS", "1")   # the ablation / schedule-sweep configurations are compiled into a variants build only (ptlflow_amd/_build.py)
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd  # noqa: E402
from ptlflow_amd.packing import pack_conv_weight, split_bf16_planes  # noqa: E402

ptlflow_amd.load_native()
ops = torch.ops.pfk

# name, cin segments, cout, kh, kw, epilogue
SHAPES = [
    ("c1", [324], 256, 1, 1, 0), ("c2", [256], 192, 3, 3, 0), ("f2", [128], 64, 3, 3, 0), ("cv", [256], 126, 3, 3, 0),
    ("zr1", [384], 256, 1, 5, 1), ("q1", [128, 256], 128, 1, 5, 2), ("zr2", [384], 256, 5, 1, 1), ("q2", [128, 256], 128, 5, 1, 2),
    ("fm", [128], 512, 3, 3, 0), ("mk", [256], 576, 1, 1, 0),
    # round 4: the GRU launches with the context slice hoisted out of the loop (h | motion features: 128 + 128 input channels)
    ("zr1h", [128, 128], 256, 1, 5, 1), ("q1h", [128, 128], 128, 1, 5, 2), ("zr2h", [128, 128], 256, 5, 1, 1), ("q2h", [128, 128], 128, 5, 1, 2),
    # the same two launches with the plain linear epilogue: what the gate arithmetic (sigmoid / tanh / blend, h and z reads) costs
    ("zr1hL", [128, 128], 256, 1, 5, 0), ("q1hL", [128, 128], 128, 1, 5, 0),
]

# --shapes enc: the BasicEncoder's convolutions (raft/extractor.py:122-195) at 440x1024; (name, cin, cout, k, epi, H, W, stride)
# with H, W the INPUT resolution; batch = images (fnet at the headline's batch 8 sees 16)
ENC_SHAPES = [
    ("l1", [64], 64, 3, 3, 0, 220, 512, 1), ("l2s", [64], 96, 3, 3, 0, 220, 512, 2), ("l2", [96], 96, 3, 3, 0, 110, 256, 1),
    ("l2d", [64], 96, 1, 1, 0, 220, 512, 2), ("l3s", [96], 128, 3, 3, 0, 110, 256, 2), ("l3", [128], 128, 3, 3, 0, 55, 128, 1),
    ("out", [128], 256, 1, 1, 0, 55, 128, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--cfgs", default="-1")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--H", type=int, default=55)
    ap.add_argument("--W", type=int, default=128)
    ap.add_argument("--only", default="")
    ap.add_argument("--shapes", default="update", choices=["update", "enc"])
    ap.add_argument("--rounds", type=int, default=1, help="time every cfg `rounds` times in round-robin order and report the median "
                    "(box-to-box and warm-up drift is +-5 %: only interleaved A/B numbers of one run are comparable)")
    args = ap.parse_args()
    cfgs = [int(c) for c in args.cfgs.split(",")]
    B, H, W = args.batch, args.H, args.W
    M = B * H * W
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tot = {c: 0.0 for c in cfgs}
    ws = torch.zeros(ops.conv_workspace_bytes(), device=dev, dtype=torch.uint8)
    for shape in (ENC_SHAPES if args.shapes == "enc" else SHAPES):
        name, segs, cout, kh, kw, epi = shape[:6]
        stride = 1
        if len(shape) > 6:
            H, W, stride = shape[6:]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        Min, M = B * H * W, B * Ho * Wo
        if args.only and name not in args.only.split(","):
            continue
        cin = sum(segs)
        xs = [torch.randn(Min, c, device=dev) for c in segs]
        wt = torch.randn(cout, cin, kh, kw, device=dev) / math.sqrt(cin * kh * kw)
        bias = torch.randn(cout, device=dev) * 0.1
        offs, o = [], 0
        for c in segs:
            offs.append((o, c, c)); o += c
        packed = pack_conv_weight(wt, offs)
        x_nchw = torch.cat(xs, 1).view(B, H, W, cin).permute(0, 3, 1, 2).contiguous()
        ref = F.conv2d(x_nchw, wt, bias, stride=stride, padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1).reshape(M, cout)
        Ch = cout // 2 if epi == 1 else cout
        hbuf0 = torch.tanh(torch.randn(M, Ch, device=dev))
        zbuf0 = torch.rand(M, Ch, device=dev)
        flops = 2.0 * M * cout * kh * kw * cin
        line = f"{name:4s} cout={cout:3d} K={kh*kw*cin:5d} {flops/1e9:5.2f} GF |"
        runs = []
        for cfg in cfgs:
            ops.debug_set_tile(-1)
            ops.debug_set_tile(cfg if cfg < 100 else 100 + cfg % 100 + 10 * (cfg // 1000))
            packed = (pack_conv_weight(wt, offs) if cfg < 100
                      else split_bf16_planes(pack_conv_weight(wt, offs), (cfg // 100) % 10))
            out = torch.zeros(M, cout, device=dev)
            hbuf, zbuf, rh = hbuf0.clone(), zbuf0.clone(), torch.zeros(M, Ch, device=dev)

            def run(packed=packed, out=out, hbuf=hbuf, zbuf=zbuf, rh=rh):      # bound now: later rounds call it again
                if epi == 0:
                    ops.conv2d(xs, B, H, W, kh, kw, packed, bias, cout, 0, False, 1.0, out, None, None, None, ws, None, stride, False)
                elif epi == 1:
                    ops.conv2d(xs, B, H, W, kh, kw, packed, bias, cout, 1, False, 1.0, None, hbuf, zbuf, rh, ws)
                else:
                    ops.conv2d(xs, B, H, W, kh, kw, packed, bias, cout, 2, False, 1.0, None, hbuf, zbuf, None, ws)
            try:
                run()
            except RuntimeError as e:       # a configuration that does not implement this epilogue (the timing ablations)
                line += f" cfg{cfg:4d}: unsupported |"
                continue
            torch.cuda.synchronize()
            if epi == 0:
                err = (out - ref).abs().max().item()
                first = out.clone()          # stream-K: the fix-up order is fixed, so a second launch must reproduce every bit
                run()
                torch.cuda.synchronize()
                if not torch.equal(first, out):
                    err = float("inf")
            elif epi == 1:
                g = torch.sigmoid(ref)
                err = max((zbuf - g[:, :Ch]).abs().max().item(), (rh - g[:, Ch:] * hbuf0).abs().max().item())
            else:
                q = torch.tanh(ref)
                err = (hbuf - ((1 - zbuf0) * hbuf0 + zbuf0 * q)).abs().max().item()
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.reps
            runs.append((cfg, run, [us], err))
        for _ in range(args.rounds - 1):         # further rounds, round-robin over the configurations
            for cfg, run, samples, _err in runs:
                ops.debug_set_tile(-1)
                ops.debug_set_tile(cfg if cfg < 100 else 100 + cfg % 100 + 10 * (cfg // 1000))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                samples.append(1e3 * e0.elapsed_time(e1) / args.reps)
        for cfg, run, samples, err in runs:
            samples.sort()
            us = samples[len(samples) // 2]
            tot[cfg] += us
            line += f" cfg{cfg:4d}: {us:7.1f} us {flops/us/1e6:6.1f} TF err {err:.1e} |"
        print(line, flush=True)
    ops.debug_set_tile(-1)
    off = ops.conv_workspace_fault_offset()
    print("stream-K faults:", int(ws[off:off + 4].view(torch.int32).item()), "| flag region all zero after the runs:",
          bool((ws[ops.conv_workspace_bytes() - 768 * 64:][:768 * 4] == 0).all().item()))
    print("sum us per iteration:", {c: round(v, 1) for c, v in tot.items()})


if __name__ == "__main__":
    main()
