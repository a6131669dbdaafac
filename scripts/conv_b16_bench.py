#!/usr/bin/env python3
"""Micro-benchmark + quick correctness of pfk::conv2d_b16 (K8b: bf16 activation storage, LDS-DMA for both operands) on the
update-block shapes (GPU box), next to the split-bf16 kernel with one plane (fp32 activations, `pfk_conv2d_bf16s`).
    python scripts/conv_b16_bench.py [--batch 8] [--cfgs 0,1,2] [--reps 30] [--rounds 3]
cfg 0 = the library's choice, 1..5 = a forced tile configuration (pfk_gemm_b16.hip::launch_b16); cfg 100 = the split kernel, one plane.
Reference = torch conv2d (fp32, MIOpen) on the bf16-rounded operands: only to catch wrong results quickly; the parity gate proper is
tests/ against the CPU oracle."""
import argparse
import math
import os
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")
os.environ.setdefault("PFK_BENCH_VARIANTS", "1")   # the ablation / experimental configurations (cfg >= 10) need a variants build
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd  # noqa: E402
from ptlflow_amd.packing import pack_conv_weight, split_bf16_planes  # noqa: E402

ptlflow_amd.load_native()
ops = torch.ops.pfk

# name, cin segments, cout, kh, kw, epilogue, out dtype
SHAPES = [
    ("c1", [328], 256, 1, 1, 0), ("c2", [256], 192, 3, 3, 0), ("f2", [128], 64, 3, 3, 0), ("cv", [256], 126, 3, 3, 0),
    ("zr1h", [128, 128], 256, 1, 5, 1), ("q1h", [128, 128], 128, 1, 5, 2), ("zr2h", [128, 128], 256, 5, 1, 1), ("q2h", [128, 128], 128, 5, 1, 2),
    ("fm", [128], 512, 3, 3, 0), ("mk", [256], 576, 1, 1, 0),
    ("zr1hL", [128, 128], 256, 1, 5, 0), ("q1hL", [128, 128], 128, 1, 5, 0),     # the GRU launches' GEMMs with the plain epilogue
    ("zr1g", [128, 256], 256, 1, 5, 1), ("q1g", [128, 256], 128, 1, 5, 2),       # GMA width (h | motion | aggregated motion)
    ("l1", [64], 64, 3, 3, 0),       # BasicEncoder layer 1 (run with --H 218 --W 512 --batch 16: fnet at the headline's batch 8)
    ("l2", [96], 96, 3, 3, 0),       # layer 2 (--H 109 --W 256 --batch 16)
    ("l3", [128], 128, 3, 3, 0),     # layer 3 (--H 55 --W 128 --batch 16)
    ("lo", [128], 256, 1, 1, 0),     # the output 1x1 convolution (--H 55 --W 128 --batch 16)
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--cfgs", default="0,100")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--H", type=int, default=55)
    ap.add_argument("--W", type=int, default=128)
    ap.add_argument("--only", default="")
    ap.add_argument("--residual", type=int, default=1, help="GRU epilogues of the K8b kernel with the bf16 context term (as the engine runs them)")
    args = ap.parse_args()
    cfgs = [int(c) for c in args.cfgs.split(",")]
    B, H, W = args.batch, args.H, args.W
    M = B * H * W
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tot = {c: 0.0 for c in cfgs}
    for name, segs, cout, kh, kw, epi in SHAPES:
        if args.only and name not in args.only.split(","):
            continue
        cin = sum(segs)
        xs = [torch.randn(M, c, device=dev).to(torch.bfloat16) for c in segs]
        wt = (torch.randn(cout, cin, kh, kw, device=dev) / math.sqrt(cin * kh * kw)).to(torch.bfloat16).float()
        bias = torch.randn(cout, device=dev) * 0.1
        offs, o = [], 0
        for c in segs:
            offs.append((o, c, c)); o += c
        x_nchw = torch.cat([x.float() for x in xs], 1).view(B, H, W, cin).permute(0, 3, 1, 2).contiguous()
        ref = F.conv2d(x_nchw, wt, bias, padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1).reshape(M, cout)
        Ch = cout // 2 if epi == 1 else cout
        hbuf0 = torch.tanh(torch.randn(M, Ch, device=dev))
        zbuf0 = torch.rand(M, Ch, device=dev)
        flops = 2.0 * M * cout * kh * kw * cin
        line = f"{name:5s} cout={cout:3d} K={kh*kw*cin:5d} {flops/1e9:5.2f} GF |"
        runs = []
        for cfg in cfgs:
            if cfg >= 100:      # the split kernel: fp32 activations, one bf16 plane
                ops.debug_set_tile(-1)
                packed = split_bf16_planes(pack_conv_weight(wt, offs), 1)
                xf = [x.float() for x in xs]
                out = torch.zeros(M, cout, device=dev)
                hbuf, zbuf, rh = hbuf0.clone(), zbuf0.clone(), torch.zeros(M, Ch, device=dev)

                def run(packed=packed, out=out, hbuf=hbuf, zbuf=zbuf, rh=rh, xf=xf):
                    if epi == 0:
                        ops.conv2d(xf, B, H, W, kh, kw, packed, bias, cout, 0, False, 1.0, out, None, None, None, None)
                    elif epi == 1:
                        ops.conv2d(xf, B, H, W, kh, kw, packed, bias, cout, 1, False, 1.0, None, hbuf, zbuf, rh, None)
                    else:
                        ops.conv2d(xf, B, H, W, kh, kw, packed, bias, cout, 2, False, 1.0, None, hbuf, zbuf, None, None)
            else:
                ops.debug_set_b16(cfg)
                packed = pack_conv_weight(wt, offs, kpad=64).to(torch.bfloat16)
                out = torch.zeros(M, (cout + 7) // 8 * 8, device=dev, dtype=torch.bfloat16)[:, :cout]
                hbuf, zbuf = hbuf0.clone(), zbuf0.to(torch.bfloat16)
                rh = torch.zeros(M, Ch, device=dev, dtype=torch.bfloat16)
                hb = hbuf0.to(torch.bfloat16)
                res = (torch.randn(M, cout, device=dev) * 0.1).to(torch.bfloat16) if args.residual else None

                def run(packed=packed, out=out, hbuf=hbuf, zbuf=zbuf, rh=rh, hb=hb, cfg=cfg, res=res):
                    ops.debug_set_b16(cfg)
                    if epi == 0:
                        ops.conv2d_b16(xs, B, H, W, kh, kw, packed, bias, cout, 0, False, 1.0, out)
                    elif epi == 1:
                        ops.conv2d_b16(xs, B, H, W, kh, kw, packed, bias, cout, 1, False, 1.0, None, None, hb, zbuf, rh, res)
                    else:
                        ops.conv2d_b16(xs, B, H, W, kh, kw, packed, bias, cout, 2, False, 1.0, None, hbuf, hb, zbuf, None, res)
            try:
                run()
            except RuntimeError as e:
                line += f" cfg{cfg:4d}: {str(e)[:40]} |"
                continue
            torch.cuda.synchronize()
            if epi == 0:
                err = (out.float() - ref).abs().max().item()
            elif epi == 1:
                r = ref + (res.float() if cfg < 100 and res is not None else 0)
                g = torch.sigmoid(r)
                hh = hbuf0.to(torch.bfloat16).float() if cfg < 100 else hbuf0
                err = max((zbuf.float() - g[:, :Ch]).abs().max().item(), (rh.float() - g[:, Ch:] * hh).abs().max().item())
            else:
                r = ref + (res.float() if cfg < 100 and res is not None else 0)
                q = torch.tanh(r)
                zz = zbuf0.to(torch.bfloat16).float() if cfg < 100 else zbuf0
                err = (hbuf - ((1 - zz) * hbuf0 + zz * q)).abs().max().item()
            for _ in range(3):
                run()
            runs.append((cfg, run, [], err))
        for _ in range(args.rounds):
            for cfg, run, samples, _err in runs:
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
    ops.debug_set_b16(0)
    print("sum us per iteration:", {c: round(v, 1) for c, v in tot.items()})


if __name__ == "__main__":
    main()
