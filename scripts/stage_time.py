#!/usr/bin/env python3
"""Where a RAFT forward spends its GPU time, by stage (HIP events on the current stream, GPU box).
    python scripts/stage_time.py [--batch 8] [--conv-precision fp32]"""
import argparse
import os
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")   # tuning script: uses the pfk_debug_set_* knobs
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptlflow_amd.corr import CorrBlock  # noqa: E402
from ptlflow_amd.raft import RAFT  # noqa: E402
from ptlflow_amd.synth import smooth_pair  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--conv-precision", default="fp32")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--torch-encoders", action="store_true")
    ap.add_argument("--corr-tile", type=int, default=-1, help="force a GEMM tile configuration while timing the correlation volume")
    ap.add_argument("--enc-tiles", default="-1", help="comma list of GEMM tile configurations to force while timing the encoders")
    args = ap.parse_args()
    dev = torch.device("cuda")
    m = RAFT(conv_precision=args.conv_precision, native_encoders=not args.torch_encoders).load_synthetic(1234).eval().to(dev)
    x = smooth_pair(args.batch, 436, 1024, seed=1).to(dev)
    with torch.no_grad():
        xp, _ = m.preprocess(x)
        i1, i2 = xp[:, 0].contiguous(), xp[:, 1].contiguous()
        both = torch.cat([i1, i2], 0)
        t_pre, _ = timed(lambda: m.preprocess(x), args.reps)
        fnet, cnet = m.encoders(dev)      # the libpfk engines (or the torch modules with native_encoders=False)
        for tile in [int(t) for t in args.enc_tiles.split(",")][::-1]:      # the library's own choice (-1) last: its numbers go on
            torch.ops.pfk.debug_set_tile(tile)
            t_f, fm = timed(lambda: fnet(both), args.reps)
            t_c, _ = timed(lambda: cnet(i1), args.reps)
            print(f"encoders with tile configuration {tile}: fnet {t_f:.2f} ms, cnet {t_c:.2f} ms")
        torch.ops.pfk.debug_set_tile(-1)
        B = args.batch
        torch.ops.pfk.debug_set_tile(args.corr_tile)
        t_corr, _ = timed(lambda: CorrBlock(fm[:B], fm[B:], num_levels=4, radius=4), args.reps)
        torch.ops.pfk.debug_set_tile(-1)
        t_all, _ = timed(lambda: m({"images": x}), args.reps)
    rest = t_all - t_pre - t_f - t_c - t_corr
    print(f"batch {B} {args.conv_precision}: forward {t_all:.2f} ms = preprocess {t_pre:.2f} + fnet {t_f:.2f} + cnet {t_c:.2f} "
          f"+ corr volume/pyramid {t_corr:.2f} + 32-iteration loop and rest {rest:.2f}")


if __name__ == "__main__":
    main()
