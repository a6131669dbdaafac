#!/usr/bin/env python3
"""Per-launch timeline of the two encoders (GPU box): every kernel of one fnet(both frames) + cnet(frame 1) call in issue order
with its duration (torch.profiler / roctracer), for batch 1 and 8 — where the encoders' share of a forward goes.
    python scripts/enc_prof.py [--batch 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptlflow_amd.raft import RAFT  # noqa: E402
from ptlflow_amd.synth import smooth_pair  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--tile", type=int, default=-1, help="force a GEMM tile configuration (pfk_debug_set_tile)")
    args = ap.parse_args()
    os.environ.setdefault("PFK_DEBUG_KNOBS", "1")
    dev = torch.device("cuda")
    m = RAFT().load_synthetic(1234).eval().to(dev)
    x = smooth_pair(args.batch, 436, 1024, seed=1).to(dev)
    from torch.profiler import ProfilerActivity, profile
    with torch.no_grad():
        xp, _ = m.preprocess(x)
        i1, i2 = xp[:, 0].contiguous(), xp[:, 1].contiguous()
        both = torch.cat([i1, i2], 0)
        fnet, cnet = m.encoders(dev)
        torch.ops.pfk.debug_set_tile(args.tile)
        for _ in range(2):
            fnet(both); cnet(i1)
        torch.cuda.synchronize()
        for name, fn in (("fnet", lambda: fnet(both)), ("cnet", lambda: cnet(i1))):
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                fn()
                torch.cuda.synchronize()
            evs = sorted([e for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA")], key=lambda e: e.time_range.start)
            tot = sum(e.device_time for e in evs) if evs and hasattr(evs[0], "device_time") else sum(e.cuda_time for e in evs)
            span = (evs[-1].time_range.end - evs[0].time_range.start) if evs else 0
            print(f"== {name} batch {args.batch}: {len(evs)} kernels, sum {tot/1e3:.3f} ms, span {span/1e3:.3f} ms")
            for e in evs:
                d = getattr(e, "device_time", None) or getattr(e, "cuda_time", 0.0)
                print(f"   {d:8.1f} us  {e.name[:110]}")


if __name__ == "__main__":
    main()
