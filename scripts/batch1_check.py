#!/usr/bin/env python3
"""Batch-1 forward (the reference's model_benchmark.py protocol shape: one 436x1024 pair) with the grouped launches of the motion encoder
on and off: time per forward, equality of the outputs (GPU box).
    python scripts/batch1_check.py [--skip-dead] [--model raft|gma]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptlflow_amd.raft import GMA, RAFT  # noqa: E402
from ptlflow_amd.synth import smooth_pair  # noqa: E402


def timed(model, x, n=20):
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); model(x); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    t.sort()
    return t[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="raft")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--skip-dead", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda")
    cls = GMA if args.model == "gma" else RAFT
    x = {"images": smooth_pair(args.batch, 436, 1024, 1234).to(dev)}
    outs = {}
    for grouped in (False, True, False, True):
        m = cls(iters=32, upsample_every_iter=not args.skip_dead).load_synthetic(1234).eval().to(dev)
        m.group_launches = grouped
        out = m(x)["flows"]
        sec = timed(m, x)
        outs[grouped] = out
        print(f"{args.model} batch {args.batch} grouped={grouped!s:5}: {1e3 * sec:7.3f} ms / forward = {args.batch / sec:6.2f} pairs/s", flush=True)
    print("identical outputs:", bool(torch.equal(outs[False], outs[True])))


if __name__ == "__main__":
    main()
