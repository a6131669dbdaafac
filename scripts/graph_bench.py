#!/usr/bin/env python3
"""Eager vs hipGraph-replayed iteration loop (GPU box).  python scripts/graph_bench.py [--height 436 --width 1024 --batch 1]"""
import os
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")   # tuning script: uses the pfk_debug_set_* knobs
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptlflow_amd.raft import RAFT  # noqa: E402
from ptlflow_amd.synth import smooth_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=436)
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--conv-precision", default="fp32")
ap.add_argument("--swizzled", action="store_true", help="stream-K and small tile grids on the 48 KB swizzled LDS layout (room for a third block per CU)")
args = ap.parse_args()
x = smooth_pair(args.batch, args.height, args.width, seed=1).cuda()
if args.swizzled:
    import ptlflow_amd
    ptlflow_amd.load_native()
    torch.ops.pfk.debug_set_tile(207)     # stream-K: swizzled, XCD groups, two blocks per CU
    torch.ops.pfk.debug_set_tile(301)     # small tile grids: swizzled
for use_graph, fork in ((False, False), (False, True), (True, None), (True, False)):
    m = RAFT(use_graph=use_graph, fork_branches=fork, conv_precision=args.conv_precision).load_synthetic(1).eval().cuda()
    for _ in range(3):
        m({"images": x})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m({"images": x})
    torch.cuda.synchronize()
    ms = 1e2 * (time.perf_counter() - t0)
    print(f"{args.height}x{args.width} batch {args.batch} {args.conv_precision} swizzled={args.swizzled} use_graph={use_graph} fork_branches={fork}: {ms:.2f} ms / forward ({args.batch * 1e3 / ms:.1f} pairs/s)")
