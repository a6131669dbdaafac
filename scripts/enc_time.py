#!/usr/bin/env python3
"""Un-profiled wall time of the two encoders of the mirror (fnet on both frames, cnet on frame 1) at 436x1024, batch 1 and 8 (GPU box).
(scripts/enc_prof.py gives the per-launch timeline through torch.profiler, whose clocks run lower.)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptlflow_amd.raft import RAFT  # noqa: E402
from ptlflow_amd.synth import smooth_pair  # noqa: E402

dev = torch.device("cuda")
m = RAFT().load_synthetic(1234).eval().to(dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for B in (1, 8):
        x = smooth_pair(B, 436, 1024, seed=1).to(dev)
        xp, _ = m.preprocess(x)
        i1, i2 = xp[:, 0].contiguous(), xp[:, 1].contiguous()
        both = torch.cat([i1, i2], 0)
        fnet, cnet = m.encoders(dev)
        tf, tc = timeit(lambda: fnet(both)), timeit(lambda: cnet(i1))
        print(f"encoders batch {B}: fnet ({2 * B} images) {tf:.3f} ms, cnet ({B} images) {tc:.3f} ms, together {tf + tc:.3f} ms", flush=True)
