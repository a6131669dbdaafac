#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one forward vs how long the GPU needs to run it? (GPU box)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
from ptlflow_amd.raft import RAFT
from ptlflow_amd.synth import smooth_pair
ptlflow_amd.load_native()
m = RAFT(iters=32).load_synthetic(1).eval().cuda()
x = {"images": smooth_pair(1, 436, 1024).cuda()}
for _ in range(3):
    m(x)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); m(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
