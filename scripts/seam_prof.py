#!/usr/bin/env python3
"""A few forwards of the drop-in seam path — SeamRAFT (the reference's caller loop in torch) + patch.accelerate — for
rocprofv3 --kernel-trace; `--unpatched` traces the same object on stock PyTorch-ROCm ops instead."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
from ptlflow_amd import patch
from ptlflow_amd.raft import RAFT
from ptlflow_amd.seam_model import SeamRAFT
ptlflow_amd.load_native()
dev = torch.device("cuda:0")
state = RAFT(iters=32).load_synthetic(1234).state_dict()
m = SeamRAFT(iters=32).eval()
m.load_state_dict(state, strict=True)
m = m.to(dev)
if "--unpatched" not in sys.argv:
    patch.accelerate(m)
g = torch.Generator().manual_seed(1234)
x = {"images": torch.rand(1, 2, 3, 436, 1024, generator=g).to(dev)}
n = 3 if "--unpatched" in sys.argv else 8
for _ in range(n):
    out = m(x)
torch.cuda.synchronize()
print("forwards", n, "flows", tuple(out["flows"].shape))
