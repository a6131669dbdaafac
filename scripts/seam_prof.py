#!/usr/bin/env python3
"""A few forwards of the drop-in seam path — the reference's own `ptlflow.models.raft.raft.RAFT` (oracle/ref_loader.py: /root/reference
or the archive staged for the GPU box; nothing is substituted where neither exists) + patch.accelerate — for rocprofv3 --kernel-trace;
`--unpatched` traces the same object on stock PyTorch-ROCm ops instead."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
from ptlflow_amd import patch
from ptlflow_amd.raft import RAFT
ptlflow_amd.load_native()
dev = torch.device("cuda:0")
state = RAFT(iters=32).load_synthetic(1234).state_dict()
try:
    from oracle import ref_loader
    real = ref_loader.reference_available()
except Exception:
    real = False
if not real:
    raise SystemExit("the reference is not importable here (no /root/reference, no oracle/_ref archive): nothing to trace")
m = ref_loader.build_raft(iters=32)
m.load_state_dict(state, strict=False)
m = m.eval()
print("model class:", type(m).__module__ + "." + type(m).__name__)
m = m.to(dev)
if "--unpatched" not in sys.argv:
    patch.accelerate(m)
g = torch.Generator().manual_seed(1234)
x = {"images": torch.rand(1, 2, 3, 436, 1024, generator=g).to(dev)}
n = 3 if "--unpatched" in sys.argv else 8
with torch.no_grad():
    for _ in range(n):
        out = m(x)
torch.cuda.synchronize()
print("forwards", n, "flows", tuple(out["flows"].shape))
