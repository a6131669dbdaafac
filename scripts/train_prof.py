#!/usr/bin/env python3
"""A few training steps of the RAFT mirror (BASELINE config 5 shape) for rocprofv3 --kernel-trace."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptlflow_amd.raft import RAFT
from ptlflow_amd.train import sequence_loss
import ptlflow_amd
ptlflow_amd.load_native()
dev = torch.device("cuda:0")
native = "--torch-encoders" not in sys.argv
model = RAFT(iters=12, native_encoders=native).load_synthetic(1234).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=4e-4, weight_decay=1e-4, eps=1e-8)
g = torch.Generator().manual_seed(99)
B, H, W = 10, 368, 496
inputs = {"images": torch.rand(B, 2, 3, H, W, generator=g).to(dev)}
gt = (torch.rand(B, 2, H, W, generator=g) * 20 - 10).to(dev)
valid = torch.ones(B, 1, H, W, device=dev)
for it in range(4):
    out = model(inputs)
    loss = sequence_loss(out["flow_preds"], gt, valid, 0.8, 400.0)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
torch.cuda.synchronize()
print("loss", float(loss))
