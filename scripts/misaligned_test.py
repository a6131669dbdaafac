#!/usr/bin/env python3
"""One-off check: sources whose base pointer is only 4-byte aligned (buffer_load_dwordx4 at dword alignment)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
from ptlflow_amd.packing import pack_conv_weight
ptlflow_amd.load_native()
ops = torch.ops.pfk
torch.manual_seed(0)
M, K, N = 96, 4000, 64
buf = torch.randn(M, K + 8, device="cuda")
w = torch.randn(N, K, 1, 1, device="cuda") / math.sqrt(K)
packed = pack_conv_weight(w, [(0, K, K)])
for off in (0, 1, 2, 3, 5):
    src = buf[:, off:off + K]
    out = torch.zeros(M, N, device="cuda")
    ops.conv2d([src], 1, 1, M, 1, 1, packed, None, N, 0, False, 1.0, out, None, None, None)
    torch.cuda.synchronize()
    ref = src.double() @ w.view(N, K).double().t()
    print("offset", off, "max err", (out.double() - ref).abs().max().item())
