#!/usr/bin/env python3
"""convf1 (7x7, 2 -> 128) — the tiled VALU kernel against the MFMA kernel, batch 1 / 8 at 55x128, fp32 and bf16 rows out (GPU box)."""
import os
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd  # noqa: E402
from ptlflow_amd.packing import pack_cin2_weight  # noqa: E402
ptlflow_amd.load_native()
ops = torch.ops.pfk
dev = torch.device("cuda")


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for B in (1, 8):
    H, W, cout = 55, 128, 128
    M = B * H * W
    hx = torch.randn(M, 384, device=dev)
    w = pack_cin2_weight(torch.randn(cout, 2, 7, 7) / 10).to(dev)
    bias = torch.randn(cout, device=dev)
    for dt in (torch.float32, torch.bfloat16):
        out = torch.empty(M, cout, device=dev, dtype=dt)
        t = {}
        for valu in (1, 2):
            ops.debug_set_cin2_valu(valu)
            t[valu] = timeit(lambda: ops.conv_cin2(hx[:, 382:384], w, bias, out, B, H, W, 7, True))
        ops.debug_set_cin2_valu(0)
        print(f"batch {B} {str(dt):15s}: VALU {t[1]:6.1f} us | MFMA {t[2]:6.1f} us", flush=True)
