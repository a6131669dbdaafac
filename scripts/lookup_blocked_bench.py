#!/usr/bin/env python3
"""K3 on the row-major vs the blocked 4 x 8 volume layout (GPU box): time per lookup, equality of the outputs.
Fields: iid +-6 px noise on the identity grid (the round-1..4 benchmark field) and a smooth field (what a flow network produces).
At batch 8 the 2.1 GB pyramid is far beyond the 256 MB Infinity Cache: every lookup streams its windows from HBM, as in the forward."""
import os
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptlflow_amd
ptlflow_amd.load_native()
ops = torch.ops.pfk
dev = torch.device("cuda")
torch.manual_seed(0)


def blocked(p):
    M, h, w = p.shape
    th, tw = (h + 3) // 4, (w + 7) // 8
    q = torch.zeros(M, th * 4, tw * 8, device=p.device, dtype=p.dtype)
    q[:, :h, :w] = p
    return q.view(M, th, 4, tw, 8).permute(0, 1, 3, 2, 4).reshape(M, th * tw * 32).contiguous()


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for (h, w) in ((55, 128), (47, 156)):
    for B in (1, 8):
        L, r = 4, 4
        N = h * w
        lv, hh, ww = [], h, w
        for l in range(L):
            lv.append(torch.randn(B * N, hh, ww, device=dev)); hh //= 2; ww //= 2
        lb = [blocked(p) for p in lv]
        lh, lw = [p.shape[1] for p in lv], [p.shape[2] for p in lv]
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
        grid = torch.stack([xs, ys], 0)[None]
        smooth = torch.nn.functional.interpolate(torch.randn(B, 2, h // 8 + 2, w // 8 + 2, device=dev) * 4, size=(h, w), mode="bicubic", align_corners=True)
        for fname, coords in (("iid-6px", (grid + torch.randn(B, 2, h, w, device=dev) * 6).contiguous()), ("smooth", (grid + smooth).contiguous())):
            o1 = torch.empty(B * N, 324, device=dev); o2 = torch.empty(B * N, 324, device=dev)
            alg = B * (N * L * (100 + 81) * 4 + 8 * N)
            line = f"lookup {h}x{w} B={B} {fname:8s}:"
            for pix in (4, 8):
                ops.debug_set_lookup_pix(pix)
                t1 = timeit(lambda: ops.corr_lookup(lv, coords, r, o1))
                t2 = timeit(lambda: ops.corr_lookup_blocked(lb, lh, lw, coords, r, o2))
                same = bool(((o1 == o2) | (torch.isnan(o1) & torch.isnan(o2))).all())
                line += f" pix{pix}: row-major {t1:6.1f} us ({alg/t1/1e6:.2f} TB/s) blocked {t2:6.1f} us ({alg/t2/1e6:.2f} TB/s = {100*alg/t2/1e6/8:.1f}% of 8) same={same} |"
            ops.debug_set_lookup_pix(4)
            print(line, flush=True)
            if B == 8:      # bf16 maps (config 3), fp32 and bf16 output rows
                lb16 = [p.to(torch.bfloat16) for p in lb]
                o16 = torch.zeros(B * N, 328, device=dev, dtype=torch.bfloat16)
                line = f"  bf16 maps, blocked, {fname:8s}:"
                for pix in (4, 8):
                    ops.debug_set_lookup_pix(pix)
                    ta = timeit(lambda: ops.corr_lookup_blocked(lb16, lh, lw, coords, r, o2))
                    tb = timeit(lambda: ops.corr_lookup_blocked(lb16, lh, lw, coords, r, o16))
                    tc = timeit(lambda: ops.corr_lookup_blocked(lb, lh, lw, coords, r, o16))
                    line += f" pix{pix}: fp32 out {ta:6.1f} us | bf16 out {tb:6.1f} us | fp32 maps, bf16 out {tc:6.1f} us |"
                ops.debug_set_lookup_pix(4)
                print(line, flush=True)
