#!/usr/bin/env python3
"""raft with conv_precision="bf16" (K8b: bf16 activation storage) next to the fp32 path on the GPU box: EPE between the two on a smooth
pair and the time per forward at batch 8 / 1 (quick A/B while tuning; the gates proper are tests/test_gpu_bf16_gate.py).
    python scripts/b16_forward_check.py [--batch 8] [--model raft|gma] [--skip-dead]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raft_oracle as O  # noqa: E402  (smooth_pair / epe only: checker side)
from ptlflow_amd.raft import GMA, RAFT  # noqa: E402


def timed(model, x, n=6):
    model(x); torch.cuda.synchronize()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); model(x); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    t.sort()
    return t[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--model", default="raft")
    ap.add_argument("--iters", type=int, default=32)
    ap.add_argument("--skip-dead", action="store_true")
    ap.add_argument("--precisions", default="fp32,bf16")
    ap.add_argument("--cfg", type=int, default=0, help="force a K8b tile configuration (needs PFK_DEBUG_KNOBS=1)")
    ap.add_argument("--graph", action="store_true", help="use_graph=True: the iteration loop replayed from a captured hipGraph")
    ap.add_argument("--no-overlap", action="store_true", help="mask head + upsampling on the main stream")
    ap.add_argument("--no-fuse", action="store_true", help="mask conv2 and the upsampling as two launches")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.cfg:
        import ptlflow_amd
        ptlflow_amd.load_native()
        torch.ops.pfk.debug_set_b16(args.cfg)
    cls = GMA if args.model == "gma" else RAFT
    kw = dict(iters=args.iters, upsample_every_iter=not args.skip_dead)
    base = cls(**kw).load_synthetic(1234).eval()
    P = base.state_dict()
    x = {"images": O.smooth_pair(args.batch, 436, 1024, 1234).to(dev)}
    ref = None
    for prec in args.precisions.split(","):
        m = cls(conv_precision=prec, **kw, **({"use_graph": True} if args.graph else {})).eval()
        m.load_state_dict(P)
        if args.no_overlap:
            m.overlap_mask_head = False
        if args.no_fuse:
            m.fuse_mask_upsample = False
        m = m.to(dev)
        out = m(x)["flows"][:, 0].float()
        torch.cuda.synchronize()
        line = f"{args.model} {prec:6s} batch {args.batch}: "
        if ref is None:
            ref = out
        else:
            mean, mx = O.epe(out.cpu(), ref.cpu())
            line += f"EPE vs {args.precisions.split(',')[0]} mean {mean:.3e} max {mx:.3e} | "
        sec = timed(m, x)
        line += f"{1e3 * sec:7.2f} ms / forward = {args.batch / sec:6.1f} pairs/s | finite {bool(torch.isfinite(out).all())}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
