#!/usr/bin/env python3
"""The reference's OWN model classes on one MI355X: stock PyTorch-ROCm ops against the same object after
`ptlflow_amd.patch.accelerate` — what a ptlflow user gets by adding that one call (GPU box; classes from the staged archive).

    python scripts/dropin_speedup.py FAMILY MODULE CLASS [--kw '{"block_dims": [64, 128, 256]}'] [--H 436 --W 1024]

Protocol = the reference's model_benchmark.py: batch 1, torch.rand input, warm-up, 10 synchronised forwards, median.  The stock
side gets 4 warm-up forwards (MIOpen picks its kernels during the first ones).  Prints one JSON line."""
import argparse
import json
import os
import statistics
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_loader  # noqa: E402  (checker-side script: loads the reference's classes)


def timed(model, x, n=10, warm=2):
    with torch.no_grad():
        for _ in range(warm):
            out = model({"images": x})["flows"]
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            out = model({"images": x})["flows"]
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
    return statistics.median(ts), out[:, 0].float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("family"); ap.add_argument("module"); ap.add_argument("cls")
    ap.add_argument("--kw", default="{}")
    ap.add_argument("--stock-kw", default=None, help="constructor arguments of the stock side when they must differ")
    ap.add_argument("--stock-without-plugin", action="store_true",
                    help="ccmr / ms_raft_plus: their default alternate_corr=True needs the compiled alt_cuda_corr extension, which "
                         "stock PyTorch-ROCm does not have — the reference then falls back to its torch IterativeCorrBlock "
                         "(ccmr/corr.py:118-119).  The stock side runs exactly that (the family's `alt_cuda_corr` global set to "
                         "None, as its failed import would), the accelerated side gets this repo's plug-in back")
    ap.add_argument("--n", type=int, default=10); ap.add_argument("--warm", type=int, default=4)
    ap.add_argument("--H", type=int, default=436); ap.add_argument("--W", type=int, default=1024)
    args = ap.parse_args()
    warnings.filterwarnings("ignore")
    from ptlflow_amd import patch
    assert ref_loader.ensure_family(args.family), f"{args.family}: not staged"
    mod = ref_loader.ref_module(f"ptlflow.models.{args.family}.{args.module}")
    dev = torch.device("cuda")
    torch.manual_seed(0)
    x = torch.rand(1, 2, 3, args.H, args.W, device=dev)

    def build(kw):
        torch.manual_seed(1234)
        return getattr(mod, args.cls)(**json.loads(kw)).eval().to(dev)

    plugin_homes = []
    if args.stock_without_plugin:
        for name in ("corr", args.module):
            m = sys.modules.get(f"ptlflow.models.{args.family}.{name}")
            if m is not None and getattr(m, "alt_cuda_corr", None) is not None:
                plugin_homes.append((m, m.alt_cuda_corr))
                m.alt_cuda_corr = None
        assert plugin_homes, "the family did not import the alt_cuda_corr plug-in"
    stock_model = build(args.stock_kw or args.kw)
    ms_stock, f_stock = timed(stock_model, x, n=args.n, warm=args.warm)
    for m, plug in plugin_homes:
        m.alt_cuda_corr = plug
    model = stock_model if args.stock_kw is None else build(args.kw)
    model.load_state_dict(stock_model.state_dict())
    patch.accelerate(model)
    ms_acc, f_acc = timed(model, x, n=args.n)
    wrapped = [a for a in ("update_block", "fnet", "cnet") if type(getattr(model, a, None)).__module__.startswith("ptlflow_amd")]
    d = (f_acc - f_stock).norm(dim=1)
    print(json.dumps({"model": f"{args.family}.{args.cls}", "input": f"1x2x3x{args.H}x{args.W}", "stock_ms": round(ms_stock, 2),
                      "accelerated_ms": round(ms_acc, 2), "speedup": round(ms_stock / ms_acc, 2),
                      "pairs_per_s": [round(1e3 / ms_stock, 1), round(1e3 / ms_acc, 1)],
                      "epe_accelerated_vs_stock": [float(d.mean()), float(d.max())], "flow_max": float(f_stock.abs().max()),
                      "wrapped": wrapped, "corr_hook": hasattr(mod, patch._ORIG),
                      "upsample_seam": getattr(model.__dict__.get("upsample_flow"), "ok", None),
                      "stock_side": "reference's IterativeCorrBlock fallback (no alt_cuda_corr extension)" if plugin_homes else "as constructed",
                      "forwards_timed": args.n}))


if __name__ == "__main__":
    main()
