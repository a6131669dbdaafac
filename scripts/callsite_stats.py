#!/usr/bin/env python3
"""Per CALL SITE durations of the update block's launches out of a rocprofv3 `--kernel-trace --output-format csv` run of
`python bench.py ...` (and, optionally, per-call-site counter averages out of `--pmc` passes of the same command).

The convolutions of one iteration share a handful of kernel instantiations (and those with the encoders), so a kernel-name
aggregate cannot isolate a launch.  The launch ORDER inside an iteration is fixed (ptlflow_amd/update.py: motion_and_gru + heads;
ptlflow_amd/raft.py: _iterate): after every `lookup_kernel` dispatch the implicit-GEMM launches come as
c1, c2, f2, cv, zr1, q1, zr2, q2, fm, mk — the same positional rule as scripts/pmc_extract.py — and the other kernels of the
iteration (lookup, conv_cin2, flow_delta, convex upsampling) have names of their own.  The side stream's two launches (mk, the
upsampling) of iteration i are dispatched after iteration i+1's lookup may already have been: position is counted per QUEUE
(stream), so the rule survives the overlap.

    callsite_stats.py <trace dir> [--forwards N] [--gflop-json bench_line.json] [--pmc DIR ...] [--title T] [--out file.md]
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

SEQ = ["c1", "c2", "f2", "cv", "zr1", "q1", "zr2", "q2", "fm", "mk"]
NAMED = {"lookup_kernel": "lookup", "conv_cin2": "convf1 (7x7)", "flow_delta": "flow_delta", "convex_upsample": "upsample",
         "mask_upsample": "mask_upsample (fused)",        # K13 (fp32) and K13b (`mask_upsample_b16_kernel`)
         "conv_gemm_v3_group_kernel": "c1+f2[+mk] (grouped)"}   # round 6, small batches: convc1 | convf2 | previous mask conv2 in one grid
GROUPED_SEQ = ["c2", "cv", "zr1", "q1", "zr2", "q2", "fm", "mk"]   # what follows a grouped launch (mk: the last iteration's only)


def find(path, pat):
    c = sorted(glob.glob(os.path.join(path, "**", pat), recursive=True))
    if not c:
        raise SystemExit(f"no {pat} under {path}")
    return c[0]


def classify(rows, fused=None):
    """rows of one trace (dicts with Kernel_Name, Dispatch_Id, Queue_Id) -> list of (callsite, row) in dispatch order."""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    idx = None
    out = []
    # with the fused kernel K13 an iteration has no separate mask conv2: nine implicit-GEMM launches per lookup, not ten
    if fused is None:
        fused = any("mask_upsample" in r["Kernel_Name"] for r in rows)
    SEQ = [k for k in globals()["SEQ"] if k != "mk"] if fused else globals()["SEQ"]
    BASE_SEQ = SEQ
    for r in rows:
        name = r["Kernel_Name"]
        site = None
        for k, v in NAMED.items():
            if k in name:
                site = v
                break
        if site == "lookup":
            idx, SEQ = 0, BASE_SEQ
        elif site == NAMED["conv_gemm_v3_group_kernel"]:
            idx, SEQ = 0, GROUPED_SEQ
        elif site is None and "conv_gemm" in name and idx is not None:
            # mk runs on the side stream when the mask branch overlaps: it is the only conv_gemm launch of that queue
            if idx < len(SEQ):
                site = SEQ[idx]
                idx += 1
        out.append((site, r))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--forwards", type=int, default=0)
    ap.add_argument("--gflop-json", default="")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--title", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(find(a.path, "*kernel_trace.csv"))))
    # the side stream reorders DISPATCH ids between queues; classify per queue so that mk (side) does not steal a main-queue slot
    queues = defaultdict(list)
    for r in rows:
        queues[r.get("Queue_Id", "0")].append(r)
    main_q = max(queues, key=lambda q: sum("lookup_kernel" in r["Kernel_Name"] for r in queues[q]))
    fused = any("mask_upsample" in r["Kernel_Name"] for r in rows)
    # the mask branch on a side queue (batch 8): mask conv2 is that queue's launch, the main queue has nine GEMM launches per lookup
    side_mk = any(q != main_q and any("convex_upsample" in r["Kernel_Name"] for r in rs) for q, rs in queues.items())
    acc = defaultdict(lambda: [0, 0])
    for q, rs in queues.items():
        if q == main_q:
            for site, r in classify(rs, fused or side_mk):
                if site:
                    e = acc[site]; e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        else:
            # a side queue: the mask branch's (mask conv2 + upsampling of iteration i next to iteration i+1) is the one that hosts the
            # upsampling launches; the encoders' side queue (cnet next to fnet at small batches) has convolutions of its own
            # (with the fused kernel K13 the branch has no separate mask conv2: only a queue that hosts the un-fused upsampling
            #  kernel can host `mk` launches)
            mask_q = any("convex_upsample" in r["Kernel_Name"] for r in rs)
            for r in rs:
                name = r["Kernel_Name"]
                site = next((v for k, v in NAMED.items() if k in name), "mk" if (mask_q and "conv_gemm" in name) else None)
                if site:
                    e = acc[site]; e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    # a serial run (no side stream) has mk inside the main queue's sequence already
    gf = {}
    if a.gflop_json:
        d = json.load(open(a.gflop_json))
        gf = {k: v.get("gflop") for k, v in d.get("kernels", {}).items() if isinstance(v, dict)}
    pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for pdir in a.pmc:
        prow = list(csv.DictReader(open(find(pdir, "*counter_collection.csv"))))
        seen = {}
        per_q = defaultdict(list)
        for r in prow:
            if r["Dispatch_Id"] not in seen:
                seen[r["Dispatch_Id"]] = None
                per_q["0"].append(r)      # PMC passes serialise dispatches: one queue order is the program order
        sites = {r["Dispatch_Id"]: site for site, r in classify(per_q["0"])}
        for r in prow:
            site = sites.get(r["Dispatch_Id"])
            if site:
                e = pmc[site][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
    lines = []
    if a.title:
        lines.append(f"### {a.title}\n")
    counters = sorted({c for s in pmc for c in pmc[s]})
    hdr = "| call site | launches | avg us |" + ("".join(f" {c} (avg) |" for c in counters))
    lines.append(hdr)
    lines.append("|---|---:|---:|" + "---:|" * len(counters))
    order = ["lookup", NAMED["conv_gemm_v3_group_kernel"]] + SEQ + ["convf1 (7x7)", "flow_delta", "upsample", "mask_upsample (fused)"]
    for site in order:
        if site not in acc:
            continue
        n, ns = acc[site]
        row = f"| {site} | {n} | {ns / n / 1e3:.1f} |"
        for c in counters:
            v = pmc[site].get(c)
            row += f" {v[0] / v[1]:.0f} |" if v and v[1] else " |"
        lines.append(row)
    text = "\n".join(lines) + "\n"
    print(text)
    if a.out:
        open(a.out, "a").write(text + "\n")


if __name__ == "__main__":
    main()
