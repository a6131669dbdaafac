#!/usr/bin/env python3
"""rocprofv3 PMC passes (…_counter_collection.csv) of `python bench.py ...` -> per update-block convolution averages
(profiles/pmc_traffic.json entries).

    pmc_extract.py --fetch DIR --write DIR [--sq DIR] --batch 8 [--out profiles/pmc_traffic.json] [--tag ""]

The update block's convolutions share three kernel instantiations with the encoders, so launches are identified by their
POSITION in the iteration: after every `lookup_kernel` dispatch the implicit-GEMM launches come in the fixed order
c1, c2, f2, cv, zr1, q1, zr2, q2, fm, mk (ptlflow_amd/update.py: motion_and_gru + heads; conv_cin2 / flow_delta are other kernels)."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

SEQ = ["c1", "c2", "f2", "cv", "zr1", "q1", "zr2", "q2", "fm", "mk"]


def load(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    idx, last = None, None
    # round 5: with the fused mask conv2 + softmax + upsampling kernel (K13) an iteration has nine implicit-GEMM launches, not ten
    fused = any("mask_upsample" in r["Kernel_Name"] for r in rows)      # K13 (fp32) or K13b (`mask_upsample_b16_kernel`)
    seq = [k for k in SEQ if k != "mk"] if fused else SEQ
    for r in rows:
        name, did = r["Kernel_Name"], r["Dispatch_Id"]
        if did != last:                       # several counters per dispatch: advance the position once per dispatch
            last = did
            if "lookup_kernel" in name:
                idx = 0
                cur = "lookup"
            elif "mask_upsample" in name:
                cur = "mku"
            elif "conv_gemm" in name and idx is not None and idx < len(seq):
                cur = seq[idx]
                idx += 1
            else:
                cur = None
        if cur:
            a = acc[cur][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a2 = acc[cur]["_dur_ns"]
            a2[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a2[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--sq", default="")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out", default="")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    fetch, write = load(a.fetch), load(a.write)
    sq = load(a.sq) if a.sq else {}
    entries = {}
    for key in SEQ + ["lookup", "mku"]:
        if key not in fetch:
            continue
        e = {"fetch_kb": round(fetch[key].get("FETCH_SIZE", 0.0)), "write_kb": round(write.get(key, {}).get("WRITE_SIZE", 0.0)),
             "avg_us": round(fetch[key]["_dur_ns"] / 1e3, 1)}
        s = sq.get(key)
        if s and s.get("SQ_WAVE_CYCLES"):
            e["sq_wait_any_frac"] = round(s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"], 3)
            e["sq_active_inst_frac"] = round(s.get("SQ_ACTIVE_INST_ANY", 0.0) / s["SQ_WAVE_CYCLES"], 3)
            if s.get("SQ_BUSY_CYCLES"):
                e["mfma_busy_per_sq_busy"] = round(s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / s["SQ_BUSY_CYCLES"], 3)
        entries[f"{key}{a.tag}@b{a.batch}"] = e
    print(json.dumps(entries, indent=1))
    if a.out:
        doc = json.load(open(a.out)) if os.path.exists(a.out) else {"entries": {}}
        doc.setdefault("entries", {}).update(entries)
        doc["note_r02"] = ("round-2 entries (keys zr1/zr2/q1/q2/lookup, and every key re-measured after the LDS-transposed epilogue): separate "
                           "--pmc passes of `python bench.py --steps 1 --warmup 1 ...` (scripts/gpu_final2.sh), launches identified by their "
                           "position after each lookup_kernel dispatch; FETCH_SIZE is as reported (KB): consumers apply the x2 gfx950 "
                           "correction of MI355X_MICROARCH.md for 16-byte/lane coalesced reads")
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
