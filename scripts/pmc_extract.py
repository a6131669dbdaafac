#!/usr/bin/env python3
"""Per (kernel, grid) averages of rocprofv3 PMC passes (…_counter_collection.csv) -> profiles/pmc_traffic.json entries.

    pmc_extract.py --fetch DIR --write DIR [--sq DIR] --batch 8 --out profiles/pmc_traffic.json

Conv launches of one forward share a few kernel instantiations; the update block's convolutions are told apart by
(epilogue template argument, grid size): at M = B*7040 pixels and 64x64 tiles, tiles_n = ceil(cout / 64)."""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def load(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], int(r["Grid_Size"]))
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--sq", default="")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    fetch, write = load(a.fetch), load(a.write)
    sq = load(a.sq) if a.sq else {}
    tiles_m = (a.batch * 7040 + 63) // 64
    want = {"fm": (0, 8), "c2": (0, 3), "cv": (0, 2), "zr": (1, 4), "q": (2, 2), "mk": (0, 9), "c1": (0, 4), "f2": (0, 1)}
    entries = {}
    for key, (epi, tn) in want.items():
        grid = tiles_m * tn * 256
        for (name, g), c in fetch.items():
            m = re.search(r"conv_gemm(?:_v3)?_kernel<64, 64, 32, 32, (\d)", name)
            if not m or int(m.group(1)) != epi or g != grid:
                continue
            e = {"fetch_kb": round(c.get("FETCH_SIZE", 0.0)), "write_kb": round(write.get((name, g), {}).get("WRITE_SIZE", 0.0)), "kernel": name[:60]}
            s = sq.get((name, g))
            if s and s.get("SQ_BUSY_CYCLES"):
                # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over SEs like SQ_BUSY_CYCLES; 4 SIMDs per CU share ... report the raw ratio
                e["mfma_busy_over_grbm"] = round(s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(s.get("GRBM_GUI_ACTIVE", 1.0), 1.0), 4)
                e["sq_wait_any_frac"] = round(s.get("SQ_WAIT_ANY", 0.0) / max(s.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)
            entries[f"{key}@b{a.batch}"] = e
            break
    print(json.dumps(entries, indent=1))
    if a.out:
        doc = json.load(open(a.out)) if os.path.exists(a.out) else {"entries": {}}
        doc.setdefault("entries", {}).update(entries)
        doc["note_r02"] = ("round-2 entries re-measured after the LDS-transposed epilogue (scripts/gpu_final2.sh: separate --pmc passes "
                           "of `python bench.py --steps 1 --warmup 1 ...`); FETCH_SIZE x2 is the gfx950 correction of MI355X_MICROARCH.md")
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
