#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --output-format csv` run (…_kernel_trace.csv, optionally the …_counter_collection.csv of a
--pmc pass) as a markdown table: per kernel calls / total / average duration / share, per-forward time, registers and LDS.

    trace_stats.py <dir-or-csv> [--forwards N] [--title T] [--out file.md] [--pmc counter_collection.csv ...]
"""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"pfkg::", "", name)
    return name if len(name) <= 120 else name[:117] + "..."


def find(path, pat):
    if os.path.isfile(path):
        return path
    c = sorted(glob.glob(os.path.join(path, "**", pat), recursive=True))
    if not c:
        raise SystemExit(f"no {pat} under {path}")
    return c[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--forwards", type=int, default=0)
    ap.add_argument("--title", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--top", type=int, default=28)
    ap.add_argument("--pmc", nargs="*", default=[])
    a = ap.parse_args()
    rows = list(csv.DictReader(open(find(a.path, "*kernel_trace.csv"))))
    agg = defaultdict(lambda: [0, 0, 0, 0, 0])
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    busy = 0
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e = agg[r["Kernel_Name"]]
        e[0] += 1
        e[1] += d
        e[2] = max(e[2], int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count") or 0))
        e[3] = max(e[3], int(r["LDS_Block_Size"]))
        e[4] = max(e[4], int(r["Scratch_Size"]))
        busy += d
    pmc = defaultdict(lambda: defaultdict(float))
    pmc_n = defaultdict(lambda: defaultdict(int))
    for f in a.pmc:
        for r in csv.DictReader(open(find(f, "*counter_collection.csv"))):
            pmc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            pmc_n[r["Kernel_Name"]][r["Counter_Name"]] += 1
    counters = sorted({c for k in pmc for c in pmc[k]})
    out = []
    if a.title:
        out += [f"### {a.title}", ""]
    line = (f"kernel dispatches: {len(rows)}; traced window {1e-6 * (t1 - t0):.1f} ms; sum of kernel time {1e-6 * busy:.1f} ms "
            f"({100.0 * busy / (t1 - t0):.1f}% of the window)")
    if a.forwards:
        line += f"; {a.forwards} forwards => {1e-6 * busy / a.forwards:.2f} ms of kernels per forward"
    out += [line, ""]
    hdr = "| kernel | calls | total ms | avg us | " + ("ms per forward | " if a.forwards else "") + "% | regs | LDS B | scratch |"
    hdr += "".join(f" {c} / call |" for c in counters)
    out += [hdr, "|---|---:|---:|---:|" + ("---:|" if a.forwards else "") + "---:|---:|---:|---:|" + "---:|" * len(counters)]
    for name, (n, tot, vg, lds, scr) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
        row = f"| `{short(name)}` | {n} | {tot / 1e6:.2f} | {tot / n / 1e3:.1f} | "
        if a.forwards:
            row += f"{tot / 1e6 / a.forwards:.2f} | "
        row += f"{100.0 * tot / busy:.1f} | {vg} | {lds} | {scr} |"
        for c in counters:
            row += f" {pmc[name][c] / pmc_n[name][c]:.4g} |" if pmc_n[name].get(c) else " |"
        out.append(row)
    text = "\n".join(out)
    print(text)
    if a.out:
        with open(a.out, "a") as f:
            f.write(text + "\n\n")


if __name__ == "__main__":
    main()
