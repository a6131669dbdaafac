#!/usr/bin/env python3
"""Sum rocprofv3 PMC counters per kernel name over one or more pass directories:  pmc_by_kernel.py DIR [DIR ...] [--match lookup,pool,copy]"""
import csv, glob, os, sys
from collections import defaultdict
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = [m for a in sys.argv[1:] if a.startswith("--match=") for m in a[8:].split(",")]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if match and not any(m in n for m in match):
                continue
            a = acc[n[:110]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            a = acc[n[:110]]["_dur_us"]
            a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[1] += 1
for n, cs in acc.items():
    print(n)
    for c, (v, k) in sorted(cs.items()):
        print(f"    {c:28s} avg {v / k:16.1f}  (n={k})")
