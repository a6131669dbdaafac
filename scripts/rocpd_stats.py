#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / share, plus the
GPU-busy vs wall split of the traced window.  Usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info('kernels')")]
    rows = cur.execute("select * from kernels").fetchall()
    ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
    agg = {}
    t0, t1 = min(r[si] for r in rows), max(r[ei] for r in rows)
    busy = 0
    for r in rows:
        d = r[ei] - r[si]
        a = agg.setdefault(r[ni], [0, 0])
        a[0] += 1
        a[1] += d
        busy += d
    lines = [f"kernel dispatches: {len(rows)}; traced window {1e-6*(t1-t0):.2f} ms; sum of kernel time {1e-6*busy:.2f} ms "
             f"({100.0*busy/(t1-t0):.1f}% of window)", "",
             "| kernel | calls | total ms | avg us | % of kernel time |", "|---|---:|---:|---:|---:|"]
    for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        lines.append(f"| `{short(name)}` | {n} | {tot/1e6:.3f} | {tot/n/1e3:.2f} | {100.0*tot/busy:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
