#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / counter collection) into small per-kernel
summaries that can be committed under profiles/.   usage: prof_summary.py <rocprof_out_dir> <out.txt>"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(d, out):
    lines = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(list)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        tot = sum(sum(v) for v in agg.values())
        lines.append(f"# kernel trace: {os.path.relpath(f, d)}  (total kernel time {tot/1e6:.3f} ms)")
        lines.append(f"{'calls':>7} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10} {'pct':>6}  kernel")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"{len(v):7d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {sum(v)/1e6:10.3f} "
                         f"{100*sum(v)/tot:6.2f}  {k}")
        lines.append("")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        lines.append(f"# counters (mean per dispatch): {os.path.relpath(f, d)}")
        for k, cs in sorted(agg.items()):
            desc = "  ".join(f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items()))
            lines.append(f"{k}\n    {desc}")
        lines.append("")
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
