#!/usr/bin/env python3
"""Timeline of ONE steady-state bench step from a rocprofv3 kernel trace: every launch with its start offset,
duration and the idle gap before it.   usage: timeline.py <rocprof_out_dir> <out.txt> [anchor-kernel] [step-index]"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:90]


def main(d, out, anchor="gen_rays_kernel", which=-3):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if r[2].startswith(anchor)]
    a, b = idx[which], idx[which + 1]
    step = rows[a:b + 1]
    t0 = step[0][0]
    lines = [f"# one step: {len(step) - 1} launches, {(step[-1][0] - t0) / 1e3:.1f} us from {anchor} to the next {anchor}",
             f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel"]
    prev_end = None
    busy = gap_tot = 0.0
    for s, e, k in step[:-1]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        lines.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap:7.2f}  {k}")
        busy += (e - s) / 1e3
        gap_tot += max(gap, 0.0)
        prev_end = e if prev_end is None else max(prev_end, e)
    gap_tot += (step[-1][0] - prev_end) / 1e3
    lines.append(f"# busy {busy:.1f} us, idle {gap_tot:.1f} us")
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4]), *(int(x) for x in sys.argv[4:5]))
