#!/usr/bin/env python3
"""GPU idle time inside steady-state training iterations from a rocprofv3 kernel trace: the window between two
`mlp_bwd_sweep_kernel` launches N iterations apart; busy = union of kernel intervals.
usage: train_gaps.py <trace_dir> [first_sweep_index] [n_iterations]"""
import csv, glob, os, sys
from collections import defaultdict


def main(d, i0=6, n=8):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                             r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]))
    rows.sort()
    sw = [k for k, r in enumerate(rows) if "mlp_bwd_sweep" in r[2]]
    a, b = sw[i0], sw[i0 + n]
    win = rows[a:b]
    t0, t1 = win[0][0], rows[b][0]
    busy, cur_end, gaps = 0, t0, []
    for k, (s, e, nm) in enumerate(win):
        if s > cur_end:
            gaps.append((s - cur_end, win[k - 1][2] if k else "", nm))
            cur_end = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    tot = t1 - t0
    print(f"{n} iterations: {tot / n / 1e6:.3f} ms per iteration, GPU busy {busy / n / 1e6:.3f} ms ({100 * busy / tot:.1f} %), idle {(tot - busy) / n / 1e6:.3f} ms, "
          f"{len(win) / n:.0f} dispatches per iteration")
    agg = defaultdict(lambda: [0, 0])
    for g, p, q in gaps:
        agg[(p, q)][0] += g
        agg[(p, q)][1] += 1
    print("largest idle gaps (sum over the window, ms / count / after kernel -> before kernel):")
    for (p, q), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {g / 1e6:8.3f} {c:5d}  {p} -> {q}")
    # one iteration, launch by launch (runs of the same kernel folded): start offset us, count, kernel us, idle-before us
    a1 = sw[i0 + 1]
    one = rows[a:a1]
    print(f"\none iteration: {len(one)} dispatches")
    k = 0
    while k < len(one):
        j, busy_us, idle_us = k, 0.0, 0.0
        while j < len(one) and one[j][2] == one[k][2]:
            busy_us += (one[j][1] - one[j][0]) / 1e3
            if j:
                idle_us += max(0, one[j][0] - max(r[1] for r in one[max(0, j - 4):j])) / 1e3
            j += 1
        print(f"  {(one[k][0] - one[0][0]) / 1e3:9.1f}  x{j - k:<3d} {busy_us:8.1f}  idle {idle_us:7.1f}  {one[k][2]}")
        k = j


if __name__ == "__main__":
    main(sys.argv[1], *(int(x) for x in sys.argv[2:]))
