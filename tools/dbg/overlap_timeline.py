#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of training iterations: the launches between the last two prep_render_kernel pairs, with queue
id and how long each launch ran while ANOTHER launch was running (concurrency between streams).
usage: overlap_timeline.py <rocprof_out_dir> [n_rows]"""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[3].startswith("prep_render_kernel")]
a, b = idx[-4], idx[-1]
seg = rows[a:b]
t0 = seg[0][0]
tot_overlap = 0
for i, (s, e, q, k) in enumerate(seg):
    ov = 0
    for j, (s2, e2, q2, k2) in enumerate(seg):
        if j != i and q2 != q:
            ov += max(0, min(e, e2) - max(s, s2))
    tot_overlap += ov
    if len(sys.argv) > 2:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} ov {ov / 1e3:7.2f} q{q} {k}")
queues = sorted({r[2] for r in seg})
print(f"# {len(seg)} launches over {(seg[-1][1] - t0) / 1e3:.1f} us on queues {queues}; cross-queue overlap {tot_overlap / 2e3:.1f} us")
for q in queues:
    print(f"# queue {q}: {sum(1 for r in seg if r[2] == q)} launches, busy {sum(r[1] - r[0] for r in seg if r[2] == q) / 1e3:.1f} us")
