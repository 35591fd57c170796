#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 kernel trace: usage gap_summary.py <rocprof_out_dir> [skip_first_ms]"""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# keep the steady-state tail: last 60 % of the kernels
rows = rows[int(len(rows) * 0.4):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [max(0, rows[i][0] - rows[i - 1][1]) for i in range(1, len(rows))]
print(f"kernels {len(rows)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %)  idle {sum(gaps)/1e6:.3f} ms")
big = sorted(((g, rows[i][2][:60], rows[i + 1][2][:60]) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, a, b in big:
    print(f"  gap {g/1e3:8.1f} us  after {a}  before {b}")
