#!/usr/bin/env python3
"""Kernel launches and kernel time per training iteration from two rocprofv3 kernel traces of the same command with
different iteration counts (the difference removes warm-up, graph capture and the forward-only bench legs).
usage: train_launches.py <trace_dir_a> <iters_a> <trace_dir_b> <iters_b>"""
import csv
import glob
import os
import sys
from collections import Counter


def load(d):
    calls, busy = Counter(), Counter()
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
                calls[k] += 1
                busy[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return calls, busy


def main(da, na, db, nb):
    ca, ba = load(da)
    cb, bb = load(db)
    dn = nb - na
    tot_calls = (sum(cb.values()) - sum(ca.values())) / dn
    tot_busy = (sum(bb.values()) - sum(ba.values())) / dn / 1e6
    print(f"# per training iteration (Trainer.train_step at the bench's C2 configuration), from traces of {na} and {nb} iterations")
    print(f"# kernel dispatches per iteration: {tot_calls:.1f} (hipGraph-replayed kernels count individually)   kernel time per iteration: {tot_busy:.3f} ms")
    print(f"{'calls/it':>9} {'ms/it':>9}  kernel")
    rows = []
    for k in cb:
        dc = (cb[k] - ca.get(k, 0)) / dn
        dt = (bb[k] - ba.get(k, 0)) / dn / 1e6
        if dc > 0.01:
            rows.append((dt, dc, k))
    for dt, dc, k in sorted(rows, reverse=True)[:40]:
        print(f"{dc:9.2f} {dt:9.4f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]))
