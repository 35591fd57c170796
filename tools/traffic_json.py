#!/usr/bin/env python3
"""profiles/r2_traffic.json from the rocprofv3 PMC summaries (tools/prof_summary.py output of a FETCH_SIZE pass and a
WRITE_SIZE pass over `bench.py` at C2): HBM-side bytes per launch of the dominant kernel, stamped with the digest of
csrc/ so that bench.py reports `roofline.traffic` only for the kernel sources that were actually profiled.

    tools/traffic_json.py <fetch_summary.txt> <write_summary.txt> <kernel-name-substring> <entry-key> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 it reports half
the bytes of wide coalesced reads)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counter(path, kernel, name):
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        if kernel in l and i + 1 < len(lines) and name in lines[i + 1]:
            m = re.search(name + r"=([0-9.e+\-]+)", lines[i + 1])
            if m:
                return float(m.group(1))
    raise SystemExit(f"{name} of {kernel} not found in {path}")


def main(fetch_txt, write_txt, kernel, key, out):
    import bench
    f = counter(fetch_txt, kernel, "FETCH_SIZE") * 1024.0
    w = counter(write_txt, kernel, "WRITE_SIZE") * 1024.0
    rec = {"csrc_digest": bench.csrc_digest(), "entries": {}}
    if os.path.exists(out):
        old = json.load(open(out))
        if old.get("csrc_digest") == rec["csrc_digest"]:
            rec["entries"] = old.get("entries", {})
    rec["entries"][key] = {
        "kernel": kernel, "fetch_size_bytes": f, "write_size_bytes": w, "bytes_per_launch": 2.0 * f + w,
        "source": f"profiles/{os.path.basename(fetch_txt)} + profiles/{os.path.basename(write_txt)}: FETCH_SIZE x2 (gfx950 "
                  f"correction) = {2 * f / 1e9:.3f} GB read + WRITE_SIZE = {w / 1e9:.3f} GB written per launch"}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec["entries"][key]))


if __name__ == "__main__":
    main(*sys.argv[1:6])
