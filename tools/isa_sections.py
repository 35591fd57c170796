"""Per-section instruction counts of one kernel in a gfx950 listing: sections are the `; OI_MARK <name>` comments the
kernels emit (csrc/mlp_bwd.hip).  Shows where the register spills (scratch_load / scratch_store) sit.

    python tools/isa_sections.py /tmp/isa/mlp_bwd.s sweep_kernelILi4ELb0"""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    inside, sec = False, "(prologue)"
    cnt = collections.OrderedDict()
    for line in open(path):
        s = line.strip()
        if not inside:
            if s.endswith(":") or ": ;" in s or s.split(";")[0].strip().endswith(":"):
                name = s.split(":")[0]
                if pat in name and not name.startswith("."):
                    inside, sec = True, "(prologue)"
            continue
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
            if s.startswith(".Lfunc_end"):
                break
        m = re.search(r"; OI_MARK (.*)", s)
        if m:
            sec = m.group(1).strip()
            continue
        op = s.split()[0] if s else ""
        d = cnt.setdefault(sec, collections.Counter())
        for key, pre in (("spill_ld", "scratch_load"), ("spill_st", "scratch_store"), ("mfma", "v_mfma"), ("ds_read", "ds_read"),
                         ("vmem", "buffer_"), ("accvgpr", "v_accvgpr"), ("waitcnt", "s_waitcnt"), ("nop", "s_nop")):
            if op.startswith(pre):
                d[key] += 1
                break
        else:
            if op.startswith("v_"):
                d["valu"] += 1
    for k, v in cnt.items():
        print(f"{k:16s}", "  ".join(f"{a}={b}" for a, b in sorted(v.items())))


main()
