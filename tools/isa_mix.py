"""Static instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/isa_mix.py /tmp/isa/mlp_bwd.s [substring of the mangled kernel name ...]

The MLP kernels are straight-line (fully unrolled, no loops inside a tile), so the static count IS the per-tile
dynamic count; for kernels with loops the count is per loop body and only good for A/B comparisons.
Classes: mfma, ds_read, ds_write/atomic, vmem (buffer/global), trans (v_sin/cos/rcp/exp/...), pk (v_pk_*),
cvt, accvgpr moves, dpp (any VALU with a dpp modifier), other VALU, salu, waitcnt, nop.
With --loops N the basic blocks LLVM annotates as belonging to a loop ("in Loop: Header=BBx" / "This Inner Loop Header")
are weighted N times when that loop holds at least 48 MFMAs (the layer loops of csrc/mlp_bwd.hip run 7 times per tile;
small wave-level loops -- waterfall / CAS -- count once), and a "[dynamic per tile]" mix is printed next to the static one."""
import collections
import re
import sys

TRANS = ("v_sin_", "v_cos_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_")


def classify(op, line):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_cvt") or op.startswith("v_fma_mix"):
        return "cvt/mix"
    if op.startswith("v_"):
        if "dpp" in line or "quad_perm" in line or "row_" in line:
            return "dpp"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith("v_fract"):
            return "fract"
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    pats = [a for a in sys.argv[2:] if not a.startswith("-")]
    trips = int(sys.argv[sys.argv.index("--loops") + 1]) if "--loops" in sys.argv else 0
    pats = [a for a in pats if not a.isdigit()]
    cur = None
    blocks = {}
    mixes = collections.OrderedDict()
    ops = collections.defaultdict(collections.Counter)
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith("."):
            name = m.group(1)
            cur = name if (not pats or any(p in name for p in pats)) else None
            if cur is not None:
                mixes.setdefault(cur, collections.Counter())
            continue
        if cur is None:
            continue
        s = line.strip()
        lm = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", s)
        if lm:
            hm = re.search(r"Header=(BB\d+_\d+)", s)
            loop = hm.group(1) if hm else (s.split(":")[0][1:] if "Loop Header" in s else None)
            blocks.setdefault(cur, []).append([loop, collections.Counter()])
            continue
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") or s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        c = classify(op, s)
        mixes[cur][c] += 1
        ops[cur][op] += 1
        if cur not in blocks:
            blocks[cur] = [[None, collections.Counter()]]
        blocks[cur][-1][1][c] += 1
    if trips:
        for k in list(mixes):
            per_loop = collections.Counter()
            for loop, mix in blocks.get(k, []):
                if loop:
                    per_loop[loop] += mix["mfma"]
            dyn = collections.Counter()
            for loop, mix in blocks.get(k, []):
                w = trips if loop and per_loop[loop] >= 48 else 1
                for c, n in mix.items():
                    dyn[c] += n * w
            mixes[k + " [dynamic per tile, layer loops x%d]" % trips] = dyn
    for k, mix in mixes.items():
        tot = sum(mix.values())
        valu_like = sum(mix[c] for c in ("valu", "pk", "cvt/mix", "dpp", "cndmask", "trans", "accvgpr", "fract"))
        print(f"== {k}\n   total {tot}  VALU-class {valu_like}  per-mfma {valu_like / max(1, mix['mfma']):.2f}")
        print("   " + "  ".join(f"{c}={n}" for c, n in sorted(mix.items(), key=lambda x: -x[1])))
        if "-v" in sys.argv:
            print("   top ops: " + "  ".join(f"{o}={n}" for o, n in ops[k].most_common(40)))


if __name__ == "__main__":
    main()
