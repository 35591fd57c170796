"""Builds liboi_hip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python object-intrinsics_amd/build.py [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the repo
snapshot to the GPU box.  No torch headers are involved: the library has a plain C ABI
(include/oi_hip.h)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "oi_amd")
LIB = os.path.join(OUT_DIR, "liboi_hip.so")
STAMP = os.path.join(OUT_DIR, ".liboi_hip.stamp")
SOURCES = ["mlp.hip", "mlp_fwd3.hip", "mlp_fwd3b.hip", "color_head.hip", "render.hip", "disc.hip", "disc_small.hip", "disc_large.hip", "mlp_bwd.hip", "render_bwd.hip", "disc_bwd.hip", "optim.hip", "loss.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + os.environ.get("OI_FLAGS", "").split()
# per-file extra flags.  mlp_fwd3.hip pins 256 parked values to the AGPR half of the register file; hipcc's default
# (AGPR-form MFMA accumulators) would need 32 more AGPRs than exist, the VGPR form keeps the accumulators with the VALU.
EXTRA = {"mlp_fwd3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
         # (224 AGPRs hold the parked cosines; without the SLP vectoriser hipcc selects v_fma_mix_f32 for acc * fp16 half)
         "mlp_fwd3b.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"],
         # backward sweep + weight-gradient GEMM: same choice -- no private-memory scratch left (76 spilled dwords before),
         # sweep -1.5 %, weight-gradient GEMM -5 % on the same box
         "mlp_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "oi_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode())  # content + name, not the checkout's location
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(EXTRA).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in _sources():
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *EXTRA.get(s, []), "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
