#!/usr/bin/env python3
"""Static check of the relaxed waits in the register-resident forward kernel (csrc/mlp_fwd3.hip, ring_sync<N>):

    python tools/check_vmcnt.py [listing.s]      (without an argument: compiles csrc/mlp_fwd3.hip with hipcc -S)

`s_waitcnt vmcnt(N)` with N > 0 in front of a barrier is only correct if the N youngest vector-memory operations are
all YOUNGER than the last LDS-DMA instruction of the image the barrier publishes (vmcnt retires in issue order).  The kernel
is straight-line, so the listing order is the issue order: for every such wait, count the unconditional 128-bit buffer
stores between the last `buffer_load ... lds` and the wait and require count >= N."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def listing():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    out = os.path.join(tempfile.mkdtemp(), "mlp_fwd3.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm",
                           "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "object-intrinsics_amd", "csrc", "mlp_fwd3.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def check(text):
    found, kernels = 0, 0
    for name, body in re.findall(r"^(_Z\w*sdf_mlp_full3_kernel\w*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        kernels += 1
        stores_since_dma, seen_dma = 0, False
        lines = [l.strip() for l in body.split("\n")]
        for i, l in enumerate(lines):
            if l.startswith("buffer_load") and l.endswith("lds"):
                stores_since_dma, seen_dma = 0, True
            elif l.startswith("buffer_store_dwordx4"):
                stores_since_dma += 1
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)$", l)
            # (only the kernel's own waits, i.e. inline asm: hipcc adds redundant waits of its own right behind them)
            if m and int(m.group(1)) > 0 and lines[i - 1] == ";;#ASMSTART" and any(x.startswith("s_barrier") for x in lines[i + 1:i + 5]):
                n = int(m.group(1))
                found += 1
                assert seen_dma and stores_since_dma >= n, (
                    f"{name}: s_waitcnt vmcnt({n}) in front of a barrier with only {stores_since_dma} stores behind the last LDS-DMA")
    assert kernels >= 1, "kernel not found in the listing"
    return kernels, found


if __name__ == "__main__":
    k, f = check(listing())
    print(f"ok: {f} relaxed waits in {k} kernel variant(s), every one has enough stores behind the image's last LDS-DMA")
