#!/usr/bin/env python3
"""Index of one round's evidence in profiles/, GENERATED from the files (every number in the table is read out of the file it
describes -- the round-5 index was prose and contradicted the files it indexed).

    python tools/profiles_index.py r6            # prints the markdown table
    python tools/profiles_index.py r6 --write    # replaces the block between the r6 markers in profiles/README.md (appends it if absent)
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def kernel_rows(path):
    """[(calls, avg_us, min_us, name)] of a tools/prof_summary.py file."""
    rows = []
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if m:
            rows.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), m.group(7).strip()))
    return rows


def find(rows, needle):
    for r in rows:
        if needle in r[3]:
            return r
    return None


def k_us(rows, needle):
    r = find(rows, needle)
    return "n/a" if r is None else f"{r[1]:.1f} us avg / {r[2]:.1f} min over {r[0]} launches"


def bench_line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def describe(tag):
    out = []
    f = lambda name: os.path.join(P, f"{tag}_{name}")
    ex = os.path.exists

    if ex(f("bench_f16x3.json")):
        d = bench_line(f("bench_f16x3.json"))
        r, t, b = d["roofline"], d.get("training") or {}, d.get("bf16_mode") or {}
        s = (f"`python bench.py`, un-profiled: **{d['value'] / 1e6:.3f} M rays/s, {d['ms_per_step']:.4f} ms/step**; dominant kernel "
             f"{r['kernel'].split(' ')[0]} {r['kernel_ms']:.4f} ms by HIP events ({r['kernel_ms_samples']} pairs) = {r['achieved']:.1f} TFLOP/s, "
             f"`frac` {r['frac']:.4f} (executed MFMA {r.get('executed_mfma_frac_of_peak', 0):.3f}); traffic "
             f"{(r['traffic'] or 0) / 1e9:.3f} GB per launch; D forward {d.get('d_images_per_s', 0):.0f} images/s through the plan, "
             f"**{d.get('d_images_per_s_eager', 0):.0f} through the module's own `forward`**")
        if t:
            s += (f"; training {t['ms_per_it']:.3f} ms/it (render fwd + bwd {t['render_fwd_bwd']['ms']:.3f}, graphed D step "
                  f"{t['d_step']['ms']:.3f})")
        if b:
            s += (f"; `bf16_mode` {b['value'] / 1e6:.2f} M rays/s, {b['ms_per_step']:.4f} ms/step, kernel {b['kernel_ms']:.4f} ms, frac "
                  f"{b['roofline']['frac']:.3f}")
            if b.get("training"):
                s += f", training {b['training']['ms_per_it']:.3f} ms/it"
        e = d.get("extras") or {}
        if e.get("training_shipped_config"):
            s += f"; shipped configuration {e['training_shipped_config']['it_per_s']:.1f} it/s"
        if e.get("inference"):
            s += f"; inference {e['inference']['s_per_frame'] * 1e3:.2f} ms per frame"
        for k in ("B1", "B4", "B64", "B1_128"):
            if (e.get("discriminator") or {}).get(k):
                s += f"; D {k} {e['discriminator'][k]['ms'] * 1e3:.1f} us"
        c = d.get("cpu_baseline") or {}
        if c:
            s += f"; cpu_baseline {c['value']:.0f} rays/s on {c['cores']} threads ({c['kind']})"
        out.append((f"{tag}_bench_f16x3.json", s))
    if ex(f("bench_c4_f16x3.json")):
        d = bench_line(f("bench_c4_f16x3.json"))
        r = d["roofline"]
        out.append((f"{tag}_bench_c4_f16x3.json", f"C4 per GPU (128^2, 128 + 128 samples, 4 up-sampling steps): {d['value'] / 1e6:.3f} M rays/s, "
                    f"{d['ms_per_step']:.3f} ms/step, full MLP kernel {r['kernel_ms']:.3f} ms, frac {r['frac']:.4f}, traffic {(r['traffic'] or 0) / 1e9:.2f} GB"))
    if ex(f("traffic.json")):
        d = json.load(open(f("traffic.json")))
        s = "; ".join(f"{k}: {v['bytes_per_launch'] / 1e9:.3f} GB ({v['fetch_size_bytes'] * 2 / 1e9:.3f} read x2-corrected + {v['write_size_bytes'] / 1e9:.3f} written)"
                      for k, v in d["entries"].items())
        out.append((f"{tag}_traffic.json", f"PMC bytes per launch (FETCH_SIZE x2 + WRITE_SIZE), csrc digest `{d['csrc_digest'][:12]}`: {s}"))
    for name, what, keys in (
            ("kernel_stats_f16x3.txt", "forward steps only (`--train-steps 0`)", ("sdf_mlp_full3_kernel", "sdf_mlp_kernel<4", "film_blob_f3_kernel", "composite_fwd_kernel", "prep_render_kernel")),
            ("kernel_stats_bf16.txt", "`--precision bf16`, forward steps", ("sdf_mlp_full3p_kernel", "film_images_b_kernel", "sdf_mlp_kernel<2")),
            ("kernel_stats_train.txt", "f16x3 training, 30 iterations", ("mlp_bwd_sweep_kernel", "mlp_wgrad_f16_kernel", "sdf_mlp_full3_kernel", "composite_bwd_kernel")),
            ("kernel_stats_train_bf16.txt", "bf16-mode training, 20 iterations", ("mlp_bwd_sweep_kernel", "mlp_wgrad_f16_kernel", "sdf_mlp_full3p_kernel")),
            ("kernel_stats_disc_b64.txt", "ADADiscriminatorView forward at batch 64", ("dl_gemm_kernel", "dl_conv1_kernel", "dl_reduce_kernel", "ada_pad_up2_kernel", "ada_resample_down2_kernel")),
            ("kernel_stats_train_rccl_1rank.txt", "the training job inside a one-rank RCCL process group", ("ncclDevKernel", "mlp_bwd_sweep_kernel"))):
        if ex(f(name)):
            rows = kernel_rows(f(name))
            out.append((f"{tag}_{name}", what + ": " + "; ".join(f"`{k}` {k_us(rows, k)}" for k in keys if find(rows, k))))
    for name in sorted(glob.glob(f("timeline_*.txt"))):
        head = [l.strip("# \n") for l in open(name) if l.startswith("#")][:3]
        out.append((os.path.basename(name), " / ".join(head)))
    for name, kern in (("pmc_sq_f16x3.txt", "sdf_mlp_full3_kernel"), ("pmc_sq_f16x3.txt", "mlp_bwd_sweep_kernel"),
                       ("pmc_sq_f16x3.txt", "mlp_wgrad_f16_kernel"), ("pmc_sq_bf16.txt", "sdf_mlp_full3p_kernel")):
        if ex(f(name)):
            lines = open(f(name)).read().splitlines()
            start = next((i for i, l in enumerate(lines) if l.startswith("# counters")), len(lines))
            vals = next((lines[i + 1].strip() for i in range(start, len(lines) - 1) if lines[i].lstrip().startswith(kern)), None)
            desc = "see file"
            if vals:
                c = {k: float(v) for k, v in re.findall(r"(\w+)=([\d.e+-]+)", vals)}
                desc = re.sub(r" \(n=\d+\)", "", vals)
                if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and c.get("GRBM_GUI_ACTIVE"):
                    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024   # 8 XCDs count the same interval; 256 CUs x 4 SIMDs
                    desc += f" -> MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.1%} of the SIMD cycles"
            out.append((f"{tag}_{name}", "SQ counters (separate --pmc pass), mean per dispatch of `" + kern + "`: " + desc))
    for name in ("pmc_fetch_f16x3.txt", "pmc_write_f16x3.txt", "pmc_fetch_bf16.txt", "pmc_write_bf16.txt", "pmc_fetch_c4.txt", "pmc_write_c4.txt",
                 "pmc_fetch_train_bf16.txt", "pmc_write_train_bf16.txt"):
        if ex(f(name)):
            out.append((f"{tag}_{name}", "the FETCH_SIZE / WRITE_SIZE pass behind the matching entry of " + tag + "_traffic.json"))
    if ex(f("c5_mlp_microbench.jsonl")):
        rows = [json.loads(l) for l in open(f("c5_mlp_microbench.jsonl")) if l.strip().startswith("{")]
        out.append((f"{tag}_c5_mlp_microbench.jsonl", "2^21 points, sdf-only / full pass: " + ", ".join(
            f"{r['mode']} {r['sdf_only']['ms']:.3f} / {r['full']['ms']:.3f} ms" for r in rows if "sdf_only" in r and "full" in r)))
    if ex(f("color_head.txt")):
        out.append((f"{tag}_color_head.txt", open(f("color_head.txt")).read().strip().splitlines()[-1][:240]))
    if ex(f("gradient_margins.txt")):
        rows = []
        for l in open(f("gradient_margins.txt")):
            m = re.match(r"(\S+)\s+(\d+)\s+([\d.e+-]+)\s+([\d.e+-]+)\s+(.*)", l)
            if m and m.group(1) != "case":
                rows.append((m.group(1), float(m.group(3))))
        pick = [r for r in rows if any(k in r[0] for k in ("c2_size", "f6_generator", "f9_g_step_grads", "mlp_backward_vs", "training_trajectory", "generator_fit"))]
        out.append((f"{tag}_gradient_margins.txt", "`tools/grad_margin.py`, worst relative error per case: " + ", ".join(f"{n} {v:.2e}" for n, v in pick)))
    known = {o[0] for o in out}
    for name in sorted(glob.glob(os.path.join(P, f"{tag}_*"))):
        b = os.path.basename(name)
        if b not in known:
            first = open(name, errors="replace").readline().strip("# \n")[:200]
            out.append((b, first))
    return out


def table(tag):
    lines = [f"<!-- BEGIN {tag} (generated by tools/profiles_index.py {tag} --write: do not edit by hand) -->",
             f"## Round {tag[1:]} (`{tag}_` prefix; `gpurun -- 'OI_PROFILE_TAG={tag} bash tools/refresh_profiles.sh'`, index generated from the files)", "",
             "| file | what it holds (numbers read from the file) |", "|---|---|"]
    for name, what in describe(tag):
        lines.append(f"| {name} | {what.replace('|', '/')} |")
    lines.append(f"<!-- END {tag} -->")
    return "\n".join(lines)


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
    txt = table(tag)
    if "--write" in sys.argv:
        path = os.path.join(P, "README.md")
        s = open(path).read()
        pat = re.compile(rf"<!-- BEGIN {tag} .*?<!-- END {tag} -->", re.S)
        if pat.search(s):
            s = pat.sub(lambda _: txt, s)
        else:   # in front of the first round section
            i = s.index("## Round")
            s = s[:i] + txt + "\n\n" + s[i:]
        open(path, "w").write(s)
    else:
        print(txt)
