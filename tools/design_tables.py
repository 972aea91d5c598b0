#!/usr/bin/env python3
"""Rewrite the two measured tables of DESIGN.md from the committed evidence of a round (so that the prose can never drift from the files):
   the kernel table of section 3 (between the KERNEL-TABLE markers: rocprofv3 averages of profiles/rNN_*_kernel_stats.csv) and the bench table of
   section 4 (between the BENCH-TABLE markers: profiles/rNN_bench_*.json).   usage: python tools/design_tables.py r06"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
PEAK = 157.3


def avg(csvname, kernel):
    with open(os.path.join(P, csvname)) as f:
        hit = [r for r in csv.DictReader(f) if kernel in r["Name"]]
    assert len(hit) == 1, (csvname, kernel, len(hit))
    return float(hit[0]["AverageNs"]) / 1e6


def line(n):
    with open(os.path.join(P, f"{tag}_bench_{n}.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def num(v):
    return f"{v:,.0f}".replace(",", " ")


rows = [("`fused_fwd_kernel<256, 0, true, 0, true>` (train, Q4)", "train", "fused_fwd_kernel<256, 0, true, 0, true>", 160.99, "of 157.3 TF"),
        ("`fused_fwd_kernel<256, 0, false, 0, false>` (eval)", "eval", "fused_fwd_kernel<256, 0, false, 0, false>", 160.99, ""),
        ("`bwd_data_kernel<256, 0>`", "train", "bwd_data_kernel<256, 0>", 160.99, ""),
        ("`dw_big_kernel<0>`", "train", "dw_big_kernel<0>", 146.0, "(146 GFLOP)"),
        ("`dw_small_kernel`", "train", "dw_small_kernel", 15.5, "(15.5 GFLOP); HBM floor ≈140 µs"),
        ("`tail_bwd_kernel`", "train", "tail_bwd_kernel", None, "VALU issue"),
        ("`tail_bwd_kernel` at K = 64", "k64_train", "tail_bwd_kernel", None, "VALU issue"),
        ("`fused_fwd_kernel<512, 0, true, 0, true>` (W512)", "w512_train", "fused_fwd_kernel<512, 0, true, 0, true>", 308.57, "(W512: 308.6 GFLOP)")]


def pmc_bytes(kernel):
    """HBM bytes per launch from the round's PMC summary (first FETCH_SIZE / WRITE_SIZE pass that lists the kernel; KB, FETCH x 2 on gfx950)"""
    import re
    f = w = None
    for ln in open(os.path.join(P, f"{tag}_pmc_summary.txt")):
        if ln.startswith(kernel):
            m = re.search(r"FETCH_SIZE=([0-9.e+]+)", ln)
            if m and f is None:
                f = float(m.group(1))
            m = re.search(r"WRITE_SIZE=([0-9.e+]+)", ln)
            if m and w is None:
                w = float(m.group(1))
    return (2 * f + w) * 1024


kt = "| kernel (C2 unless named) | csv (`profiles/`) | rocprofv3 avg ms | of its bound |\n|---|---|---|---|\n"
for label, stats, kern, gf, note in rows:
    name = f"{tag}_{stats}_kernel_stats.csv"
    a = avg(name, kern)
    frac = f"{gf / a / PEAK:.3f} {note}".strip() if gf else note
    if kern == "dw_small_kernel":               # its bound is HBM: quote both
        b = pmc_bytes("dw_small_kernel")
        frac = (f"{gf / a / PEAK:.3f} of the MFMA peak (15.5 GFLOP); **HBM-bound**: {b / 1e9:.3f} GB (PMC) ÷ {a:.4f} ms = {b / a / 1e9:.2f} TB/s = "
                f"{b / a / 1e9 / 8.0:.2f} of the 8 TB/s spec ({b / a / 1e9 / 6.29:.2f} of the 6.29 TB/s measured achievable)")
    kt += f"| {label} | `{name}` | {a:.4f} | {frac} |\n"

bt = ("| config | rays/s | ms / step | fused forward ms (frac of 157.3 TF, HIP events) | backward-data | dW leg | tail | whole step of peak |\n"
      "|---|---|---|---|---|---|---|---|\n")
for label, n in (("**C2** train (N 1024, K 4) — the default line", "default"), ("C2 train (`--config C2`)", "c2"), ("C3 train (N 4096, K 8, no NDC)", "c3"),
                 ("C4 shard (N 1024, K 16)", "c4"), ("**K64** (N 1024, K 64: the reference's default, RUN:631)", "k64"),
                 ("**W512** authors' recipe (W 512, h_α 64, K 32, N 512)", "w512"), ("N8192: 8192 rays per step in 8 slices of 1024", "n8192")):
    x = line(n)
    k = x["kernel_ms"]
    bt += (f"| {label} | {num(x['value'])} | {x['ms_per_step']:.3f} | {x['roofline']['launch_ms']:.4f} ({x['roofline']['frac']:.3f}) | {k['bwd_data']:.3f} | "
           f"{k['bwd_dw']:.3f} | {k['bwd_tail']:.3f} | {x['step_frac_of_peak']:.3f} |\n")
e, c1, c5, D = line("eval"), line("c1"), line("c5"), line("default")
bt += f"| C2 eval (`--mode eval`) | {num(e['value'])} | {e['ms_per_step']:.3f} | {e['roofline']['launch_ms']:.4f} ({e['roofline']['frac']:.3f}) | | | | |\n"
bt += f"| C1 (N 256, K 1, forward) | {num(c1['value'])} | {c1['ms_per_step']:.3f} | ({c1['roofline']['frac']:.3f}) | | | | |\n"
bt += f"| C5 (800 × 800, K 32 eval, fused uncertainty) | {num(c5['value'])} = {c5['ms_per_step'] / 1e3:.4f} s / image | | ({c5['roofline']['frac']:.3f}) | | | | |\n"
ap = D["alt_precision"]
bt += f"| C2 bf16x3 (opt-in, `alt_precision`) | {num(ap['value'])} | {ap['ms_per_step']:.3f} | | | | | {ap['roofline']['frac']:.3f} of its 833 TF-equivalent |\n"
cb, sb = D["cpu_baseline"], D["sliced_batch_n8192"]
bt += (f"\nIn the default line too: `cpu_baseline` (the oracle on the host cores, the 1024-ray C2 step, thread count swept) {cb['value']:.0f} rays/s at {cb['cores']} threads; "
       f"`roofline.traffic` observed by the run {D['roofline']['traffic'] / 1e9:.3f} GB per forward launch (1.46 GB of it the stash; algorithmic ≈ 3.1 MB), PMC MFMA-busy "
       f"{D['roofline']['mfma_busy_frac_pmc']:.3f}; `sliced_batch_n8192` {num(sb['value'])} rays/s on a {sb['workspace_bytes'] / 2 ** 30:.2f} GiB workspace; "
       f"`psnr` {D['psnr']['value']:.1f} dB after {D['psnr']['steps']} steps ({D['psnr']['wall_s_incl_eval']:.1f} s incl. the evaluations), held-out PSNR within "
       f"{D['psnr']['vs_reference_run']['max_abs_held_out_psnr_diff_db']['120']:.1e} dB of the real reference's own run at step 120 (G19).\n")

path = os.path.join(ROOT, "DESIGN.md")
d = open(path).read()


def put(d, a, b, body):
    i, j = d.index(a), d.index(b)
    i = d.index("\n", i) + 1
    return d[:i] + body + d[j:]


d = put(d, "<!-- KERNEL-TABLE", "<!-- /KERNEL-TABLE -->", kt)
d = put(d, "<!-- BENCH-TABLE", "<!-- /BENCH-TABLE -->", bt)
open(path, "w").write(d)
print(kt)
print(bt)
