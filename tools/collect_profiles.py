#!/usr/bin/env python3
"""Assemble profiles/rNN_* from the output of tools/profile_round.sh (gpurun_out/i_*):
   kernel-stats CSVs as rocprofv3 wrote them, one PMC summary text, and rNN_traffic.json (what bench.py's roofline.traffic reads).
   usage: python tools/collect_profiles.py r03"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(ROOT, "profiles")

for n, out in (("train", "train"), ("eval", "eval"), ("train_b16", "train_bf16x3"), ("w512", "w512_train"), ("c4", "c4_train"), ("c3", "c3_train"), ("k64", "k64_train")):
    src = os.path.join(G, f"i_{n}", f"{n}_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{out}_kernel_stats.csv"))


def pmc(name):
    """kernel -> {counter: mean per dispatch} (+ avg_us) from one rocprofv3 --pmc pass"""
    path = os.path.join(G, f"i_{name}", "pmc_counter_collection.csv")
    acc, dur = {}, {}
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cfnerf::", "")
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            dur.setdefault(k, {})[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, avg_us=sum(dur[k].values()) / len(dur[k])) for k, cs in acc.items()}


with open(os.path.join(P, f"{tag}_pmc_summary.txt"), "w") as out:
    out.write("# rocprofv3 --kernel-trace --pmc <counters> (one counter set per pass, tools/profile_round.sh), mean per dispatch.\n"
              "# FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE reports HALF of a wide coalesced read on gfx950 (MI355X_MICROARCH.md): double it.\n"
              "# MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).\n")
    for name in ("pmc_mfma", "pmc_fetch", "pmc_write", "pmc_mfma_eval", "pmc_fetch_eval", "pmc_write_eval", "pmc_b16", "pmc_mfma_w512",
                 "pmc_fetch_w512", "pmc_write_w512"):
        d = pmc(name)
        out.write(f"## {name}\n")
        for k in sorted(d, key=lambda k: -d[k]["avg_us"]):
            line = f"{k:44s} avg_us={d[k]['avg_us']:9.1f}"
            for c, v in sorted(d[k].items()):
                if c != "avg_us":
                    line += f"  {c}={v:.4g}"
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d[k] and d[k].get("GRBM_GUI_ACTIVE"):
                line += f"  mfma_busy_frac={d[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d[k]['GRBM_GUI_ACTIVE'] / 8):.3f}"
            out.write(line + "\n")

traffic = {}
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/%s_pmc_summary.txt, recipe tools/profile_round.sh); FETCH_SIZE "
        "doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); WRITE_SIZE uncorrected; mfma_busy_frac = "
        "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)" % tag)
for key, suffix, kern in (("C2:train", "", "fused_fwd_kernel<256, 0, true, 0, true>"), ("C2:eval", "_eval", "fused_fwd_kernel<256, 0, false, 0, false>"),
                          ("W512:train", "_w512", "fused_fwd_kernel<512, 0, true, 0, true>")):      # (train: the Q4 variant, round 5)
    f, w, m = pmc("pmc_fetch" + suffix), pmc("pmc_write" + suffix), pmc("pmc_mfma" + suffix)
    if kern in f and kern in w:
        e = {"kernel": kern, "FETCH_SIZE_KB": f[kern]["FETCH_SIZE"], "WRITE_SIZE_KB": w[kern]["WRITE_SIZE"],
             "hbm_bytes_per_launch": (2 * f[kern]["FETCH_SIZE"] + w[kern]["WRITE_SIZE"]) * 1024.0, "note": note}
        if kern in m:
            e["mfma_busy_frac"] = m[kern]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m[kern]["GRBM_GUI_ACTIVE"] / 8)
        # whole-step traffic of the big kernels, for DESIGN.md
        e["step_kernels_hbm_bytes"] = {k: (2 * f[k]["FETCH_SIZE"] + w.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024.0 for k in f if f[k]["FETCH_SIZE"] > 1000}
        traffic[key] = e
with open(os.path.join(P, f"{tag}_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
# bench lines of the round: gpurun_out/bench_<tag>_<name>.json (written by tools/bench_round.sh) -> profiles/<tag>_bench_<name>.json
import glob
for src in sorted(glob.glob(os.path.join(G, f"bench_{tag}_*.json"))):
    with open(src) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith("{")]
    if lines:
        name = os.path.basename(src)[len(f"bench_{tag}_"):]
        with open(os.path.join(P, f"{tag}_bench_{name}"), "w") as o:
            o.write(json.dumps(json.loads(lines[-1])) + "\n")
print("wrote", sorted(x for x in os.listdir(P) if x.startswith(tag)))
