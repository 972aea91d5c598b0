#!/usr/bin/env python3
"""Sharper same-box A/B of library builds than tools/ab_bench.sh: ONE process per build (CFNERF_LIB selects it), `--steps` train steps
of a bench.py config with EVERY stage timed by HIP events on every step (cfnerf_timing_enable(m, 1)), median / mean / p10 per stage.
bench.py's kernel_ms comes from the last of three such steps - too noisy to see a 0.5 % effect.
    python tools/ab_kernels.py [--config C2] [--steps 200] [--mode train|eval]"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--mode", default=None)
ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
torch.cuda.set_device(0)
wl = bench.Workload(a.config, a.mode, 0, 1, torch.device("cuda", 0), a.precision)
h = wl.net.handle
wl.lib.cfnerf_timing_enable(h, 1)
names = (("fwd", 0),) + ((("bwd_tail", 1), ("bwd_data", 2), ("bwd_dw", 3), ("adam", 4)) if wl.mode == "train" else ())
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
acc = {k: [] for k, _ in names}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
steps = []
for _ in range(a.steps):
    ev[0].record()
    wl.step()
    ev[1].record()
    torch.cuda.synchronize()
    steps.append(ev[0].elapsed_time(ev[1]))
    for k, i in names:
        acc[k].append(wl.lib.cfnerf_timing_last_ms(h, i))


def summ(v):
    v = sorted(v)
    return {"median": round(statistics.median(v), 4), "mean": round(sum(v) / len(v), 4), "p10": round(v[len(v) // 10], 4)}


print(json.dumps({"lib": os.environ.get("CFNERF_LIB", "default").split("/")[-1], "config": a.config, "mode": wl.mode, "steps": a.steps,
                  "step_ms": summ(steps), **{k: summ(v) for k, v in acc.items()}}))
