"""Development probe (round 5): do workgroups of a SECOND stream run beside the step's kernels on the same CUs, and what does that cost the step?
A streaming elementwise kernel (torch.add over 512 MB: no LDS, few registers, 256 threads) loops on a side stream while the train step runs
on the main stream with every stage timed; reported: the stage medians with and without the side traffic and the side kernel's own
throughput alone and beside the step.  Motivation: the small weight-gradient launch is HBM-bound (MFMA-busy 0.53), the big one MFMA-bound -
if a light kernel co-resides with the big one, the small jobs could run in its shadow.
    python tests/tools/coresidency_probe.py [--config C2] [--steps 100]"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--mb", type=int, default=512)
a = ap.parse_args()
torch.cuda.set_device(0)
wl = bench.Workload(a.config, None, 0, 1, torch.device("cuda", 0), "fp32")
h = wl.net.handle
wl.lib.cfnerf_timing_enable(h, 1)
names = (("fwd", 0), ("bwd_tail", 1), ("bwd_data", 2), ("bwd_dw", 3), ("adam", 4))
side = torch.cuda.Stream()
x = torch.ones(a.mb * 1024 * 1024 // 4, device="cuda")
y = torch.empty_like(x)


def side_burst(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            torch.add(x, 1.0, out=y)


def run(with_side):
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    acc = {k: [] for k, _ in names}
    steps = []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bursts = []
    for _ in range(a.steps):
        if with_side:
            s0.record(side)
            side_burst(40)            # keeps the side stream busy for longer than one step
            s1.record(side)
        ev[0].record()
        wl.step()
        ev[1].record()
        ev[1].synchronize()
        steps.append(ev[0].elapsed_time(ev[1]))
        for k, i in names:
            acc[k].append(wl.lib.cfnerf_timing_last_ms(h, i))
        if with_side:
            side.synchronize()        # the burst's remainder runs alone; the next step starts with a fresh burst
            bursts.append(s0.elapsed_time(s1))
    out = {"step_ms": round(statistics.median(steps), 4), **{k: round(statistics.median(v), 4) for k, v in acc.items()}}
    if bursts:
        out["side_burst_of_40_ms"] = round(statistics.median(bursts), 4)
    return out


base = run(False)
torch.cuda.synchronize()
# the side kernel alone
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
side_burst(5)
side.synchronize()
e0.record(side)
side_burst(40)
e1.record(side)
side.synchronize()
alone_ms = e0.elapsed_time(e1) / 40
both = run(True)
print(json.dumps({"config": a.config, "side_kernel": f"torch.add over {a.mb} MB in + {a.mb} MB out", "side_alone_ms": round(alone_ms, 4),
                  "side_alone_TBps": round(2 * a.mb / 1024 / 1024 / (alone_ms * 1e-3), 3) if alone_ms else None,
                  "side_burst_of_40_alone_ms": round(alone_ms * 40, 4), "step_alone": base, "step_beside_side_traffic": both}))
