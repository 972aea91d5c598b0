"""-m gpu: bench.py keeps the driver's contract - ONE JSON line last on stdout with the agreed keys (metric / value / unit /
n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the
`roofline` and `cpu_baseline` objects, and the numbers in it are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    return json.loads(last)


def test_default_line_has_the_contract_keys_and_is_self_consistent():
    d = run_bench("--steps", "6", "--warmup", "2", "--no-alt")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("C2") and "model" not in d["config"]
    # value = rays of the job / time: 1024 rays per step
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["peak"] - 157.3) < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["launch_ms"] * 1e-3) / 1e12) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    # the kernels of one step cannot take longer than the step
    assert sum(d["kernel_ms"].values()) <= d["ms_per_step"] * 1.05     # (stage times come from extra, event-timed steps)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "rays/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["value"] > 50 * c["value"]


def test_eval_mode_and_other_configs_run():
    d = run_bench("--mode", "eval", "--steps", "4", "--warmup", "1", "--no-alt", "--no-cpu-baseline")
    assert d["config"]["workload"].startswith("C2") and "eval" in d["config"]["workload"] and d["value"] > 0
    d = run_bench("--config", "C4", "--steps", "3", "--warmup", "1", "--no-alt", "--no-cpu-baseline")
    assert d["config"]["workload"].startswith("C4") and "K=16" in d["config"]["workload"]
