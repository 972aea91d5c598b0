"""-m gpu: bench.py keeps the driver's contract - ONE JSON line last on stdout with the agreed keys (metric / value / unit /
n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the
`roofline`, `cpu_baseline` and `psnr` objects, the numbers in it are consistent with each other, and `--gpus N` works both
under a launcher and on its own.

bench.py runs as __main__ inside a fresh child of the fork server tests/conftest.py starts at collection time, never as a
fork+exec of this (possibly GPU-initialised) pytest process."""
import json
import os
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None, timeout=900):
    from conftest import FORKSERVER_CTX as ctx
    assert ctx is not None, "the fork server must have been started at collection time (tests/conftest.py)"
    import mp_workers
    with tempfile.NamedTemporaryFile("r", suffix=".out", delete=False) as f:
        path = f.name
    try:
        env = dict(env or {})
        env.setdefault("CFNERF_BENCH_LIVE_PMC", "0")        # the rocprofv3 child passes of the default line: only where a test asks for them
        p = ctx.Process(target=mp_workers.bench_child, args=(list(args), path, env))
        p.start()
        p.join(timeout)
        assert p.exitcode == 0, f"bench.py {' '.join(args)} exited with {p.exitcode}"
        with open(path) as f:
            out = f.read()
    finally:
        os.unlink(path)
    last = out.strip().splitlines()[-1]
    return json.loads(last)


def test_default_line_has_the_contract_keys_and_is_self_consistent():
    d = run_bench("--steps", "6", "--warmup", "2", "--no-alt")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "ms_per_step_median"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("C2") and "model" not in d["config"]
    # value = rays of the job / time: 1024 rays per step
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"]
    assert 0.9 * d["ms_per_step"] <= d["ms_per_step_median"] <= 1.05 * d["ms_per_step"]
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
    # eval WITHOUT --no-alt: the train-only side measurements (stress row, extension, PSNR) must stay out of an eval line
    d = run_bench("--mode", "eval", "--steps", "4", "--warmup", "1", "--no-cpu-baseline")
    assert d["config"]["workload"].startswith("C2") and "eval" in d["config"]["workload"] and d["value"] > 0
    assert "stress_w512" not in d and "alt_config" not in d and "psnr" not in d and "kernel_ms" not in d
    d = run_bench("--config", "C4", "--steps", "3", "--warmup", "1", "--no-alt", "--no-cpu-baseline")
    assert d["config"]["workload"].startswith("C4") and "K=16" in d["config"]["workload"]


def test_default_line_observes_its_own_counters():
    """roofline.traffic / mfma_busy_frac_pmc of the default line are measured by the run itself: rocprofv3 --pmc child passes of the same
    workload, started before the bench process touches the GPU (falls back to the committed figures only without rocprofv3)."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("no rocprofv3 on this box")
    d = run_bench("--steps", "4", "--warmup", "1", "--psnr-steps", "0", "--no-cpu-baseline", env={"CFNERF_BENCH_LIVE_PMC": "1"})
    r = d["roofline"]
    assert r["traffic_source"] == "live" and "OBSERVED BY THIS RUN" in r["traffic_unit"]
    # train forward: ~1.46 GB of stash written + ~0.1 GB read per launch; matrix pipe 75-90 % busy
    assert 1.2e9 <= r["traffic"] <= 2.2e9 and 0.6 <= r["mfma_busy_frac_pmc"] <= 0.95
    assert abs(r["traffic"] - (2 * r["pmc_live"]["FETCH_SIZE_KB"] + r["pmc_live"]["WRITE_SIZE_KB"]) * 1024) <= 1e-6 * r["traffic"]


def test_psnr_block_is_in_the_default_line():
    """the second half of the headline metric: held-out PSNR on the synthetic stand-in scene, plus the HIP-vs-oracle agreement"""
    d = run_bench("--steps", "4", "--warmup", "1", "--psnr-steps", "300")
    p = d["psnr"]
    assert "synthetic stand-in for LLFF-fern" in p["scene"] and p["unit"] == "dB" and p["steps"] == 300
    c = p["held_out_psnr_db_by_step"]
    assert set(c) == {"0", "250", "300"} and c["300"] > c["0"] + 3.0, c          # it trains: 8.6 dB at init, ~18 dB at 250 steps
    assert p["value"] == c["300"]
    v = p["vs_oracle"]
    assert v["agree"] is True and v["abs_psnr_diff_db_step0"] <= 1e-3 and v["abs_psnr_diff_db_step25"] <= v["tolerance_db"]
    r = p["vs_reference_run"]               # ... and against the real reference's own 120-step training run (fixture G19)
    assert r["agree"] is True and set(r["max_abs_held_out_psnr_diff_db"]) == {"0", "40", "80", "120"}
    assert r["max_abs_held_out_psnr_diff_db"]["120"] <= r["tolerance_db_at_last_step"] <= 0.01
    assert r["reference_held_out_psnr_db"]["120"] > r["reference_held_out_psnr_db"]["0"] + 4.0
    for k in ("alt_precision", "stress_w512", "eval", "cpu_baseline"):
        assert k in d, k
    assert "alt_config" not in d                                             # the parity-unpinned extension is opt-in (--extension) since round 5
    # the other half of SURVEY 8(d)'s metric in the default line: eval-render throughput of the C2 batch and of config 5 (800 x 800, K 32)
    e = d["eval"]
    e2, e5 = e["config2_eval"], e["config5_full_image"]
    assert e2["workload"].startswith("C2") and "mode=eval" in e2["workload"] and e2["unit"] == "rays/s"
    assert abs(e2["value"] - 1024 / (e2["ms_per_step"] * 1e-3)) <= 1e-6 * e2["value"]
    assert e2["fwd_launch_ms"] <= e2["ms_per_step"] * 1.02 and 0.4 < e2["roofline"]["frac"] < 1.0
    assert e2["value"] > 2.0 * d["value"]                                   # eval = one of the step's three GEMM passes, and no stash
    assert e5["workload"].startswith("C5") and "K=32" in e5["workload"] and e5["images"] >= 2
    assert abs(e5["value"] - 640000 / e5["s_per_image"]) <= 1e-6 * e5["value"]
    assert e5["fwd_launch_ms"] * 1e-3 <= e5["s_per_image"] * 1.02 and 0.4 < e5["roofline"]["frac"] < 1.0


@pytest.mark.parametrize("argv,workload", [(("--config", "C4", "--steps", "4", "--warmup", "2"), "C4"),
                                           (("--config", "C5", "--steps", "1", "--warmup", "1"), "C5")])
def test_gpus_2_starts_its_own_ranks(argv, workload):
    """`python bench.py --gpus 2` with no launcher above it: the process starts its two ranks itself (here both on cuda:0,
    exchanging over gloo), prints ONE line with n_gpus == 2, and the whole job is about as fast as the same config on one rank
    (two ranks time-share the one GPU, so the aggregate rate cannot exceed it and should not fall far below)."""
    env = {"CFNERF_BENCH_SAME_GPU": "1"}
    one = run_bench("--gpus", "1", *argv, "--no-alt", "--no-cpu-baseline")
    two = run_bench("--gpus", "2", *argv, "--no-alt", "--no-cpu-baseline", env=env)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["workload"].startswith(workload) and "dp2" in two["config"]["parallelism"]
    assert "SAME_GPU" in two["config"]["parallelism"]
    if workload == "C4":
        assert "all-reduce" in two["config"]["parallelism"]
        # per rank: 1024 rays per step at both N; value = world * 1024 * steps / time
        assert abs(two["value"] - 2 * 1024 / (two["ms_per_step"] * 1e-3)) <= 1e-6 * two["value"]
    ratio = two["value"] / one["value"]
    assert 0.5 <= ratio <= 1.10, (ratio, one["value"], two["value"])


def test_gpus_2_default_line_explains_itself():
    """The N > 1 line carries the whole metric and its own diagnosis: `psnr` (sharded ray pool + Trainer on world x 1024-ray global
    batches, with the pass/fail `vs_single_process`), `comm` (exposed exchange time from two HIP events around the all-reduce, payload,
    the backend and world size torch.distributed reports), `rank_skew`, and `cpu_baseline` on rank 0.  Two ranks on cuda:0 over gloo."""
    d = run_bench("--gpus", "2", "--steps", "4", "--warmup", "2", "--psnr-steps", "60", env={"CFNERF_BENCH_SAME_GPU": "1"}, timeout=1500)
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("C2")
    c = d["comm"]
    assert c["backend"] == "gloo" and c["world_size_reported"] == 2 and c["steps"] == 4
    assert c["payload_bytes"] >= 617410 * 4 and c["form"] == "one all-reduce"
    assert 0.0 < c["exposed_ms_min"] <= c["exposed_ms_mean"] <= c["exposed_ms_max"]
    assert c["exposed_ms_mean_min_over_ranks"] <= c["exposed_ms_mean"] <= c["exposed_ms_mean_max_over_ranks"] + 1e-9
    assert 0.0 < c["frac_of_step"] < 1.0
    k = d["rank_skew"]
    assert k["ms_per_step_min"] <= k["ms_per_step_max"] and abs(k["ms_per_step_max"] - d["ms_per_step"]) <= 1e-6 * d["ms_per_step"]
    assert k["fwd_launch_ms_min"] <= d["roofline"]["launch_ms"] <= k["fwd_launch_ms_max"] + 1e-9
    p = d["psnr"]
    assert p["n_gpus"] == 2 and p["global_batch"] == 2048 and p["steps"] == 60 and "world=2" in p["feeder"]
    cv = p["held_out_psnr_db_by_step"]
    assert cv["60"] > cv["0"] + 1.0, cv                                     # it trains across the ranks
    v = p["vs_single_process"]
    assert v["agree"] is True and v["abs_psnr_diff_db"] <= v["tolerance_db"] == 0.05
    assert v["max_abs_param_diff_rel"] < 5e-2                               # (Adam sign flips of ~0 gradients move single weights)
    assert p["vs_oracle"]["agree"] is True and p["vs_reference_run"]["agree"] is True
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0
    assert "config4_k16" in d and "alt_precision" in d
    e = d["eval"]                                                           # eval render at N = 2: C2 rays per rank, config 5's rows tiled over the ranks
    assert abs(e["config2_eval"]["value"] - 2 * 1024 / (e["config2_eval"]["ms_per_step"] * 1e-3)) <= 1e-6 * e["config2_eval"]["value"]
    assert "rows 0..400 of 800" in e["config5_full_image"]["workload"] and e["config5_full_image"]["value"] > 0


def test_k64_config_runs():
    """the reference's default latent count (--K_samples 64, RUN:631) as a bench configuration"""
    d = run_bench("--config", "K64", "--steps", "3", "--warmup", "1", "--no-alt", "--no-cpu-baseline")
    assert d["config"]["workload"].startswith("K64") and "K=64" in d["config"]["workload"] and d["value"] > 0
    assert d["kernel_ms"]["bwd_tail"] > 0.1                                 # the tail kernel is K-proportional: ~0.34 ms at K = 64
