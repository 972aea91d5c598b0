#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X-native CF-NeRF ray-batch hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic fern-shaped rays at
BASELINE.json configs[1]: N_rand = 1024 rays per GPU, S = 128 samples (the reference's hard-coded
table, single pass - it has no fine network), K = 4 latent samples, W = 256, D = 8.
  --mode train (default when the backward is built): forward + KDE-NLL loss + backward + Adam
  --mode eval : fused forward render only
Rays are sharded across ranks (weak scaling: N_rand per GPU is fixed); the only exchange is one
RCCL all-reduce of the flat gradient per train step.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FP32_MFMA_PEAK_TF = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
N_RAND, S, K, W, D = 1024, 128, 4, 256, 8


def gemm_flops_per_point(W=256, ha=32, hr=64, ic=63, icv=27, F=4):
    """Algorithmic MACs per point (deduplicated flow heads) x 2."""
    macs = ic * W + 4 * W * W + (W + ic) * W + 2 * W * W          # trunk (D = 8: layers 0, 1-4, 5 (skip), 6-7)
    macs += W * ha + W * W + (W + icv) * (W // 2) + (W // 2) * hr  # h_alpha, feature, views, h_rgb
    macs += hr * 18 * F + ha * 3 * F                               # live flow-parameter heads
    return 2 * macs


def fern_rays(rng, n, H=378, W=504, focal=407.5658):
    """Fern-shaped synthetic rays: LLFF fern at factor 8, near-identity pose, n random pixels."""
    c2w = np.eye(4, dtype=np.float32)[:3]
    c2w[:, 3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    pix = rng.choice(H * W, size=n, replace=False)
    j, i = np.divmod(pix, W)
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i, dtype=np.float64)], -1)
    rays_d = (dirs[:, None, :] * c2w[:3, :3]).sum(-1).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape).astype(np.float32)
    return torch.tensor(np.stack([rays_o, rays_d], 0)), (H, W, focal)


def synth_batch(rank, n, device):
    rng = np.random.default_rng(1000 + rank)
    rays, (H, Wd, focal) = fern_rays(rng, n)
    target = torch.tensor(rng.uniform(0, 1, (n, 3)), dtype=torch.float32)
    return rays.to(device), target.to(device), (H, Wd, focal)


def cpu_baseline(mode, budget_s=30.0):
    """The CPU oracle (op-for-op PyTorch-CPU restatement of the reference, oracle/) timed on this box's
    host cores on a bounded sample of the same workload.  The thread count is swept (all usable
    cores is NOT the fastest for these GEMM sizes) and the best setting is reported with its count."""
    from oracle import cfnerf_oracle as O      # the ONLY use of oracle/ in this file: the checker timed as the CPU baseline
    n_rays = N_RAND                              # the full per-GPU workload of one step
    cfg = O.OracleCfg(netwidth=W, K_samples=K)
    p = O.make_params(cfg, 0)
    rng = np.random.default_rng(5)
    rays, (H, Wd, focal) = fern_rays(rng, n_rays)
    packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.)
    t_rand = torch.rand(n_rays, S)
    ea, er = torch.randn(K, 1), torch.randn(K, 3)
    target = torch.rand(n_rays, 3)
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1

    def one():
        t0 = time.perf_counter()
        if mode == "train":
            scal, grads, _ = O.train_step(p, packed, target, cfg, ea, er, t_rand, 0.01)
            O.adam_step({k: v.clone() for k, v in p.items()}, grads, {}, 1, 5e-4)
        else:
            with torch.no_grad():
                O.render_rays(p, packed, cfg, ea, er, False)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    best, best_thr, tried = float("inf"), 1, []
    torch.set_num_threads(min(usable, 16))
    one()                                       # warm-up (thread pool, allocator, autograd graph caches)
    for thr in (16, 32, 8, 64):                 # more threads than 64 only thrash on these GEMM sizes
        if thr > usable or time.perf_counter() - t_start > budget_s:
            continue
        torch.set_num_threads(thr)
        dt = one()
        tried.append(thr)
        if dt < best:
            best, best_thr = dt, thr
    out = {"value": n_rays / best, "unit": "rays/s", "cores": best_thr, "kind": "port",
           "sample": f"{n_rays} rays x {S} samples x K={K}, W={W}: one {mode} step of the PyTorch-CPU oracle per thread count, best at "
                     f"{best_thr} threads (tried {tried}; {usable} usable cores; anomaly detection off)"}
    if mode == "train" and time.perf_counter() - t_start < budget_s:
        # the reference ships with torch.autograd.set_detect_anomaly(True) (HLP:2, MOD:5): one more step that way
        torch.set_num_threads(best_thr)
        with torch.autograd.set_detect_anomaly(True):
            out["value_anomaly_on"] = n_rays / one()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "eval"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra bf16x3 measurement reported next to the fp32 line")
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default="fp32",
                    help="fp32 = exact-fp32 MFMA (default, the measured parity path); bf16x3 = opt-in split-bf16 MFMA mode")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("CFNERF_BENCH_FORCE_DIST") == "1"      # exercise the RCCL path on one GPU (tests)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if force_dist and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    import cfnerf_amd
    from cfnerf_amd import _lib as L
    try:
        from cfnerf_amd import train as T
        have_train = T.backward_available()
    except Exception:
        T, have_train = None, False
    mode = args.mode or ("train" if have_train else "eval")
    if mode == "train" and not have_train:
        raise SystemExit("train mode requested but the backward kernels are not built")

    torch.manual_seed(0)                        # same random-init weights on every rank (nn.Linear-style init of the product)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):                # create_nerf prints the reference's "No reloading" notice
        kw_train, kw_test, _, _, _ = cfnerf_amd.create_nerf(cfnerf_amd.default_args(netwidth=W, netdepth=D, K_samples=K, device=dev))
    model = kw_train["network_fn"]
    net = model.module
    rays, target, (H, Wd, focal) = synth_batch(rank, N_RAND, dev)
    lib = L.lib()
    lib.cfnerf_timing_enable(net.handle, 1)
    net.set_precision(args.precision)
    g = torch.Generator(device=dev).manual_seed(1234)           # same latent samples on every rank

    if mode == "train":
        trainer = T.Trainer(net, lrate=5e-4, lrate_decay=250, beta1=0.01, world_size=world, force_allreduce=force_dist)

    def step():
        t_rand = torch.rand(N_RAND, S, device=dev)
        eps = torch.randn(K, 4, device=dev, generator=g)
        if mode == "train":
            return trainer.step(H, Wd, focal, rays, target, t_rand=t_rand, eps=eps, **kw_train)
        with torch.no_grad():
            return cfnerf_amd.render(H, Wd, focal, rays=rays, **kw_test)

    def sync():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    kern_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP-event duration of the dominant kernel (fused forward) on the launch stream
    sync()
    dt = time.perf_counter() - t0
    # per-launch duration of the dominant kernel from the library's HIP events (last step; queried after the timed region)
    fwd_ms = lib.cfnerf_timing_last_ms(net.handle, 0)
    extra_ms = {}
    if mode == "train":
        for name, idx in (("bwd_tail", 1), ("bwd_data", 2), ("bwd_dw", 3), ("adam", 4)):
            extra_ms[name] = lib.cfnerf_timing_last_ms(net.handle, idx)
    if world > 1 or force_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the opt-in split-bf16 mode, measured the same way right after (every rank runs it: the all-reduce is inside)
    alt = None
    if args.precision == "fp32" and not args.no_alt:
        net.set_precision("bf16x3")
        for _ in range(min(3, args.warmup) or 1):
            step()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dta = time.perf_counter() - t1
        alt_fwd_ms = lib.cfnerf_timing_last_ms(net.handle, 0)
        if world > 1 or force_dist:
            t = torch.tensor([dta], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dta = float(t.item())
        net.set_precision("fp32")
        alt = {"precision": "bf16x3 (opt-in): fp32 operands carried as hi+lo bf16, product = hi*hi + hi*lo + lo*hi on "
                            "v_mfma_f32_32x32x16_bf16, fp32 accumulate; forward, backward-data and the large weight-gradient GEMMs; "
                            "held to the same parity tolerances (tests/test_hip_bf16x3.py)",
               "value": N_RAND * world * args.steps / dta, "unit": "rays/s", "ms_per_step": dta / args.steps * 1e3,
               "fwd_launch_ms": alt_fwd_ms}

    if rank == 0:
        rays_per_s = N_RAND * world * args.steps / dt
        fl = gemm_flops_per_point(W) * N_RAND * S          # forward GEMM FLOPs of one launch of the fused forward kernel
        # HBM bytes per launch of that kernel from the committed PMC passes (rocprofv3 cannot run inside the bench)
        traffic, mfma_busy = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                tj = json.load(f)[mode]
            traffic, mfma_busy = tj["hbm_bytes_per_launch"], tj.get("mfma_busy_frac")
        except Exception:
            pass
        roof = {"bound": "mfma", "kernel": f"fused_fwd_kernel<256,rays,{'train' if mode == 'train' else 'eval'}>",
                "achieved": fl / (fwd_ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "traffic": traffic,
                "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_summary.txt)",
                "mfma_busy_frac_pmc": mfma_busy, "launch_ms": fwd_ms, "flops_per_launch": fl}
        roof["frac"] = roof["achieved"] / roof["peak"]
        out = {
            "metric": "rays/sec (train step)" if mode == "train" else "rays/sec (eval render, fused forward)",
            "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16x3-split MFMA, f32 accumulate (opt-in mode)", "data": "synthetic",
            "config": {"workload": f"LLFF-fern-shaped synthetic rays, N_rand={N_RAND}/GPU, S={S} (reference table, single pass), "
                                   f"K={K}, W={W}, D={D}, NDC, mode={mode}, precision={args.precision}",
                       "parallelism": f"ray-sharded dp{world}" + (", 1 RCCL all-reduce of flat grads/step" if mode == "train" and world > 1 else "")},
            "roofline": roof,
        }
        if extra_ms:
            out["kernel_ms"] = dict(fwd=fwd_ms, **extra_ms)
        if alt is not None:
            alt["fwd_fp32_equiv_tflops"] = fl / (alt["fwd_launch_ms"] * 1e-3) / 1e12
            out["alt_precision"] = alt
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mode)
    # the JSON line is the LAST thing on stdout: RCCL prints its version banner through libc's buffered stdout (it would
    # otherwise surface at process exit, after the line), so every rank flushes that before the final barrier
    def flush_c_stdio():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
    flush_c_stdio()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
