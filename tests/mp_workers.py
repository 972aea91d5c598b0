"""Rank bodies of the multi-process tests (importable by name from a fork-server child)."""
import os


def trainer_rank(rank, world, port, q, spec):
    """One rank of a world-size-`world` run of the PRODUCT train step (cfnerf_amd.train.Trainer on the HIP kernels),
    every rank on cuda:0, exchanging over a gloo group.  Rank 0 reports the parameters after the last step."""
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group(spec.get("backend", "gloo"), rank=rank, world_size=world,
                                **({"device_id": torch.device("cuda", 0)} if spec.get("backend") == "nccl" else {}))
        import cfnerf_amd                                      # noqa: F401
        from cfnerf_amd import train as TR
        from oracle import cfnerf_oracle as O
        from util_hip import build_model, fern_rays
        cfg = O.OracleCfg(netwidth=spec["W"], K_samples=spec["K"], h_alpha_size=spec.get("ha", 32))
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            _, kw_train, _, model, p, _ = build_model(cfg, spec["seed"])
        rng = np.random.default_rng(spec["data_seed"])
        N = spec["N"]
        rays, (H, Wd, focal) = fern_rays(rng, N)
        target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32)
        lo, hi = TR.shard_bounds(N, rank, world)
        tr = TR.Trainer(model, lrate=5e-4, lrate_decay=250, beta1=spec["beta1"], world_size=world, force_allreduce=spec.get("force", False),
                        overlap_comm=spec.get("overlap", True))
        torch.manual_seed(1000 + rank)                        # DIFFERENT seeds per rank: the latents must still agree
        eps_used, losses = [], []
        for step in range(spec["steps"]):
            if step in spec.get("precision_at", {}):           # the arithmetic mode (and with it the stash layout) changes between steps
                model.module.set_precision(spec["precision_at"][step])
            t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
            kw = dict(t_rand=t_rand[lo:hi].cuda())
            if spec.get("explicit_eps"):
                kw["eps"] = torch.tensor(np.random.default_rng(7000 + step).standard_normal((spec["K"], 4)), dtype=torch.float32).cuda()
            else:
                e = tr._step_eps()                             # what step() is about to use (idempotent)
                eps_used.append(e.cpu().numpy().copy())
                kw["eps"] = e
            sc = tr.step(H, Wd, focal, (rays[0, lo:hi].cuda(), rays[1, lo:hi].cuda()), target[lo:hi].cuda(), **kw)
            if not spec.get("explicit_eps"):
                pass
            s = sc[:2].clone() if spec.get("backend") == "nccl" else sc[:2].clone().cpu()
            dist.all_reduce(s)                                 # loss, nll: contributions sum to the global value
            s = s.cpu()
            losses.append(s.numpy())
        torch.cuda.synchronize()
        q.put((rank, "ok", model.module.flat.detach().cpu().numpy(), np.array(losses), np.array(eps_used)))
        dist.destroy_process_group()
    except Exception as e:                                     # surface the failure in the parent instead of a timeout
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))
        raise


def bench_child(argv, out_path, env):
    """bench.py as __main__ inside a FRESH fork-server child (it has not touched the GPU, so bench.py may both use the GPU and -
    for `--gpus N` without a launcher - start its own ranks); stdout (fd 1, C stdio included) goes to `out_path`."""
    import runpy
    import sys
    os.environ.update(env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fd = os.open(out_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.dup2(fd, 1)
    os.close(fd)
    sys.stdout = os.fdopen(1, "w", buffering=1, closefd=False)
    sys.argv = [os.path.join(root, "bench.py"), *argv]
    os.chdir(root)
    try:
        runpy.run_path(sys.argv[0], run_name="__main__")
    except SystemExit as e:
        sys.stdout.flush()
        code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        if code:
            print(e.code, file=sys.stderr)
        os._exit(code)
    sys.stdout.flush()


def run_program(cmd, out_path):
    """Run an external program from a FRESH fork-server child (a process that has never touched the GPU may fork + exec); its
    stdout + stderr go to `out_path`, its exit code becomes this child's."""
    import subprocess
    with open(out_path, "w") as f:
        r = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT)
    os._exit(r.returncode if 0 <= r.returncode < 256 else 255)
