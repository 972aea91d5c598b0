"""-m gpu: the C ABI is re-entrant per model handle (include/cfnerf.h, "Conventions"): two host threads, each with its own
model, workspace and HIP stream, run train steps and eval renders at the same time; every result must be bit-identical to
the same work done alone.  (ctypes releases the GIL inside every library call, so the calls really overlap.)"""
import threading

import numpy as np
import pytest
import torch

import cfnerf_amd
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _job(seed, W, K, N, steps, out, use_stream):
    """`steps` fused train steps + one eval render of a private model; returns parameters, scalars and the eval maps."""
    try:
        cfg = O.OracleCfg(netwidth=W, K_samples=K)
        stream = torch.cuda.Stream() if use_stream else torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            _, kw_train, kw_test, model, _, _ = build_model(cfg, seed)
            rng = np.random.default_rng(seed)
            net = model.module                      # eval latents: create_nerf drew them from torch's global generator
            net.sample_rgb = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
            net.sample_alpha = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
            rays, (H, Wd, focal) = fern_rays(rng, N)
            rays = rays.to(DEV)
            target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32, device=DEV)
            tr = TR.Trainer(model, beta1=0.01)
            scal = []
            for i in range(steps):
                t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32, device=DEV)
                eps = torch.tensor(rng.standard_normal((K, 4)), dtype=torch.float32, device=DEV)
                scal.append(tr.step(H, Wd, focal, rays, target, t_rand=t_rand, eps=eps).clone())
            with torch.no_grad():
                rgb, disp, depth, _ = cfnerf_amd.render(H, Wd, focal, rays=rays, **kw_test)
            stream.synchronize()
            out[seed] = (model.module.flat.detach().clone(), torch.stack(scal), rgb.clone(), depth.clone())
    except Exception as e:      # surfaced by the main thread
        out[seed] = e


def test_two_models_on_two_threads_match_their_solo_runs():
    jobs = [(11, 256, 4, 512, 6), (12, 128, 8, 300, 6), (13, 64, 3, 700, 6)]
    solo = {}
    for seed, W, K, N, steps in jobs:
        _job(seed, W, K, N, steps, solo, use_stream=False)
        assert not isinstance(solo[seed], Exception), solo[seed]
    torch.cuda.synchronize()
    for _round in range(2):
        both = {}
        threads = [threading.Thread(target=_job, args=(seed, W, K, N, steps, both, True)) for seed, W, K, N, steps in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        for seed, *_ in jobs:
            assert not isinstance(both[seed], Exception), both[seed]
            for a, b, what in zip(solo[seed], both[seed], ("parameters", "scalars", "eval rgb_map", "eval depth_map")):
                assert torch.equal(a, b), f"model {seed}: {what} differ between the solo and the concurrent run"
