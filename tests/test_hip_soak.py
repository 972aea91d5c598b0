"""-m gpu: short forms of the two builder-run checks of rounds 2-3, so that the DRIVER's suite carries them too (<= 20 s together):
   * device memory after create / train / eval / destroy cycles (long form: tests/tools/leak_check.py, 60 cycles),
   * bit-reproducibility of repeated identical steps at W = 512 and in the opt-in bf16x3 mode - no float atomics anywhere, so a
     difference is a race (long form: tests/tools/repro_soak.py, 300 steps x 8 configurations; W = 256 fp32 has its own 300-step
     soak in tests/test_hip_train.py)."""
import contextlib
import gc
import io

import numpy as np
import pytest
import torch

import cfnerf_amd
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

pytestmark = pytest.mark.gpu


def _cycle(i):
    """one life of a model: create (packed weights, event pools), a train step (lent workspace, weight-gradient plan), an eval
    render, destroy"""
    cfg = O.OracleCfg(netwidth=[64, 128, 256][i % 3], K_samples=4)
    with contextlib.redirect_stdout(io.StringIO()):
        _, kw_train, kw_test, model, _, _ = build_model(cfg, i)
    rng = np.random.default_rng(i)
    rays, (H, W, focal) = fern_rays(rng, 256)
    tr = TR.Trainer(model.module, beta1=0.01)
    tr.step(H, W, focal, rays.cuda(), torch.rand(256, 3, device="cuda"))
    with torch.no_grad():
        cfnerf_amd.render(H, W, focal, rays=rays.cuda(), **kw_test)
    torch.cuda.synchronize()


def _free_bytes():
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]          # hipMemGetInfo: what the DEVICE reports, not torch's allocator


def test_device_memory_is_unchanged_after_model_life_cycles():
    for i in range(3):                           # warm every width once: code objects, torch's pools, the fork server's share
        _cycle(i)
    f0 = _free_bytes()
    for i in range(10):
        _cycle(i)
    f1 = _free_bytes()
    assert abs(f0 - f1) <= 2 << 20, f"device memory free before {f0 / 2 ** 20:.1f} MiB, after 10 create / train / eval / destroy cycles {f1 / 2 ** 20:.1f} MiB"


@pytest.mark.parametrize("W,K,N,prec", [(512, 8, 512, "fp32"), (256, 4, 1024, "bf16x3")])
def test_repeated_steps_are_bit_identical(W, K, N, prec):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    with contextlib.redirect_stdout(io.StringIO()):
        _, kw_train, kw_test, model, _, _ = build_model(cfg, 5)
    net = model.module
    net.set_precision(prec)
    rng = np.random.default_rng(W + K)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    rays = rays.cuda()
    target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32).cuda()
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32).cuda()
    eps = torch.tensor(rng.standard_normal((K, 4)), dtype=torch.float32).cuda()
    tr = TR.Trainer(net, beta1=0.01)
    ref, diffs = None, 0
    for _ in range(50):
        g = tr.forward_backward(H, Wd, focal, rays, target, t_rand=t_rand, eps=eps).clone()
        cur = (g, tr.rgb_map.clone(), tr.scalars.clone())
        if ref is None:
            ref = cur
            assert torch.isfinite(g).all() and float(g.abs().max()) > 0
        elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
            diffs += 1
    with torch.no_grad():
        e0 = cfnerf_amd.render(H, Wd, focal, rays=rays, **kw_test)[0].clone()
        ed = sum(0 if torch.equal(e0, cfnerf_amd.render(H, Wd, focal, rays=rays, **kw_test)[0]) else 1 for _ in range(15))
    net.release_workspace()
    assert diffs == 0 and ed == 0, f"W={W} K={K} N={N} {prec}: {diffs} of 49 repeated train steps and {ed} of 15 eval renders differ from the first"
