"""-m gpu: the real multi-rank PRODUCT path.  Two ranks (fresh processes forked from the clean fork server that
tests/conftest.py starts before anything touches the GPU) share cuda:0, each runs cfnerf_amd.train.Trainer.step on
its shard of the rays with world_size = 2 and they exchange the flat gradient over a gloo group; after three steps
their parameters equal those of one process that trained on the full batch.  Replaces the seam nn.DataParallel
(RUN:330,336)."""
import socket

import numpy as np
import pytest
import torch

from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(spec, world=2):
    from conftest import FORKSERVER_CTX as ctx
    assert ctx is not None, "the fork server must have been started at collection time (tests/conftest.py)"
    import mp_workers
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=mp_workers.trainer_rank, args=(r, world, port, q, spec)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, status, a, b, c = q.get(timeout=600)
        assert status == "ok", f"rank {rank} failed:\n{a}"
        got[rank] = (a, b, c)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def _single_process(spec, eps_per_step):
    cfg = O.OracleCfg(netwidth=spec["W"], K_samples=spec["K"], h_alpha_size=spec.get("ha", 32))
    _, kw_train, _, model, p, _ = build_model(cfg, spec["seed"])
    rng = np.random.default_rng(spec["data_seed"])
    N = spec["N"]
    rays, (H, Wd, focal) = fern_rays(rng, N)
    target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32)
    tr = TR.Trainer(model, lrate=5e-4, lrate_decay=250, beta1=spec["beta1"])
    losses = []
    for step in range(spec["steps"]):
        if step in spec.get("precision_at", {}):
            model.module.set_precision(spec["precision_at"][step])
        t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
        sc = tr.step(H, Wd, focal, rays.to(DEV), target.to(DEV), t_rand=t_rand.to(DEV), eps=torch.tensor(eps_per_step[step]).to(DEV))
        losses.append(sc[:2].cpu().numpy())
    return model.module.flat.detach().cpu().numpy(), np.array(losses)


@pytest.mark.parametrize("explicit_eps", [False, True])
def test_two_ranks_on_the_product_path_match_the_single_process_full_batch(explicit_eps):
    spec = dict(W=64, K=4, N=64, seed=21, data_seed=5, beta1=0.01, steps=3, explicit_eps=explicit_eps)
    got = _run_ranks(spec)
    flat0, losses0, eps0 = got[0]
    flat1, losses1, eps1 = got[1]
    assert np.array_equal(flat0, flat1), "the two ranks diverged (no parameter broadcast after step 0: updates must be identical)"
    if explicit_eps:
        eps = [np.random.default_rng(7000 + s).standard_normal((spec["K"], 4)).astype(np.float32) for s in range(spec["steps"])]
    else:
        # the ranks were seeded DIFFERENTLY; rank 0's draws reached rank 1 (step 0: broadcast, later: all-reduce tail)
        assert np.array_equal(eps0, eps1) and len(eps0) == spec["steps"]
        assert np.abs(eps0[0] - eps0[1]).max() > 1e-3          # fresh latents every step
        eps = list(eps0)
    flat_ref, losses_ref = _single_process(spec, eps)
    # three Adam steps move a weight by ~1.5e-3; sharded and full-batch gradients differ by summation order only
    # (an entry whose ~0 gradient changes sign with the summation order moves by up to 2 lr in Adam's first steps:
    # allow a vanishing fraction of those)
    d = np.abs(flat0 - flat_ref)
    assert (d > 1e-6).mean() <= 1e-3, ((d > 1e-6).mean(), d.max())
    assert d.max() <= 3 * 2 * 5e-4 + 1e-6, d.max()
    np.testing.assert_allclose(losses0, losses_ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("overlap", [True, False])
def test_rccl_exchange_path_on_one_rank_is_the_identity(overlap):
    """The RCCL ("nccl") exchange of the product - with overlap: the early bucket gathered and all-reduced on a side stream
    under the small-job launch (cfnerf_grad_early_ranges / cfnerf_stream_wait_grad_early), the late bucket after - on
    a one-rank group, where a sum all-reduce changes nothing: the parameters after three steps equal those of the
    plain single-process run bit for bit, so every element went through exactly one bucket and came back in place."""
    spec = dict(W=64, K=4, N=64, seed=21, data_seed=5, beta1=0.01, steps=3, explicit_eps=True, backend="nccl", force=True, overlap=overlap)
    got = _run_ranks(spec, world=1)
    flat0, losses0, _ = got[0]
    eps = [np.random.default_rng(7000 + s).standard_normal((spec["K"], 4)).astype(np.float32) for s in range(spec["steps"])]
    flat_ref, losses_ref = _single_process(spec, eps)
    assert np.array_equal(flat0, flat_ref)
    np.testing.assert_allclose(losses0, losses_ref, rtol=0, atol=0)


def test_two_bucket_exchange_survives_a_change_of_the_stash_layout():
    """h_alpha_size = 128: the job g_ha x h runs in the big launch while the stash is row-major (bf16x3) and in the small-job launch with the
    Q4 layout (fp32, whole tiles).  The Trainer caches the early ranges after its first backward; round 5 reported h_alpha_linear.weight as
    early in one layout and late in the other, so a run that changed the precision between steps all-reduced a stale early bucket BEFORE the
    small jobs wrote that tensor and copied it back over the right gradient.  Such a job is never early now (DwTile::late): the two-bucket
    exchange on a one-rank RCCL group across fp32 -> bf16x3 -> fp32 equals the plain single-process run bit for bit."""
    spec = dict(W=128, ha=128, K=4, N=64, seed=23, data_seed=6, beta1=0.01, steps=4, explicit_eps=True, backend="nccl", force=True, overlap=True,
                precision_at={1: "bf16x3", 3: "fp32"})
    got = _run_ranks(spec, world=1)
    flat0, losses0, _ = got[0]
    eps = [np.random.default_rng(7000 + s).standard_normal((spec["K"], 4)).astype(np.float32) for s in range(spec["steps"])]
    flat_ref, losses_ref = _single_process(spec, eps)
    assert np.array_equal(flat0, flat_ref)
    np.testing.assert_allclose(losses0, losses_ref, rtol=0, atol=0)
