"""world_size-2 gloo test (CPU) of the multi-GPU semantics of cf-nerf_amd/train.py: rays sharded per rank,
each rank differentiates  nll_local_sum / (3 N_total) + (beta1 / world) * entropy_local, ONE all-reduce(sum)
of the flat gradient reproduces the single-process gradient of the global means.  The per-rank arithmetic is
played by the CPU oracle here (the HIP kernels need a GPU); the sharding / normalisation / collective code is
the product's (shard_bounds, allreduce_sum_, lr_at)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cfnerf_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(grads, cfg):
    out = []
    for k, shp in O.param_shapes(cfg).items():
        g = grads[k]
        out.append(torch.zeros(int(np.prod(shp))) if g is None else g.reshape(-1))
    return torch.cat(out)


def _problem():
    cfg = O.OracleCfg(netwidth=64, K_samples=3)
    rng = np.random.default_rng(0)
    N = 8
    o = torch.tensor(rng.uniform(-0.2, 0.2, (N, 3)), dtype=torch.float32)
    d = torch.tensor(rng.standard_normal((N, 3)) * 0.2 + np.array([0, 0, -1.0]), dtype=torch.float32)
    packed = O.pack_rays(378, 504, 407.5, o, d, True, 0., 1.)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((3, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((3, 3)), dtype=torch.float32)
    target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32)
    return cfg, O.make_params(cfg, 3), packed, t_rand, ea, er, target


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cfnerf_amd.train as TR
    torch.set_num_threads(2)
    cfg, p, packed, t_rand, ea, er, target = _problem()
    lo, hi = TR.shard_bounds(packed.shape[0], rank, world)
    beta1 = 0.05
    # local shard, same latent samples on every rank
    scal, grads, _ = O.train_step(p, packed[lo:hi], target[lo:hi], cfg, ea, er, t_rand[lo:hi], beta1)
    g = _flat(grads, cfg) / world             # = grad of nll_local_sum/(3 N_total) + (beta1/world) entropy_local
    TR.allreduce_sum_(g, world)
    if rank == 0:
        q.put(g.numpy())
    dist.destroy_process_group()


def test_two_rank_ray_sharding_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    g2 = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    cfg, p, packed, t_rand, ea, er, target = _problem()
    scal, grads, _ = O.train_step(p, packed, target, cfg, ea, er, t_rand, 0.05)
    g1 = _flat(grads, cfg).numpy()
    scale = np.abs(g1).max()
    assert np.abs(g2 - g1).max() <= 2e-5 * scale + 1e-9, np.abs(g2 - g1).max() / scale


def test_shard_bounds_and_lr_schedule():
    import cfnerf_amd.train as TR
    assert [TR.shard_bounds(8192, r, 8) for r in (0, 7)] == [(0, 1024), (7168, 8192)]
    try:
        TR.shard_bounds(10, 0, 4)
        assert False
    except ValueError:
        pass
    # RUN:1073-1077: lr = lrate * 0.1 ** (global_step / (lrate_decay * 1000)), applied after the step
    assert TR.lr_at(5e-4, 250, 0, 0) == 5e-4
    assert abs(TR.lr_at(5e-4, 250, 0, 250001) - 5e-5) < 1e-12
    for s in (0, 10, 1000):
        assert abs(TR.lr_at(5e-4, 250, 100, s + 1) - O.lr_schedule(5e-4, 250, 100 + s)) < 1e-15


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cfnerf_amd import evaluate as E
    H, W = 7, 5                                      # 7 rows over 2 ranks: shards of 4 and 3 rows
    full = {"rgb_mean": torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3), "rgb_unc": torch.ones(H, W, 3) * 0.5,
            "disp_mean": torch.arange(H * W, dtype=torch.float32).reshape(H, W), "depth_mean": -torch.arange(H * W, dtype=torch.float32).reshape(H, W),
            "sq_err": torch.full((H, W, 3), 0.25)}
    r0, r1 = E.row_shard(H, rank, world)
    got = E.gather_rows({k: v[r0:r1].clone() for k, v in full.items()}, H, world, rank)
    if rank == 0:
        q.put({k: (v.numpy() if torch.is_tensor(v) and v.ndim else float(v)) for k, v in got.items()})
    else:
        assert got is None
    dist.destroy_process_group()


def test_row_tiled_evaluation_gathers_the_full_image_on_rank_0():
    """image rows tiled across ranks with no exchange for the render; the optional gather of the fused maps to rank 0"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    H, W = 7, 5
    np.testing.assert_array_equal(got["rgb_mean"], np.arange(H * W * 3, dtype=np.float32).reshape(H, W, 3))
    np.testing.assert_array_equal(got["disp_mean"], np.arange(H * W, dtype=np.float32).reshape(H, W))
    np.testing.assert_array_equal(got["depth_mean"], -np.arange(H * W, dtype=np.float32).reshape(H, W))
    assert got["rgb_unc"].shape == (H, W, 3) and abs(got["mse"] - 0.25) < 1e-7


# ---- sharded ray pool (replaces RUN:860-884, 942-951 + the DataParallel split of RUN:330 across processes) ----------
_POOL_M, _POOL_NRAND, _POOL_STEPS = 1003, 64, 20          # 1003 rays, windows of 2 x 64: the epoch's last window is short AND odd


def _pool_table():
    return torch.arange(_POOL_M * 9, dtype=torch.float32).reshape(_POOL_M, 3, 3)


def _pool_worker(rank, world, port, q, sync):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cfnerf_amd.data import RayPool
    torch.manual_seed(100 + rank)                         # DIFFERENT global seeds per rank: the permutation must still be ONE
    if sync == "broadcast":                               # rank 0's own generator draws it, the others receive it
        pool = RayPool.from_rays_rgb(_pool_table(), _POOL_NRAND, rank=rank, world=world, sync="broadcast",
                                     generator=torch.Generator().manual_seed(77 + rank))
    else:
        pool = RayPool.from_rays_rgb(_pool_table(), _POOL_NRAND, rank=rank, world=world, sync="seed", seed=77)
    assert pool.sync == sync and pool.global_batch == world * _POOL_NRAND
    out = []
    for _ in range(_POOL_STEPS):
        rays, target = pool.next_batch()
        out.append((rays.clone().numpy(), target.clone().numpy(), pool.epoch, pool.i_batch))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _run_pool(sync):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pool_worker, args=(r, world, port, q, sync)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return got


def test_sharded_ray_pool_union_of_shards_is_the_single_process_batch():
    """Two ranks over gloo, rank 0's permutation broadcast once per epoch: at every step - across two epoch boundaries with a short,
    odd last window - the concatenation of the shards is the batch ONE process with N_rand_global = 2 x 64 draws from the same
    permutation, the ranks re-shuffle at the same step, and no ray is delivered twice inside an epoch."""
    from cfnerf_amd.data import RayPool
    got = _run_pool("broadcast")
    # the one-process pool replays rank 0's draws: same generator seed
    one = RayPool.from_rays_rgb(_pool_table(), 2 * _POOL_NRAND, generator=torch.Generator().manual_seed(77))
    seen, epochs = set(), []
    for step in range(_POOL_STEPS):
        e_before = one.epoch
        rays1, tgt1 = one.next_batch()
        (r0, t0, ep0, ib0), (r1, t1, ep1, ib1) = got[0][step], got[1][step]
        assert (ep0, ib0) == (ep1, ib1) == (one.epoch, one.i_batch), step          # same epoch / cursor on every rank at every step
        assert r0.shape == r1.shape and r0.shape[1] in (_POOL_NRAND, (_POOL_M % (2 * _POOL_NRAND)) // 2)
        union_r, union_t = np.concatenate([r0, r1], 1), np.concatenate([t0, t1], 0)
        n = union_r.shape[1]
        assert n == rays1.shape[1] - (rays1.shape[1] % 2)                         # a ragged window loses at most world - 1 rays
        np.testing.assert_array_equal(union_r, rays1.numpy()[:, :n])
        np.testing.assert_array_equal(union_t, tgt1.numpy()[:n])
        ids = set((union_t[:, 0] // 9).astype(int).tolist())                      # row id of the table the ray came from
        if e_before != (epochs[-1] if epochs else 0):
            seen = set()
        epochs.append(e_before)
        assert not (ids & seen), step
        seen |= ids
    assert one.epoch == 2                                                         # two re-shuffles were crossed


def test_sharded_ray_pool_seed_form_needs_no_exchange():
    from cfnerf_amd.data import RayPool
    got = _run_pool("seed")
    one = RayPool.from_rays_rgb(_pool_table(), 2 * _POOL_NRAND, seed=77)
    for step in range(_POOL_STEPS):
        rays1, tgt1 = one.next_batch()
        union_t = np.concatenate([got[0][step][1], got[1][step][1]], 0)
        np.testing.assert_array_equal(union_t, tgt1.numpy()[:union_t.shape[0]])


def _pool_mismatch_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cfnerf_amd.data import RayPool
    try:                                                   # the ranks DISAGREE about the seed: what another device kind / generator would look like
        RayPool.from_rays_rgb(_pool_table(), _POOL_NRAND, rank=rank, world=world, sync="seed", seed=77 + rank)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_seed_form_notices_ranks_that_drew_different_permutations():
    """sync="seed" exchanges nothing per step; once per epoch the ranks compare a checksum of the permutation they drew, so ranks whose
    generators disagree (another device kind, a host-resident pool on one rank) fail loudly instead of training on overlapping shards."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pool_mismatch_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all("DIFFERENT permutations" in got[r] for r in range(world)), got
