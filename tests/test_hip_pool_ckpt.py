"""-m gpu: device-side ray pool (RUN:860-884, 942-951; get_rays_np HLP:350-357) and checkpoint interop (RUN:345-378, 1085-1100)."""
import os
import tempfile

import numpy as np
import pytest
import torch

import cfnerf_amd
from oracle import cfnerf_oracle as O
from util_hip import build_model, close, make_args

pytestmark = pytest.mark.gpu


def test_ray_pool_matches_get_rays_and_feeds_every_pixel_once_per_epoch():
    rng = np.random.default_rng(0)
    V, H, W, focal = 4, 6, 8, 9.5
    images = torch.tensor(rng.uniform(0, 1, (V, H, W, 3)), dtype=torch.float32)
    poses = torch.zeros(V, 3, 5)
    for v in range(V):
        a = 0.2 * v
        poses[v, :, :3] = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        poses[v, :, 3] = torch.tensor(rng.uniform(-1, 1, 3))
    i_train = [0, 2, 3]
    pool = cfnerf_amd.RayPool(images, poses, H, W, focal, i_train, N_rand=16, shuffle=False)
    assert len(pool) == 3 * H * W and pool.rays_rgb.shape == (3 * H * W, 3, 3)
    for slot, v in enumerate(i_train):                      # the unshuffled pool is get_rays per view + the pixels
        ro, rd = O.get_rays(H, W, focal, poses[v, :3, :4])
        blk = pool.rays_rgb[slot * H * W:(slot + 1) * H * W].cpu()
        close(blk[:, 0], ro.reshape(-1, 3), atol=0, rtol=0, what="origins")
        close(blk[:, 1], rd.reshape(-1, 3), atol=1e-6, rtol=1e-6, what="directions")
        assert torch.equal(blk[:, 2], images[v].reshape(-1, 3))
    ref_rows = pool.rays_rgb.reshape(-1, 9).cpu()
    pool._shuffle()
    seen = []
    for _ in range(len(pool) // 16):
        batch_rays, target = pool.next_batch()
        assert batch_rays.shape == (2, 16, 3) and target.shape == (16, 3)
        seen.append(torch.cat([batch_rays[0], batch_rays[1], target], -1).cpu())
    seen = torch.cat(seen)
    assert pool.epoch == 1 and pool.i_batch == 0            # re-shuffled after the epoch (RUN:946-949)
    key = lambda t: sorted(map(tuple, t.numpy().round(6).tolist()))
    assert key(seen) == key(ref_rows)


def test_sharded_device_pool_feeds_the_global_batch_across_an_epoch_boundary():
    """RayPool(rank, world) on the device (rays from the product's kernel): four shards in the seed form (no exchange) - at every step the
    concatenation of the shards is the batch the one-process pool with N_rand_global = 4 x 6 delivers; every pixel once per epoch."""
    rng = np.random.default_rng(1)
    V, H, W, focal = 3, 5, 7, 8.0
    images = torch.tensor(rng.uniform(0, 1, (V, H, W, 3)), dtype=torch.float32)
    poses = torch.zeros(V, 3, 4)
    for v in range(V):
        poses[v, :, :3] = torch.eye(3)
        poses[v, :, 3] = torch.tensor(rng.uniform(-1, 1, 3))
    world, n_rand = 4, 6
    one = cfnerf_amd.RayPool(images, poses, H, W, focal, [0, 1, 2], N_rand=world * n_rand, seed=3)
    shards = [cfnerf_amd.RayPool(images, poses, H, W, focal, [0, 1, 2], N_rand=n_rand, rank=r, world=world, seed=3, sync="seed") for r in range(world)]
    assert len(one) == 105 and all(s.rays_rgb.is_cuda and torch.equal(s.rays_rgb, one.rays_rgb) for s in shards)
    seen = []
    for step in range(11):                                   # 105 = 4 x 24 + 9: the last window of an epoch is short and not a multiple of 4
        rays1, tgt1 = one.next_batch()
        got = [s.next_batch() for s in shards]
        rays, tgt = torch.cat([g[0] for g in got], 1), torch.cat([g[1] for g in got], 0)
        n = rays.shape[1]
        assert n in (world * n_rand, 8) and all(g[0].shape[1] == n // world for g in got)
        assert torch.equal(rays, rays1[:, :n]) and torch.equal(tgt, tgt1[:n])
        assert all(s.epoch == one.epoch and s.i_batch == one.i_batch for s in shards)
        if one.epoch == 0 or step < 5:
            seen.append(tgt.cpu())
    assert one.epoch == 2
    first_epoch = torch.cat(seen[:5])                        # 4 full windows + the short one (8 of its 9 rays)
    assert first_epoch.shape[0] == 104 and len({tuple(r) for r in first_epoch.numpy().round(6).tolist()}) == 104


def test_checkpoint_roundtrip_in_the_reference_format():
    cfg = O.OracleCfg(netwidth=64, K_samples=2)
    _, kw_train, _, model, p, optimizer = build_model(cfg, 9)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "{:06d}_{:02d}.tar".format(1234, 1))              # RUN:1086
        cfnerf_amd.save_checkpoint(path, 1234, model, optimizer)
        ck = torch.load(path, map_location="cpu")
        assert set(ck) == {"global_step", "network_fn_state_dict", "optimizer_state_dict"}
        keys = set(ck["network_fn_state_dict"])
        assert "module.pts_linears.5.weight" in keys and "module.flows_rgb.flow_3.diag_idx" in keys and "module.flows_alpha.triu_mask" in keys
        assert not any(k.endswith("sample_alpha") for k in keys)                 # R9: eval latents are not in the state_dict
        for k, v in p.items():
            assert torch.equal(ck["network_fn_state_dict"]["module." + k], v)
        # reload through create_nerf(ft_path=...) into a fresh model (RUN:345-374)
        args = make_args(cfg, ft_path=path, no_reload=False)
        kw2, _, start, _, _ = cfnerf_amd.create_nerf(args)
        assert start == 1234
        m2 = kw2["network_fn"].module
        for k, v in p.items():
            assert torch.equal(m2.view(k).cpu(), v), k
        # a checkpoint with extra / missing keys is filtered like RUN:370
        ck["network_fn_state_dict"]["module.not_a_key"] = torch.zeros(3)
        del ck["network_fn_state_dict"]["module.rgb_mean"]
        torch.save(ck, path)
        kw3, _, _, _, _ = cfnerf_amd.create_nerf(make_args(cfg, ft_path=path, no_reload=False))
        m3 = kw3["network_fn"].module
        assert torch.equal(m3.view("pts_linears.0.weight").cpu(), p["pts_linears.0.weight"])
        assert torch.equal(m3.view("rgb_mean").cpu(), torch.zeros(3))           # kept its init value


def test_checkpoint_written_by_the_reference_loads_and_renders():
    """G22: the .tar the REFERENCE's own save lines wrote (RUN:1085-1100; tests/golden/g22_reference_ckpt.tar, made by
    tests/golden/make_golden.py from the real nn.DataParallel(NeRF_Flows) after two iterations of its loop) goes through
    create_nerf(ft_path=...) -> _maybe_reload (the loader's rules, RUN:345-378): training resumes at its global_step, every parameter is
    the checkpoint's bit for bit, and the fused eval render from those weights equals what the reference rendered from them."""
    import numpy as np
    from conftest import GOLDEN
    from util_hip import close
    g = dict(np.load(os.path.join(GOLDEN, "g22_reference_ckpt.npz"), allow_pickle=False))
    tar = os.path.join(GOLDEN, "g22_reference_ckpt.tar")
    cfg = O.OracleCfg(netwidth=int(g["netwidth"]), netdepth=int(g["netdepth"]), K_samples=int(g["K"]))
    args = make_args(cfg, ft_path=tar, no_reload=False)
    kw_train, kw_test, start, _, _ = cfnerf_amd.create_nerf(args)
    assert start == int(g["global_step"]) == 2
    net = kw_train["network_fn"].module
    ck = torch.load(tar, map_location="cpu", weights_only=True)["network_fn_state_dict"]
    sd = kw_train["network_fn"].state_dict()
    assert set(sd) == set(ck)
    for k, v in ck.items():
        assert torch.equal(sd[k].cpu().to(v.dtype), v), k
    net.sample_alpha, net.sample_rgb = torch.tensor(g["sample_alpha"]), torch.tensor(g["sample_rgb"])
    with torch.no_grad():
        rgbs, disp, depth, _ = cfnerf_amd.render(int(g["H"]), int(g["W"]), float(g["focal"]), rays=torch.tensor(g["rays"]).cuda(), near=0., far=1., **kw_test)
    close(rgbs, g["rgb_map_eval"], what="rgb_map_eval")
    close(depth, g["depth_map_eval"], what="depth_map_eval")
    # ... and a checkpoint this build writes from that model holds the same key LIST in the same order with the same tensors
    with tempfile.TemporaryDirectory() as d:
        out = cfnerf_amd.save_checkpoint(os.path.join(d, "000002_01.tar"), 2, kw_train["network_fn"])
        mine = torch.load(out, map_location="cpu", weights_only=True)
    assert list(mine) == ["global_step", "network_fn_state_dict", "optimizer_state_dict"] and mine["global_step"] == 2
    assert set(mine["network_fn_state_dict"]) == set(ck)
    for k, v in ck.items():
        assert torch.equal(mine["network_fn_state_dict"][k].to(v.dtype), v), k
