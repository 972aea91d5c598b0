"""-m gpu: full-image evaluation (render_path_train RUN:247-314, uncertainty maps RUN:1117-1131) on the HIP path."""
import numpy as np
import pytest
import torch

import cfnerf_amd
from cfnerf_amd import evaluate as E
from oracle import cfnerf_oracle as O
from util_hip import ATOL_DISP, build_model, close

pytestmark = pytest.mark.gpu
T = lambda a: torch.tensor(np.asarray(a))


def test_render_path_train_vs_reference_golden(golden):
    g = golden("g6_render_c2w")
    cfg = O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=int(g["K"]))
    _, kw_train, kw_test, model, p, _ = build_model(cfg, int(g["seed"]))
    net = model.module
    net.sample_alpha = T(g["eps_alpha"]).clone()
    net.sample_rgb = T(g["eps_rgb"]).clone()
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    pose = torch.cat([T(g["c2w"]), torch.tensor([[H, W, focal]], dtype=torch.float32).t()], 1)     # [3,5] like LLFF poses
    rgbs, disps = cfnerf_amd.render_path_train(pose, (H, W, focal), 8192, kw_test)
    assert rgbs.shape == (1, H, W, 3, 4) and disps.shape == (1, H, W, 4) and isinstance(rgbs, np.ndarray)
    close(rgbs[0], g["rgb_map"], what="rgbs")
    close(disps[0], g["disp_map"], atol=ATOL_DISP, rtol=1e-3, what="disps")
    with pytest.raises(NotImplementedError):
        cfnerf_amd.render_path_train(torch.stack([pose, pose]), (H, W, focal), 8192, kw_test)


@pytest.mark.parametrize("white_bkgd,ndc,K", [(True, False, 32), (False, True, 4)])
def test_fused_uncertainty_maps_match_the_per_k_maps(white_bkgd, ndc, K):
    """Config-5 shaped (Blender-like intrinsics, white background, K = 32) at a small resolution."""
    cfg = O.OracleCfg(netwidth=256, K_samples=K)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, 55, white_bkgd=white_bkgd, no_ndc=not ndc)
    net = model.module
    H, W, focal = 20, 24, 33.3
    th, ph = np.deg2rad(30.0), np.deg2rad(-30.0)
    c2w = torch.tensor([[np.cos(th), -np.sin(th) * np.sin(ph), np.sin(th) * np.cos(ph), 4 * np.sin(th) * np.cos(ph)],
                        [0, np.cos(ph), np.sin(ph), 4 * np.sin(ph)],
                        [-np.sin(th), -np.cos(th) * np.sin(ph), np.cos(th) * np.cos(ph), 4 * np.cos(th) * np.cos(ph)]],
                       dtype=torch.float32)
    near, far = (2.0, 6.0) if not ndc else (0.0, 1.0)
    if ndc:
        c2w = torch.tensor([[1, 0, 0, 0.1], [0, 1, 0, -0.1], [0, 0, 1, 0.0]], dtype=torch.float32)
    kw = dict(near=near, far=far, ndc=ndc, white_bkgd=white_bkgd)
    full = E.render_uncertainty(H, W, focal, c2w, model, want_maps=True, **kw)
    rgbs = full["rgb_map"].cpu().numpy()                       # [H,W,3,K]
    n = rgbs.shape[-1]
    close(full["rgb_mean"], np.mean(rgbs, -1), atol=1e-6, rtol=1e-5, what="rgb_mean")                     # RUN:1122
    close(full["rgb_unc"], np.std(rgbs, -1) * n / (n - 1), atol=1e-6, rtol=1e-4, what="rgb_unc")          # RUN:1129-1130
    close(full["disp_mean"], np.mean(full["disp_map"].cpu().numpy(), -1), atol=1e-5, rtol=1e-5, what="disp_mean")
    close(full["depth_mean"], np.mean(full["depth_map"].cpu().numpy(), -1), atol=1e-6, rtol=1e-5, what="depth_mean")
    # the per-K maps equal what render() returns for the same pose
    with torch.no_grad():
        r = cfnerf_amd.render(H, W, focal, c2w=c2w, near=near, far=far, **kw_test)
    assert torch.equal(r[0], full["rgb_map"])
    # row tiling across "ranks" reproduces the image bit-for-bit, without the per-K maps leaving the chip
    parts = [E.render_uncertainty(H, W, focal, c2w, model, rows=E.row_shard(H, rk, 3), **kw) for rk in range(3)]
    assert torch.equal(torch.cat([q["rgb_mean"] for q in parts], 0), full["rgb_mean"])
    assert torch.equal(torch.cat([q["rgb_unc"] for q in parts], 0), full["rgb_unc"])
    # per-pixel squared error of the K-mean prediction (the integrand of img2mse(rgb_mean, target), RUN:1028) from the same
    # launch (cfnerf_render_eval): equals the torch expression on the fused means, also on a row shard
    gt = torch.tensor(np.random.default_rng(3).uniform(0, 1, (H, W, 3)), dtype=torch.float32)
    fe = E.render_uncertainty(H, W, focal, c2w, model, gt=gt, **kw)
    assert torch.equal(fe["rgb_mean"], full["rgb_mean"])
    close(fe["sq_err"], ((full["rgb_mean"].cpu() - gt) ** 2).numpy(), atol=1e-7, rtol=1e-6, what="sq_err")
    close(fe["mse"], cfnerf_amd.img2mse(full["rgb_mean"].cpu(), gt), atol=1e-7, rtol=1e-5, what="mse")
    r0, r1 = E.row_shard(H, 1, 3)
    fs = E.render_uncertainty(H, W, focal, c2w, model, gt=gt[r0:r1], rows=(r0, r1), **kw)
    assert torch.equal(fs["sq_err"], fe["sq_err"][r0:r1])
    # and against the CPU oracle
    ea, er = net.sample_alpha.clone(), net.sample_rgb.clone()
    ea[-1] = 0
    er[-1] = 0
    o = O.render(p, H, W, focal, cfg, ea, er, False, c2w=c2w, ndc=ndc, near=near, far=far, white_bkgd=white_bkgd)
    close(full["rgb_map"], o["rgb_map"], what="rgb_map vs oracle")


def test_full_size_config5_row_tiling_and_determinism():
    """configs[4] at full size (800 x 800 image, K = 32, W = 256, white background, no NDC), where the oracle is far too
    slow to be the checker: the image rendered as 8 row tiles ("8 ranks", no exchange) equals the image rendered in one
    launch bit for bit, rendering twice gives identical bits, and the fused statistics are finite and in range."""
    K = 32
    cfg = O.OracleCfg(netwidth=256, K_samples=K)
    _, _, _, model, _, _ = build_model(cfg, 9, white_bkgd=True, no_ndc=True)
    H = W = 800
    focal = 1111.1
    th, ph = np.deg2rad(30.0), np.deg2rad(-30.0)
    c2w = torch.tensor([[np.cos(th), -np.sin(th) * np.sin(ph), np.sin(th) * np.cos(ph), 4 * np.sin(th) * np.cos(ph)],
                        [0, np.cos(ph), np.sin(ph), 4 * np.sin(ph)],
                        [-np.sin(th), -np.cos(th) * np.sin(ph), np.cos(th) * np.cos(ph), 4 * np.cos(th) * np.cos(ph)]],
                       dtype=torch.float32)
    kw = dict(near=2.0, far=6.0, ndc=False, white_bkgd=True)
    full = E.render_uncertainty(H, W, focal, c2w, model, **kw)
    again = E.render_uncertainty(H, W, focal, c2w, model, **kw)
    for k in ("rgb_mean", "rgb_unc", "disp_mean", "depth_mean"):
        assert torch.equal(full[k], again[k]), k
        assert torch.isfinite(full[k]).all(), k
    parts = [E.render_uncertainty(H, W, focal, c2w, model, rows=E.row_shard(H, rk, 8), **kw) for rk in range(8)]
    for k in ("rgb_mean", "rgb_unc", "disp_mean", "depth_mean"):
        assert torch.equal(torch.cat([q[k] for q in parts], 0), full[k]), k
    assert list(full["rgb_mean"].shape) == [H, W, 3]
    assert float(full["rgb_mean"].min()) >= -1e-5 and float(full["rgb_mean"].max()) <= 1 + 1e-5     # sigmoid colours + white background
    assert float(full["rgb_unc"].min()) >= 0 and float(full["depth_mean"].min()) >= 0
