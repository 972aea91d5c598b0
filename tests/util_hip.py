"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import argparse

import numpy as np
import torch

import cfnerf_amd
from oracle import cfnerf_oracle as O

# fp32 tolerance of the path: the reference's own fp32-vs-fp64 drift is 1e-7..1e-6,
# the HIP path sums in a different order (MFMA k-chunks, wave scans) and uses device libm.
ATOL, RTOL = 1e-5, 1e-4
ATOL_DISP = 1e-4          # disp = 1/(depth/acc): amplifies the relative error of two sums


def close(a, b, atol=ATOL, rtol=RTOL, what=""):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), what + ": non-finite output"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} out of tolerance, max err {err.max():.3e} at "
                           f"{np.unravel_index(err.argmax(), err.shape)} (ref {b.flat[err.argmax()]:.4e})")


def make_args(cfg: O.OracleCfg, **over):
    a = argparse.Namespace(
        multires=cfg.multires, multires_views=cfg.multires_views, i_embed=0, use_viewdirs=True, N_importance=0,
        netdepth=cfg.netdepth, netwidth=cfg.netwidth, K_samples=cfg.K_samples, h_alpha_size=cfg.h_alpha_size,
        h_rgb_size=cfg.h_rgb_size, z_size=4, n_flows=cfg.n_flows, type_flows="triangular", n_hidden=128,
        netchunk_per_gpu=1024 * 64, n_gpus=1, lrate=5e-4, ft_path=None, basedir="/tmp/cfnerf_logs", dataname="d",
        expname="e", no_reload=True, index_step=-1, is_train=True, uniformsample=False, perturb=1.0, N_samples=128,
        white_bkgd=False, raw_noise_std=0.0, dataset_type="llff", no_ndc=False, lindisp=False,
        device=torch.device("cuda"))
    for k, v in over.items():
        setattr(a, k, v)
    return a


def build_model(cfg: O.OracleCfg, seed: int, **over):
    """create_nerf() on the HIP path with the oracle's deterministic weights loaded through load_state_dict."""
    args = make_args(cfg, **over)
    kw_train, kw_test, start, grad_vars, optimizer = cfnerf_amd.create_nerf(args)
    p = O.make_params(cfg, seed)
    model = kw_train["network_fn"]
    sd = model.state_dict()
    for k, v in p.items():
        assert "module." + k in sd, k
        sd["module." + k] = v
    model.load_state_dict(sd)
    return args, kw_train, kw_test, model, p, optimizer


def fern_rays(rng, n, H=378, W=504, focal=407.5658):
    c2w = np.eye(4, dtype=np.float32)[:3]
    c2w[:, 3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    pix = rng.choice(H * W, size=n, replace=False)
    j, i = np.divmod(pix, W)
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i, dtype=np.float64)], -1)
    rays_d = (dirs[:, None, :] * c2w[:3, :3]).sum(-1).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape).astype(np.float32)
    return torch.tensor(np.stack([rays_o, rays_d], 0)), (H, W, focal)
