"""Procedural PSNR scene: the SYNTHETIC STAND-IN for LLFF-fern (no dataset exists in this environment).

Analytic emissive-absorbing Gaussian blobs rendered by dense fp64 quadrature (torch, data generation only) into 20 views
on a sphere, 17 train / 3 held out like llffhold=8 (RUN:742-748).  The rays of the ground-truth renderer come from the
product's own ray kernel (cfnerf_rays_setup, get_rays formula HLP:288-297).  Used by bench.py's `psnr` block and by
tests/tools/psnr_*.py."""
import ctypes as C

import numpy as np
import torch

H, W, FOCAL, NEAR, FAR = 60, 80, 90.0, 2.0, 6.0
I_TEST = (2, 10, 18)
LABEL = "procedural gaussian blobs (synthetic stand-in for LLFF-fern)"


def pose_spherical(theta_deg, phi_deg, radius):
    """load_blender.py:29-34"""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    trans = np.eye(4); trans[2, 3] = radius
    rot_phi = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]])
    rot_th = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    c2w = rot_th @ rot_phi @ trans
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w
    return torch.tensor(c2w[:3, :4], dtype=torch.float32)


def blobs(rng, dev, n_blobs=6):
    return dict(c=torch.tensor(rng.uniform(-0.8, 0.8, (n_blobs, 3)), dtype=torch.float64, device=dev),
                s=torch.tensor(rng.uniform(0.25, 0.45, n_blobs), dtype=torch.float64, device=dev),
                a=torch.tensor(rng.uniform(3.0, 8.0, n_blobs), dtype=torch.float64, device=dev),
                col=torch.tensor(rng.uniform(0.1, 1.0, (n_blobs, 3)), dtype=torch.float64, device=dev))


def camera_rays(c2w, dev):
    """rays_o, rays_d [H*W,3] of the view, from the product's ray set-up kernel (no NDC)."""
    from cfnerf_amd import _lib as L
    packed = torch.empty(H * W, 11, device=dev)
    pose = torch.as_tensor(c2w, dtype=torch.float32).cpu()[:3, :4].contiguous()
    L.check(L.lib().cfnerf_rays_setup(H, W, FOCAL, C.cast(pose.data_ptr(), C.POINTER(C.c_float)), None, None, H * W, 0, 0, NEAR, FAR,
                                      L.ptr(packed), L.stream()), "cfnerf_rays_setup")
    return packed[:, 0:3].contiguous(), packed[:, 3:6].contiguous()


@torch.no_grad()
def render_truth(sc, c2w, dev, n_quad=512):
    ro, rd = camera_rays(c2w, dev)
    ro, rd = ro.double(), rd.double()
    t = torch.linspace(NEAR, FAR, n_quad, dtype=torch.float64, device=dev)
    pts = ro[:, None, :] + rd[:, None, :] * t[None, :, None]                        # [R,Q,3]
    d2 = ((pts[:, :, None, :] - sc["c"][None, None]) ** 2).sum(-1)                  # [R,Q,B]
    dens = sc["a"] * torch.exp(-d2 / (2 * sc["s"] ** 2))
    sigma = dens.sum(-1)
    col = (dens[..., None] * sc["col"]).sum(-2) / (sigma[..., None] + 1e-12)
    delta = (t[1] - t[0]) * rd.norm(dim=-1, keepdim=True)
    alpha = 1 - torch.exp(-sigma * delta)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    return ((alpha * T)[..., None] * col).sum(1).reshape(H, W, 3).float()


_MADE = {}


def make(dev="cuda", seed=7):
    """(poses [20,3,4] cpu, images [20,H,W,3] cpu, i_train, i_test); rendered once per (device, seed) and process"""
    key = (str(dev), int(seed))
    if key not in _MADE:
        _MADE[key] = _make(dev, seed)
    poses, images, i_train, i_test = _MADE[key]
    return poses.clone(), images.clone(), list(i_train), list(i_test)


def _make(dev, seed):
    rng = np.random.default_rng(seed)
    sc = blobs(rng, dev)
    poses = [pose_spherical(th, -20.0 - 10.0 * (i % 3), 4.0) for i, th in enumerate(np.linspace(-60, 60, 20))]
    images = torch.stack([render_truth(sc, p, dev) for p in poses]).cpu()
    i_train = [i for i in range(20) if i not in I_TEST]
    return torch.stack(poses), images, i_train, list(I_TEST)


def held_out_psnr(net, poses, images, idx, dev="cuda"):
    """mean over the held-out views of mse2psnr(img2mse(K-mean prediction, ground truth)) (RUN:1027-1029 on a full image)"""
    from cfnerf_amd import evaluate as E
    ps = []
    for v in idx:
        out = E.render_uncertainty(H, W, FOCAL, poses[v], net, near=NEAR, far=FAR, ndc=False)
        ps.append(float(-10 * torch.log10(torch.mean((out["rgb_mean"] - images[v].to(dev)) ** 2))))
    return float(np.mean(ps))
