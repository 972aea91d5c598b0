#!/usr/bin/env python3
"""Generate golden vectors by importing the REAL reference (build container only).

Run:  python tests/golden/make_golden.py [--out DIR] [fixture names ...]
      (writes DIR/*.npz, default tests/golden/, and DIR/MANIFEST.json: per fixture, the sha256 of every array.
       tests/test_oracle_golden.py::test_fixtures_match_the_manifest holds the committed files to that manifest and,
       where /root/reference exists, test_committed_generator_reproduces_the_fixtures re-runs this script and
       compares every array bit for bit.)

The reference (``/root/reference``, read-only, Python) cannot travel to the GPU
box, so its outputs are captured here as small fixtures.  A fixture holds only
*data*: seeds, inputs and the reference's outputs - never reference source.
Weights come from the build's own deterministic generator
(``oracle.cfnerf_oracle.make_params``) and are loaded into the reference module
with ``load_state_dict``, so fixtures hold seeds instead of weights.

Third-party modules the hot path never touches are stubbed:
cv2, imageio, skimage.metrics, torch.utils.tensorboard, kornia, configargparse.
Randomness is made explicit by temporarily replacing the two generator calls
the hot path makes (``torch.rand`` RUN:524, ``Tensor.normal_`` MOD:234,246)
with functions that hand back pre-chosen tensors; the reference's arithmetic is
untouched.
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import torch  # noqa: E402

from oracle import cfnerf_oracle as O  # noqa: E402


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod("cv2")
    mod("imageio")
    sk = mod("skimage")
    sk.metrics = mod("skimage.metrics", structural_similarity=lambda *a, **k: 0.0)
    mod("kornia", create_meshgrid=lambda *a, **k: None)
    mod("configargparse")
    tb = mod("torch.utils.tensorboard", SummaryWriter=object)
    torch.utils.tensorboard = tb


def import_reference():
    _stub_modules()
    import run_nerf_uncertainty_NF as R   # noqa
    torch.autograd.set_detect_anomaly(False)   # MOD:5 / HLP:2 turn it on at import
    return R


def ref_args(cfg: O.OracleCfg, tmpdir: str, **over):
    a = argparse.Namespace(
        multires=cfg.multires, multires_views=cfg.multires_views, i_embed=0, use_viewdirs=True,
        N_importance=0, netdepth=cfg.netdepth, netwidth=cfg.netwidth, K_samples=cfg.K_samples,
        h_alpha_size=cfg.h_alpha_size, h_rgb_size=cfg.h_rgb_size, z_size=4, n_flows=cfg.n_flows,
        type_flows="triangular", n_hidden=128, netchunk_per_gpu=1024 * 64, n_gpus=1, lrate=5e-4,
        ft_path=None, basedir=tmpdir, dataname="d", expname="e", no_reload=True, index_step=-1,
        is_train=True, uniformsample=False, perturb=1.0, N_samples=128, white_bkgd=False,
        raw_noise_std=0.0, dataset_type="llff", no_ndc=False, lindisp=False)
    for k, v in over.items():
        setattr(a, k, v)
    os.makedirs(os.path.join(tmpdir, "d", "triangular", "e"), exist_ok=True)
    return a


class ExplicitRandom:
    """Feed chosen tensors to the reference's torch.rand / Tensor.normal_ call sites."""

    def __init__(self, t_rand=None, normals=()):
        self.t_rand = t_rand
        self.normals = list(normals)
        self._orig_rand = torch.rand
        self._orig_normal = torch.Tensor.normal_

    def __enter__(self):
        outer = self

        def rand(*shape, **kw):
            shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            assert outer.t_rand is not None and tuple(outer.t_rand.shape) == shp, (shp,)
            return outer.t_rand.clone()

        def normal_(self_t, *a, **k):
            v = outer.normals.pop(0)
            assert tuple(v.shape) == tuple(self_t.shape), (v.shape, self_t.shape)
            self_t.copy_(v)
            return self_t
        torch.rand = rand
        torch.Tensor.normal_ = normal_
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig_rand
        torch.Tensor.normal_ = self._orig_normal


def build_reference_model(R, cfg, seed, tmpdir, **over):
    args = ref_args(cfg, tmpdir, **over)
    kw_train, kw_test, start, grad_vars, optimizer = R.create_nerf(args)
    model = kw_train["network_fn"]                       # nn.DataParallel(NeRF_Flows)
    p = O.make_params(cfg, seed)
    sd = model.state_dict()
    for k, v in p.items():
        assert ("module." + k) in sd, k
        sd["module." + k] = v.clone()
    model.load_state_dict(sd)
    return args, kw_train, kw_test, model, p, optimizer


def reference_kde_nll(rgbs, target, nk, eps=1e-05):
    """The reference's KDE negative log-likelihood, lines RUN:1032-1042 executed on reference tensors (``train()`` cannot be imported
    as a function, so its loss lines are restated HERE, once, operation for operation - the fixtures pin the result bit for bit):
    bandwidth H = unbiased std_K * n/(n-1) * (0.8/n)^(-1/7) + eps (detached), p = mean_K N(target; c_k, H^2) + eps, nll = -mean log p."""
    import math
    rgb_std = torch.std(rgbs, -1) * nk / (nk - 1)
    H_sqrt = (rgb_std.detach() * torch.pow(0.8 / nk, torch.tensor(-1 / 7)) + eps)[..., None]
    r_P_C_1 = torch.exp(-((rgbs - target[..., None]) ** 2) / (2 * H_sqrt * H_sqrt))
    r_P_C_2 = torch.pow(torch.tensor(2 * math.pi), -1.5) / H_sqrt
    return -torch.log((r_P_C_1 * r_P_C_2).mean(-1) + eps).mean()


def fern_rays(rng, n, H=378, W=504, focal=407.5658):
    """Fern-shaped synthetic rays: identity-ish pose, random pixels."""
    c2w = np.eye(4, dtype=np.float32)[:3]
    c2w[:, 3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    pix = rng.choice(H * W, size=n, replace=False)
    j, i = np.divmod(pix, W)
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i, dtype=np.float64)], -1)
    rays_d = (dirs[:, None, :] * c2w[:3, :3]).sum(-1).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape).astype(np.float32)
    return np.stack([rays_o, rays_d], 0), (H, W, focal)


G19 = dict(H=24, W=32, focal=36.0, near=2.0, far=6.0, n_views=20, i_test=(2, 10, 18), n_rand=128, n_steps=120, every=40,
           netwidth=64, K=4, seed=65, scene_seed=1190, step_seed=1191, lrate=5e-4, lrate_decay=250, beta1=0.01)


def g19_scene(seed, H, W, focal, near, far, n_views, n_blobs=6, n_quad=256):
    """Analytic emissive-absorbing Gaussian blobs rendered by dense fp64 quadrature into `n_views` views on a sphere (the small
    CPU twin of tools/procedural_scene.py: same construction, numpy / torch-CPU only)."""
    rng = np.random.default_rng(seed)
    c = torch.tensor(rng.uniform(-0.8, 0.8, (n_blobs, 3)))
    sg = torch.tensor(rng.uniform(0.25, 0.45, n_blobs))
    amp = torch.tensor(rng.uniform(3.0, 8.0, n_blobs))
    col = torch.tensor(rng.uniform(0.1, 1.0, (n_blobs, 3)))
    poses, images = [], []
    for i, th in enumerate(np.linspace(-60, 60, n_views)):
        th_r, ph_r = np.deg2rad(th), np.deg2rad(-20.0 - 10.0 * (i % 3))
        trans = np.eye(4); trans[2, 3] = 4.0
        rot_phi = np.array([[1, 0, 0, 0], [0, np.cos(ph_r), -np.sin(ph_r), 0], [0, np.sin(ph_r), np.cos(ph_r), 0], [0, 0, 0, 1]])
        rot_th = np.array([[np.cos(th_r), 0, -np.sin(th_r), 0], [0, 1, 0, 0], [np.sin(th_r), 0, np.cos(th_r), 0], [0, 0, 0, 1]])
        c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ (rot_th @ rot_phi @ trans)
        pose = torch.tensor(c2w[:3, :4], dtype=torch.float32)
        ro, rd = O.get_rays(H, W, focal, pose)
        ro, rd = ro.reshape(-1, 3).double(), rd.reshape(-1, 3).double()
        t = torch.linspace(near, far, n_quad, dtype=torch.float64)
        pts = ro[:, None, :] + rd[:, None, :] * t[None, :, None]
        d2 = ((pts[:, :, None, :] - c[None, None]) ** 2).sum(-1)
        dens = amp * torch.exp(-d2 / (2 * sg ** 2))
        sigma = dens.sum(-1)
        colr = (dens[..., None] * col).sum(-2) / (sigma[..., None] + 1e-12)
        alpha = 1 - torch.exp(-sigma * (t[1] - t[0]) * rd.norm(dim=-1, keepdim=True))
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
        images.append(((alpha * T)[..., None] * colr).sum(1).reshape(H, W, 3).float())
        poses.append(pose)
    return torch.stack(poses), torch.stack(images)


def g19_draws(step_rng, n_pool, n, K):
    """The draws of ONE step, in this order (the tests re-derive them from G19['step_seed'])."""
    sel = step_rng.integers(0, n_pool, n)
    t_rand = torch.tensor(step_rng.uniform(0, 1, (n, 128)), dtype=torch.float32)
    ea = torch.tensor(step_rng.standard_normal((K, 1)), dtype=torch.float32)
    er = torch.tensor(step_rng.standard_normal((K, 3)), dtype=torch.float32)
    return sel, t_rand, ea, er


def psnr_curve_fixture(R, tmp):
    import math
    c = G19
    H, W, focal, near, far, K, n = c["H"], c["W"], c["focal"], c["near"], c["far"], c["K"], c["n_rand"]
    poses, images = g19_scene(c["scene_seed"], H, W, focal, near, far, c["n_views"])
    i_test = list(c["i_test"])
    i_train = [i for i in range(c["n_views"]) if i not in i_test]
    cfg = O.OracleCfg(netwidth=c["netwidth"], K_samples=K)
    args, kw_train, kw_test, model, p, optimizer = build_reference_model(R, cfg, c["seed"], tmp, K_samples=K, no_ndc=True)
    args.lrate_decay = c["lrate_decay"]
    for kw in (kw_train, kw_test):                               # RUN:828-832 (bds_dict); create_nerf already set ndc=False (RUN:397-400)
        kw.update(near=near, far=far)
    net = model.module
    erng = np.random.default_rng(c["step_seed"] + 1)            # the eval latents (MOD:50-55 draws them at construction)
    ea_eval = torch.tensor(erng.standard_normal((K, 1)), dtype=torch.float32)
    er_eval = torch.tensor(erng.standard_normal((K, 3)), dtype=torch.float32)
    net.sample_alpha, net.sample_rgb = ea_eval.clone(), er_eval.clone()
    ro_all, rd_all, tg_all = [], [], []
    for v in i_train:
        ro, rd = O.get_rays(H, W, focal, poses[v])
        ro_all.append(ro.reshape(-1, 3)); rd_all.append(rd.reshape(-1, 3)); tg_all.append(images[v].reshape(-1, 3))
    ro_all, rd_all, tg_all = torch.cat(ro_all), torch.cat(rd_all), torch.cat(tg_all)

    def held_out():
        ps = []
        for v in i_test:
            with torch.no_grad():
                rgbs, _, _, _ = R.render(H, W, focal, chunk=8192, c2w=poses[v], **kw_test)
            ps.append(float(-10. * torch.log(torch.mean((rgbs.mean(-1) - images[v]) ** 2)) / torch.log(torch.tensor(10.))))   # HLP:15-16
        return ps

    g = dict(poses=poses, images=images, i_test=np.array(i_test), eps_alpha_eval=ea_eval, eps_rgb_eval=er_eval,
             **{k: (np.array(v) if isinstance(v, tuple) else v) for k, v in c.items() if k != "i_test"})
    loss_c, nll_c, psnr_c, ent_c, test_steps, test_psnr = [], [], [], [], [0], [held_out()]
    srng = np.random.default_rng(c["step_seed"])
    global_step = 0
    sel_sum = 0
    for step in range(c["n_steps"]):
        sel, t_rand, ea, er = g19_draws(srng, ro_all.shape[0], n, K)
        sel_sum += int(sel.sum())
        batch_rays = torch.stack([ro_all[sel], rd_all[sel]], 0)
        target = tg_all[sel]
        with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
            rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=batch_rays, verbose=False, retraw=False, **kw_train)
        rgb_mean = rgbs.mean(-1)                                                            # RUN:1027-1029
        img_loss = torch.mean((rgb_mean - target) ** 2)
        psnr = -10. * torch.log(img_loss) / torch.log(torch.Tensor([10.]))
        nk, eps = K, 1e-05                                                                  # RUN:1030-1050
        loss_nll = reference_kde_nll(rgbs, target, nk)
        loss_entropy = extras["loss_entropy"].mean()
        loss = loss_nll + c["beta1"] * loss_entropy
        optimizer.zero_grad()                                                               # RUN:1065-1067
        loss.backward()
        optimizer.step()
        new_lrate = args.lrate * (0.1 ** (global_step / (args.lrate_decay * 1000)))         # RUN:1073-1077
        for param_group in optimizer.param_groups:
            param_group['lr'] = new_lrate
        global_step += 1
        loss_c.append(float(loss)); nll_c.append(float(loss_nll)); psnr_c.append(float(psnr)); ent_c.append(float(loss_entropy))
        if (step + 1) % c["every"] == 0:
            test_steps.append(step + 1); test_psnr.append(held_out())
    g.update(loss=np.array(loss_c), loss_nll=np.array(nll_c), psnr_train=np.array(psnr_c), entropy=np.array(ent_c),
             test_steps=np.array(test_steps), psnr_test=np.array(test_psnr), sel_checksum=np.array(sel_sum))
    return g


def t2n(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        elif v is None:
            continue
        else:
            out[k] = np.asarray(v)
    return out


def main():
    argv = sys.argv[1:]
    out_dir = HERE
    if "--out" in argv:
        i = argv.index("--out")
        out_dir = argv[i + 1]
        del argv[i:i + 2]
        os.makedirs(out_dir, exist_ok=True)
    only = set(argv)                         # optional: regenerate only the named fixtures
    R = import_reference()
    import run_nerf_helpers as HLP
    tmp = tempfile.mkdtemp(prefix="cfnerf_golden_")
    rng = np.random.default_rng(1234)
    out = {}

    # ---------------- G8: encoder (HLP:54-69) ----------------
    x = torch.tensor(rng.uniform(-1.5, 1.5, (16, 3)), dtype=torch.float32)
    f10, d10 = HLP.get_embedder(10, 0)
    f4, d4 = HLP.get_embedder(4, 0)
    out["g8_encoder"] = dict(x=x, e10=f10(x), e4=f4(x), d10=d10, d4=d4)

    # ---------------- G1/G2/G3: model (small width and full width) ----------------
    for tag, cfg, seed, P in (("w64", O.OracleCfg(netwidth=64, K_samples=4), 11, 48),
                              ("w256", O.OracleCfg(netwidth=256, K_samples=4), 12, 24),
                              ("w64k1", O.OracleCfg(netwidth=64, K_samples=1), 13, 8),
                              ("w128k5", O.OracleCfg(netwidth=128, K_samples=5, h_alpha_size=64, h_rgb_size=64), 14, 16)):
        K = cfg.K_samples
        args, kw_train, kw_test, model, p, _ = build_reference_model(R, cfg, seed, tmp, K_samples=K)
        net = model.module
        x90 = torch.tensor(rng.uniform(-1, 1, (P, 90)), dtype=torch.float32)
        ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
        er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
        # G1 eval branch (MOD:192-223): module buffers, last sample zeroed by the reference itself
        net.sample_alpha = ea.clone()
        net.sample_rgb = er.clone()
        with torch.no_grad():
            raw_eval, aux = net(x90, False, True)
            h_alpha, h_rgb = net.encode(x90)
        # G2 train branch (MOD:225-291) with explicit epsilons
        with ExplicitRandom(normals=[ea, er]):
            raw_train, ent = net(x90, False, False)
        # G3 flow units (MOD:387-416) on 5 rows, direct call with chosen z0/h
        z0a = torch.tensor(rng.standard_normal((5, 1)), dtype=torch.float32)
        ha = torch.tensor(rng.standard_normal((5, cfg.h_alpha_size)), dtype=torch.float32)
        z0r = torch.tensor(rng.standard_normal((5, 3)), dtype=torch.float32)
        hr = torch.tensor(rng.standard_normal((5, cfg.h_rgb_size)), dtype=torch.float32)
        with torch.no_grad():
            za_t, lda_t = net.flows_alpha(z0a, ha, False)
            zr_t, ldr_t = net.flows_rgb(z0r, hr, False)
            za_e, _ = net.flows_alpha(z0a, ha, True)
            zr_e, _ = net.flows_rgb(z0r, hr, True)
            r1, r2, b = net.flows_rgb.encode(hr)
        out[f"g123_model_{tag}"] = dict(
            seed=seed, netwidth=cfg.netwidth, K=K, h_alpha_size=cfg.h_alpha_size, h_rgb_size=cfg.h_rgb_size,
            x90=x90, eps_alpha=ea, eps_rgb=er,
            raw_eval=raw_eval, aux_eval_absmax=aux.abs().max(), h_alpha=h_alpha, h_rgb=h_rgb,
            raw_train=raw_train, loss_entropy=ent.reshape(-1)[0], loss_entropy_shape=np.array(ent.shape),
            z0a=z0a, ha=ha, z0r=z0r, hr=hr, za_train=za_t, lda_train=lda_t, zr_train=zr_t, ldr_train=ldr_t,
            za_eval=za_e, zr_eval=zr_e, r1=r1, r2=r2, b=b)

    # ---------------- G4: composite (RUN:411-454) ----------------
    N, S, K = 8, 128, 4
    raw = torch.tensor(rng.standard_normal((N, S, K, 4)) * 2.0, dtype=torch.float32)
    raw[0, :, :, 3] = 30.0       # all-opaque ray
    raw[1, :, :, 3] = -40.0      # all-empty ray
    raw[2, :, 1, 3] = 25.0       # softplus linear branch (threshold 20) on one k
    near = torch.tensor(rng.uniform(0.0, 0.5, (N, 1)), dtype=torch.float32)
    far = near + torch.tensor(rng.uniform(0.5, 4.0, (N, 1)), dtype=torch.float32)
    tv = O.t_vals_table()
    z = near * (1 - tv) + far * tv
    d = torch.tensor(rng.standard_normal((N, 3)), dtype=torch.float32)
    g4 = dict(raw=raw, z_vals=z, rays_d=d)
    for wb in (False, True):
        with torch.no_grad():
            rgb_map, disp, w, depth = R.raw2outputs(raw, z, d, 0, wb)
        s = "wb" if wb else "nb"
        g4.update({f"rgb_map_{s}": rgb_map, f"disp_{s}": disp, f"weights_{s}": w, f"depth_{s}": depth})
    out["g4_composite"] = g4

    # ---------------- G5: render on a ray batch, G7: train step ----------------
    for tag, cfg, seed, n, over in (
            ("w64_ndc", O.OracleCfg(netwidth=64, K_samples=4), 21, 32, dict()),
            ("w64_nondc_lindisp_wb", O.OracleCfg(netwidth=64, K_samples=3), 22, 16,
             dict(no_ndc=True, lindisp=True, white_bkgd=True)),
            ("w256_ndc", O.OracleCfg(netwidth=256, K_samples=4), 23, 8, dict())):
        K = cfg.K_samples
        args, kw_train, kw_test, model, p, optimizer = build_reference_model(R, cfg, seed, tmp, K_samples=K, **over)
        net = model.module
        rays, (H, W, focal) = fern_rays(rng, n)
        rays_t = torch.tensor(rays)
        near, far = (0., 1.) if not over.get("no_ndc") else (1.2, 8.0)
        t_rand = torch.tensor(rng.uniform(0, 1, (n, 128)), dtype=torch.float32)
        ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
        er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
        target = torch.tensor(rng.uniform(0, 1, (n, 3)), dtype=torch.float32)
        # train-mode render (perturb=1) with explicit randomness
        with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
            rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=near, far=far,
                                                 verbose=False, retraw=False, **kw_train)
        # G7: loss lines RUN:1027-1050 executed verbatim on the reference tensors
        beta1 = 0.01
        rgb_mean = torch.mean(rgbs, -1)
        mse = HLP.img2mse(rgb_mean, target)
        psnr = HLP.mse2psnr(mse)
        import math
        nk = K
        loss_nll = reference_kde_nll(rgbs, target, nk)
        loss_entropy = extras["loss_entropy"].mean()
        loss = loss_nll + beta1 * loss_entropy
        optimizer.zero_grad()
        loss.backward()
        grads = {k[len("module."):]: (v.grad.clone() if v.grad is not None else None)
                 for k, v in model.named_parameters()}
        optimizer.step()
        new_params = {k[len("module."):]: v.detach().clone() for k, v in model.named_parameters()}
        # restore weights for the eval render (Adam moved them)
        sd = model.state_dict()
        for k, v in p.items():
            sd["module." + k] = v.clone()
        model.load_state_dict(sd)
        # eval-mode render (perturb=0, fixed eps, last = 0)
        net.sample_alpha = ea.clone()
        net.sample_rgb = er.clone()
        with torch.no_grad():
            rgbs_e, disp_e, depth_e, extras_e = R.render(H, W, focal, chunk=8192, rays=rays_t, near=near, far=far,
                                                         **kw_test)
        g = dict(seed=seed, netwidth=cfg.netwidth, K=K, H=H, W=W, focal=focal, near=near, far=far,
                 ndc=int(not over.get("no_ndc", False)), lindisp=int(over.get("lindisp", False)),
                 white_bkgd=int(over.get("white_bkgd", False)), beta1=beta1,
                 rays=rays_t, t_rand=t_rand, eps_alpha=ea, eps_rgb=er, target=target,
                 rgb_map=rgbs, disp_map=disp, depth_map=depth, raw=extras["raw"], pts=extras["pts"],
                 loss_entropy=loss_entropy, loss_entropy_numel=extras["loss_entropy"].numel(),
                 loss_nll=loss_nll, loss=loss, mse=mse, psnr=psnr.reshape(-1)[0],
                 rgb_map_eval=rgbs_e, disp_map_eval=disp_e, depth_map_eval=depth_e,
                 eval_extras_keys=np.array(sorted(extras_e.keys())), train_extras_keys=np.array(sorted(extras.keys())))
        dead = []
        for k, v in grads.items():
            if v is None:
                dead.append(k)
            elif cfg.netwidth >= 256 and v.numel() > 4096:
                # keep the full-width fixture small: first two rows + Frobenius norm of the big tensors
                g["gradrows." + k] = v[:2]
                g["gradnorm." + k] = v.double().norm()
            else:
                g["grad." + k] = v
                g["adam1." + k] = new_params[k]
        g["dead_params"] = np.array(sorted(dead))
        out[f"g57_render_{tag}"] = g

    # ---------------- G6: full tiny image via c2w (RUN:129-131, HLP:288-297) ----------------
    cfg = O.OracleCfg(netwidth=64, K_samples=4)
    args, kw_train, kw_test, model, p, _ = build_reference_model(R, cfg, 31, tmp, K_samples=4)
    net = model.module
    ea = torch.tensor(rng.standard_normal((4, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((4, 3)), dtype=torch.float32)
    net.sample_alpha = ea.clone()
    net.sample_rgb = er.clone()
    th = 0.3
    c2w = torch.tensor([[np.cos(th), 0, np.sin(th), 0.1], [0, 1, 0, -0.2], [-np.sin(th), 0, np.cos(th), 0.05]],
                       dtype=torch.float32)
    with torch.no_grad():
        rgbs_e, disp_e, depth_e, _ = R.render(6, 8, 7.0, chunk=8192, c2w=c2w, near=0., far=1., **kw_test)
        ro, rd = HLP.get_rays(6, 8, 7.0, c2w)
        no, nd = HLP.ndc_rays(6, 8, 7.0, 1., ro, rd)
    # c2w_staticcam (RUN:139-141): view directions from c2w, ray geometry from the static camera
    th2 = -0.2
    c2w_static = torch.tensor([[np.cos(th2), 0, np.sin(th2), -0.05], [0, 1, 0, 0.1], [-np.sin(th2), 0, np.cos(th2), 0.2]],
                              dtype=torch.float32)
    with torch.no_grad():
        rgbs_s, disp_s, depth_s, _ = R.render(6, 8, 7.0, chunk=8192, c2w=c2w, c2w_staticcam=c2w_static, near=0., far=1., **kw_test)
    out["g6_render_c2w"] = dict(seed=31, netwidth=64, K=4, H=6, W=8, focal=7.0, c2w=c2w, eps_alpha=ea, eps_rgb=er,
                                rgb_map=rgbs_e, disp_map=disp_e, depth_map=depth_e, rays_o=ro, rays_d=rd,
                                ndc_o=no, ndc_d=nd, c2w_static=c2w_static, rgb_map_static=rgbs_s, disp_map_static=disp_s,
                                depth_map_static=depth_s)

    # ---------------- G9: sparsification curves (HLP:382-438), the AUSE helper ----------------
    var_vec = torch.tensor(rng.uniform(0.0, 1.0, 1000) ** 2, dtype=torch.float32)
    err_vec = torch.tensor((rng.uniform(0.0, 1.0, 1000) * 0.5 + 0.5 * var_vec.numpy()) ** 2, dtype=torch.float32)
    g9 = dict(var_vec=var_vec, err_vec=err_vec)
    import contextlib, io
    for ut in ("c", "v"):
        for et in ("rmse", "mae"):
            with contextlib.redirect_stdout(io.StringIO()):
                a, b = HLP.sparsification_plot(var_vec, err_vec, uncert_type=ut, err_type=et)
            g9[f"oracle_{ut}_{et}"] = a
            g9[f"byvar_{ut}_{et}"] = b
    out["g9_sparsification"] = g9

    # ---------------- G11: seeded construction (RUN:317-331, MOD:38-67): torch.manual_seed -> create_nerf ----------------
    g11 = {}
    for tag, wd, ks, depth in (("w64", 64, 3, 8), ("w256", 256, 4, 8), ("w128d6", 128, 2, 6)):
        cfg = O.OracleCfg(netwidth=wd, K_samples=ks, netdepth=depth)
        args = ref_args(cfg, tmp, K_samples=ks)
        torch.manual_seed(1234)
        kw_train, kw_test, start, grad_vars, optimizer = R.create_nerf(args)
        net = kw_train["network_fn"].module
        g11[f"{tag}.netwidth"], g11[f"{tag}.K"], g11[f"{tag}.netdepth"] = wd, ks, depth
        g11[f"{tag}.sample_alpha"], g11[f"{tag}.sample_rgb"] = net.sample_alpha, net.sample_rgb
        for k, v in net.state_dict().items():
            if v.dtype != torch.float32 or "mask" in k:
                continue
            f = v.detach().reshape(-1)
            g11[f"{tag}.head.{k}"] = f[:8].clone()                      # first values, exact
            g11[f"{tag}.sum.{k}"] = f.double().sum()                    # float64 checksum of the whole tensor
            g11[f"{tag}.sumsq.{k}"] = (f.double() ** 2).sum()
    out["g11_seeded_init"] = g11

    # ---------------- G12: IMPLICIT randomness under torch.manual_seed (RUN:524 -> MOD:234 -> MOD:246) ----------------
    # no patched call sites: the reference draws t_rand, eps_alpha, eps_rgb itself from the seeded CPU generator
    rng12 = np.random.default_rng(112)          # own stream: adding this fixture does not move the others
    cfg = O.OracleCfg(netwidth=64, K_samples=4)
    args, kw_train, kw_test, model, p, _ = build_reference_model(R, cfg, 51, tmp, K_samples=4)
    rays, (H, W, focal) = fern_rays(rng12, 24)
    torch.manual_seed(4321)
    with torch.no_grad():
        rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=torch.tensor(rays), near=0., far=1., **kw_train)
    out["g12_seeded_draws"] = dict(seed=51, torch_seed=4321, netwidth=64, K=4, rays=rays, H=H, W=W, focal=focal,
                                   rgb_map=rgbs, disp_map=disp, depth_map=depth, raw_first4=extras["raw"][:4],
                                   loss_entropy=extras["loss_entropy"].reshape(-1)[0])


    # ================= fixtures added in round 2: each uses its OWN random stream, so the ones above do not move =========
    # ---------------- G13: ODD netdepth (RUN:327 skips = [netdepth / 2] is a float: no layer matches, no skip concat) ----
    rng13 = np.random.default_rng(113)
    cfg = O.OracleCfg(netwidth=64, K_samples=3, netdepth=5)
    args, kw_train, kw_test, model, p, _ = build_reference_model(R, cfg, 61, tmp, K_samples=3)
    net = model.module
    x90 = torch.tensor(rng13.uniform(-1, 1, (12, 90)), dtype=torch.float32)
    ea = torch.tensor(rng13.standard_normal((3, 1)), dtype=torch.float32)
    er = torch.tensor(rng13.standard_normal((3, 3)), dtype=torch.float32)
    net.sample_alpha, net.sample_rgb = ea.clone(), er.clone()
    with torch.no_grad():
        raw_eval, _ = net(x90, False, True)
    with ExplicitRandom(normals=[ea, er]):
        raw_train, ent = net(x90, False, False)
    shapes = {k: np.array(v.shape) for k, v in net.state_dict().items() if k.startswith("pts_linears") and k.endswith("weight")}
    out["g13_odd_depth"] = dict(seed=61, netwidth=64, netdepth=5, K=3, x90=x90, eps_alpha=ea, eps_rgb=er, raw_eval=raw_eval,
                                raw_train=raw_train, loss_entropy=ent.reshape(-1)[0],
                                **{"shape." + k: v for k, v in shapes.items()})

    # ---------------- G14: BASELINE config 1 (C1): K = 1, N_rand = 256, default width, render() forward only (R4) ----------
    rng14 = np.random.default_rng(114)
    cfg = O.OracleCfg(netwidth=256, K_samples=1)
    args, kw_train, kw_test, model, p, _ = build_reference_model(R, cfg, 62, tmp, K_samples=1)
    net = model.module
    rays, (H, W, focal) = fern_rays(rng14, 256)
    rays_t = torch.tensor(rays)
    t_rand = torch.tensor(rng14.uniform(0, 1, (256, 128)), dtype=torch.float32)
    ea = torch.tensor(rng14.standard_normal((1, 1)), dtype=torch.float32)
    er = torch.tensor(rng14.standard_normal((1, 3)), dtype=torch.float32)
    with torch.no_grad(), ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
        rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., **kw_train)
    net.sample_alpha, net.sample_rgb = ea.clone(), er.clone()
    with torch.no_grad():
        rgbs_e, disp_e, depth_e, extras_e = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., **kw_test)
    out["g14_render_c1_k1"] = dict(seed=62, netwidth=256, K=1, H=H, W=W, focal=focal, rays=rays_t, t_rand=t_rand, eps_alpha=ea, eps_rgb=er,
                                   rgb_map=rgbs, disp_map=disp, depth_map=depth, raw_first4=extras["raw"][:4],
                                   loss_entropy=extras["loss_entropy"].reshape(-1)[0],
                                   loss_entropy_shape=np.array(extras["loss_entropy"].shape),
                                   rgb_map_eval=rgbs_e, disp_map_eval=disp_e, depth_map_eval=depth_e)

    # ---------------- G15: THREE optimiser steps with the reference's own loop lines (RUN:1013-1077): Adam state past the
    #                  first step (m / sqrt(v) no longer +-1) and the learning-rate decay written after every step ----------
    rng15 = np.random.default_rng(115)
    cfg = O.OracleCfg(netwidth=64, K_samples=4)
    args, kw_train, kw_test, model, p, optimizer = build_reference_model(R, cfg, 63, tmp, K_samples=4)
    args.lrate_decay = 1                     # decay_steps = 1000: the schedule is visible within three steps
    n = 24
    rays, (H, W, focal) = fern_rays(rng15, n)
    rays_t = torch.tensor(rays)
    target = torch.tensor(rng15.uniform(0, 1, (n, 3)), dtype=torch.float32)
    beta1 = 0.01
    global_step = 0                          # RUN:820 (start = 0)
    g15 = dict(seed=63, netwidth=64, K=4, H=H, W=W, focal=focal, rays=rays_t, target=target, beta1=beta1, lrate=args.lrate,
               lrate_decay=args.lrate_decay, n_steps=3)
    import math
    # tensors whose per-step gradients are kept, so that the fused Adam kernel can be driven with the reference's own
    # gradients (the end-to-end trajectory is chaotic: Adam's first step is lr * sign(g))
    G15_KEYS = ("pts_linears.0.weight", "pts_linears.5.weight", "pts_linears.5.bias", "h_rgb_linear.weight",
                "flows_rgb.amor_d.weight", "flows_alpha.amor_b.bias", "alpha_std", "rgb_mean")
    g15["adam_keys"] = np.array(G15_KEYS)
    for step in range(3):
        t_rand = torch.tensor(rng15.uniform(0, 1, (n, 128)), dtype=torch.float32)
        ea = torch.tensor(rng15.standard_normal((4, 1)), dtype=torch.float32)
        er = torch.tensor(rng15.standard_normal((4, 3)), dtype=torch.float32)
        with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
            rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., verbose=False, retraw=False,
                                                 **kw_train)
        nk, eps = 4, 1e-05                                                                  # RUN:1026-1050
        loss_nll = reference_kde_nll(rgbs, target, nk)
        loss = loss_nll + beta1 * extras["loss_entropy"].mean()
        optimizer.zero_grad()                                                               # RUN:1065-1067
        loss.backward()
        named = {k[len("module."):]: v for k, v in model.named_parameters()}
        for k in G15_KEYS:                                                                  # what optimizer.step() is fed ...
            g15[f"g{step}." + k] = named[k].grad.clone()
            if step == 0:
                g15["p0." + k] = named[k].detach().clone()
        optimizer.step()
        for k in G15_KEYS:                                                                  # ... and what it makes of it
            g15[f"p{step + 1}." + k] = named[k].detach().clone()
        decay_rate = 0.1                                                                    # RUN:1073-1077
        decay_steps = args.lrate_decay * 1000
        new_lrate = args.lrate * (decay_rate ** (global_step / decay_steps))
        for param_group in optimizer.param_groups:
            param_group['lr'] = new_lrate
        global_step += 1                                                                    # RUN:1198
        g15[f"t_rand{step}"], g15[f"eps_alpha{step}"], g15[f"eps_rgb{step}"] = t_rand, ea, er
        g15[f"loss{step}"] = loss.detach()
        g15[f"lr_after{step}"] = new_lrate
    for k, v in model.named_parameters():
        g15["adam3." + k[len("module."):]] = v.detach().clone()
    out["g15_three_steps"] = g15

    # ---------------- G16: train step at K = 16 (BASELINE config 4's latent count; the authors train at K = 32) -------------
    rng16 = np.random.default_rng(116)
    cfg = O.OracleCfg(netwidth=64, K_samples=16)
    args, kw_train, kw_test, model, p, optimizer = build_reference_model(R, cfg, 64, tmp, K_samples=16)
    n = 6
    rays, (H, W, focal) = fern_rays(rng16, n)
    rays_t = torch.tensor(rays)
    target = torch.tensor(rng16.uniform(0, 1, (n, 3)), dtype=torch.float32)
    t_rand = torch.tensor(rng16.uniform(0, 1, (n, 128)), dtype=torch.float32)
    ea = torch.tensor(rng16.standard_normal((16, 1)), dtype=torch.float32)
    er = torch.tensor(rng16.standard_normal((16, 3)), dtype=torch.float32)
    with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
        rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., verbose=False, retraw=False, **kw_train)
    nk, eps, beta1 = 16, 1e-05, 0.01
    loss_nll = reference_kde_nll(rgbs, target, nk)
    loss = loss_nll + beta1 * extras["loss_entropy"].mean()
    optimizer.zero_grad()
    loss.backward()
    g16 = dict(seed=64, netwidth=64, K=16, H=H, W=W, focal=focal, rays=rays_t, target=target, t_rand=t_rand, eps_alpha=ea, eps_rgb=er,
               beta1=beta1, rgb_map=rgbs, depth_map=depth, loss=loss.detach(), loss_nll=loss_nll.detach(),
               loss_entropy=extras["loss_entropy"].mean().detach())
    for k, v in model.named_parameters():
        if v.grad is not None:
            g16["grad." + k[len("module."):]] = v.grad.clone()
    out["g16_train_k16"] = g16

    # ---------------- G17: a train step OUTSIDE the shipped configurations - netwidth 192, netdepth 6, n_flows 3, h_alpha 96,
    #                  h_rgb 96 (RUN:582-622 take any value): pins the widened configuration space of the build to the real reference ---
    rng17 = np.random.default_rng(117)
    cfg = O.OracleCfg(netwidth=192, netdepth=6, K_samples=3, n_flows=3, h_alpha_size=96, h_rgb_size=96)
    args, kw_train, kw_test, model, p, optimizer = build_reference_model(R, cfg, 65, tmp, K_samples=3)
    net = model.module
    n = 10
    rays, (H, W, focal) = fern_rays(rng17, n)
    rays_t = torch.tensor(rays)
    target = torch.tensor(rng17.uniform(0, 1, (n, 3)), dtype=torch.float32)
    t_rand = torch.tensor(rng17.uniform(0, 1, (n, 128)), dtype=torch.float32)
    ea = torch.tensor(rng17.standard_normal((3, 1)), dtype=torch.float32)
    er = torch.tensor(rng17.standard_normal((3, 3)), dtype=torch.float32)
    with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
        rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., verbose=False, retraw=False, **kw_train)
    nk, eps, beta1 = 3, 1e-05, 0.02
    loss_nll = reference_kde_nll(rgbs, target, nk)
    loss = loss_nll + beta1 * extras["loss_entropy"].mean()
    optimizer.zero_grad()
    loss.backward()
    net.sample_alpha, net.sample_rgb = ea.clone(), er.clone()
    with torch.no_grad():
        rgbs_e, disp_e, depth_e, _ = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., **kw_test)
    g17 = dict(seed=65, netwidth=192, netdepth=6, K=3, n_flows=3, h_alpha_size=96, h_rgb_size=96, H=H, W=W, focal=focal, rays=rays_t,
               target=target, t_rand=t_rand, eps_alpha=ea, eps_rgb=er, beta1=beta1, rgb_map=rgbs, depth_map=depth, disp_map=disp,
               raw_first2=extras["raw"][:2], loss=loss.detach(), loss_nll=loss_nll.detach(), loss_entropy=extras["loss_entropy"].mean().detach(),
               rgb_map_eval=rgbs_e, depth_map_eval=depth_e,
               state_dict_keys=np.array([k[len("module."):] for k in model.state_dict().keys()]))
    for k, v in model.named_parameters():
        k = k[len("module."):]
        if v.grad is None:
            continue
        if v.grad.numel() > 8192:            # the big trunk weights: first two rows + Frobenius norm keep the fixture small
            g17["gradrows." + k] = v.grad[:2].clone()
            g17["gradnorm." + k] = v.grad.double().norm()
        else:
            g17["grad." + k] = v.grad.clone()
    out["g17_train_wide_config"] = g17

    # ---------------- G18: train step at K = 100 latent samples (above the reference's default of 64, RUN:631; the build accepts up to
    #                  128 and runs the flow phase on the hardware transcendentals from K = 16 on) --------------------------------------
    rng18 = np.random.default_rng(118)
    cfg = O.OracleCfg(netwidth=64, K_samples=100)
    args, kw_train, kw_test, model, p, optimizer = build_reference_model(R, cfg, 66, tmp, K_samples=100)
    n = 4
    rays, (H, W, focal) = fern_rays(rng18, n)
    rays_t = torch.tensor(rays)
    target = torch.tensor(rng18.uniform(0, 1, (n, 3)), dtype=torch.float32)
    t_rand = torch.tensor(rng18.uniform(0, 1, (n, 128)), dtype=torch.float32)
    ea = torch.tensor(rng18.standard_normal((100, 1)), dtype=torch.float32)
    er = torch.tensor(rng18.standard_normal((100, 3)), dtype=torch.float32)
    with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
        rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., verbose=False, retraw=False, **kw_train)
    nk, eps, beta1 = 100, 1e-05, 0.01
    loss_nll = reference_kde_nll(rgbs, target, nk)
    loss = loss_nll + beta1 * extras["loss_entropy"].mean()
    optimizer.zero_grad()
    loss.backward()
    g18 = dict(seed=66, netwidth=64, K=100, H=H, W=W, focal=focal, rays=rays_t, target=target, t_rand=t_rand, eps_alpha=ea, eps_rgb=er,
               beta1=beta1, rgb_map=rgbs, depth_map=depth, raw_first1=extras["raw"][:1], loss=loss.detach(), loss_nll=loss_nll.detach(),
               loss_entropy=extras["loss_entropy"].mean().detach())
    for k, v in model.named_parameters():
        if v.grad is not None:
            g18["grad." + k[len("module."):]] = v.grad.clone()
    out["g18_train_k100"] = g18

    # ---------------- G19: PSNR-vs-step of the REFERENCE ITSELF, trained here for 120 steps with its own loop lines (RUN:1013-1077) on a
    #                  tiny procedural scene ("train both reference (CPU, few hundred steps, only here) and build, compare
    #                  PSNR-vs-step curves").  The fixture holds the scene (poses, images), the seeds of every draw, and the reference's
    #                  per-step loss / train-batch PSNR (RUN:1027-1029) and held-out PSNR of the K-mean prediction at four checkpoints.
    g19 = psnr_curve_fixture(R, tmp)
    out["g19_psnr_curve"] = g19

    # ================= fixtures added in round 5 (own random streams) ==========================================================
    # ---------------- G20: NON-DEFAULT positional encodings, --multires 6 --multires_views 2 (RUN:641-644, HLP:54-69): 39 + 15 input
    #                  channels change the K dimension of layer 0, the skip block and the view block and the in-kernel encoder -------------
    rng20 = np.random.default_rng(120)
    cfg = O.OracleCfg(netwidth=64, K_samples=3, multires=6, multires_views=2)
    args, kw_train, kw_test, model, p, optimizer = build_reference_model(R, cfg, 67, tmp, K_samples=3)
    net = model.module
    n = 10
    rays, (H, W, focal) = fern_rays(rng20, n)
    rays_t = torch.tensor(rays)
    target = torch.tensor(rng20.uniform(0, 1, (n, 3)), dtype=torch.float32)
    t_rand = torch.tensor(rng20.uniform(0, 1, (n, 128)), dtype=torch.float32)
    ea = torch.tensor(rng20.standard_normal((3, 1)), dtype=torch.float32)
    er = torch.tensor(rng20.standard_normal((3, 3)), dtype=torch.float32)
    with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
        rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., verbose=False, retraw=False, **kw_train)
    nk, eps, beta1 = 3, 1e-05, 0.02
    loss_nll = reference_kde_nll(rgbs, target, nk)
    loss = loss_nll + beta1 * extras["loss_entropy"].mean()
    optimizer.zero_grad()
    loss.backward()
    net.sample_alpha, net.sample_rgb = ea.clone(), er.clone()
    with torch.no_grad():
        rgbs_e, disp_e, depth_e, _ = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., **kw_test)
    x54 = torch.tensor(rng20.uniform(-1, 1, (12, cfg.input_ch + cfg.input_ch_views)), dtype=torch.float32)
    with ExplicitRandom(normals=[ea, er]):
        raw_x, ent_x = net(x54, False, False)
    g20 = dict(seed=67, netwidth=64, K=3, multires=6, multires_views=2, input_ch=args.input_ch, input_ch_views=args.input_ch_views,
               H=H, W=W, focal=focal, rays=rays_t, target=target, t_rand=t_rand, eps_alpha=ea, eps_rgb=er, beta1=beta1, rgb_map=rgbs,
               depth_map=depth, disp_map=disp, raw_first2=extras["raw"][:2], loss=loss.detach(), loss_nll=loss_nll.detach(),
               loss_entropy=extras["loss_entropy"].mean().detach(), rgb_map_eval=rgbs_e, depth_map_eval=depth_e,
               x54=x54, raw_x=raw_x.detach(), loss_entropy_x=ent_x.reshape(-1)[0].detach(),
               **{"shape." + k[len("module."):]: np.array(v.shape) for k, v in model.state_dict().items()
                  if k.endswith("pts_linears.0.weight") or k.endswith("pts_linears.5.weight") or k.endswith("views_linears.0.weight")})
    for k, v in model.named_parameters():
        if v.grad is not None:
            g20["grad." + k[len("module."):]] = v.grad.clone()
    out["g20_train_multires_6_2"] = g20

    # ---------------- G21: the AUTHORS' RECIPE (train_NF.sh:1-19 + configs/africa_ds.txt): netwidth 512, h_alpha 64, h_rgb 64, K 32, no NDC.
    #                  The build's W = 512 kernels use a different register / argument scheme from W <= 256; this pins them to the real
    #                  reference: outputs, loss, and of every gradient its Frobenius norm, float64 sum and 64 sampled entries (the
    #                  2.36 M-entry gradient itself would be 9 MB), plus one Adam step on the same samples --------------------------------
    rng21 = np.random.default_rng(121)
    cfg = O.OracleCfg(netwidth=512, K_samples=32, h_alpha_size=64, h_rgb_size=64)
    args, kw_train, kw_test, model, p, optimizer = build_reference_model(R, cfg, 68, tmp, K_samples=32, no_ndc=True)
    net = model.module
    n = 4
    rays, (H, W, focal) = fern_rays(rng21, n)
    rays_t = torch.tensor(rays)
    near, far = 1.2, 8.0
    target = torch.tensor(rng21.uniform(0, 1, (n, 3)), dtype=torch.float32)
    t_rand = torch.tensor(rng21.uniform(0, 1, (n, 128)), dtype=torch.float32)
    ea = torch.tensor(rng21.standard_normal((32, 1)), dtype=torch.float32)
    er = torch.tensor(rng21.standard_normal((32, 3)), dtype=torch.float32)
    with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
        rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=near, far=far, verbose=False, retraw=False, **kw_train)
    nk, eps, beta1 = 32, 1e-05, 0.01
    loss_nll = reference_kde_nll(rgbs, target, nk)
    loss = loss_nll + beta1 * extras["loss_entropy"].mean()
    optimizer.zero_grad()
    loss.backward()
    g21 = dict(seed=68, netwidth=512, K=32, h_alpha_size=64, h_rgb_size=64, H=H, W=W, focal=focal, near=near, far=far, ndc=0, rays=rays_t,
               target=target, t_rand=t_rand, eps_alpha=ea, eps_rgb=er, beta1=beta1, rgb_map=rgbs, depth_map=depth, disp_map=disp,
               raw_first1=extras["raw"][:1], loss=loss.detach(), loss_nll=loss_nll.detach(), loss_entropy=extras["loss_entropy"].mean().detach())
    named = {k[len("module."):]: v for k, v in model.named_parameters()}
    samples = {}
    for k, v in named.items():
        if v.grad is None:
            continue
        gf = v.grad.reshape(-1)
        idx = np.sort(rng21.choice(gf.numel(), size=min(64, gf.numel()), replace=False))
        samples[k] = torch.tensor(idx)
        g21["gradidx." + k] = idx
        g21["gradsample." + k] = gf[samples[k]].clone()
        g21["gradnorm." + k] = gf.double().norm()
        g21["gradsum." + k] = gf.double().sum()
        g21["gradabsmax." + k] = gf.abs().max()
    optimizer.step()
    for k, idx in samples.items():
        g21["adam1sample." + k] = named[k].detach().reshape(-1)[idx].clone()
    sd = model.state_dict()
    for k, v in p.items():
        sd["module." + k] = v.clone()
    model.load_state_dict(sd)
    net.sample_alpha, net.sample_rgb = ea.clone(), er.clone()
    with torch.no_grad():
        rgbs_e, disp_e, depth_e, _ = R.render(H, W, focal, chunk=8192, rays=rays_t, near=near, far=far, **kw_test)
    g21.update(rgb_map_eval=rgbs_e, depth_map_eval=depth_e)
    out["g21_train_authors_recipe_w512"] = g21


    # ---------------- G22: a CHECKPOINT THE REFERENCE WROTE (RUN:1085-1100), to be loaded through its loader's rules (RUN:345-378):
    #                  the real NeRF_Flows under nn.DataParallel, constructed by create_nerf under a seed, two iterations of the loop lines so
    #                  that torch.optim.Adam carries state, then the reference's own torch.save of {global_step, network_fn_state_dict,
    #                  optimizer_state_dict} -> tests/golden/g22_reference_ckpt.tar (a pickled dict of tensors: data), and what the reference
    #                  renders from that model in eval mode ----------------
    rng22 = np.random.default_rng(122)
    cfg = O.OracleCfg(netwidth=64, netdepth=4, K_samples=3)
    args = ref_args(cfg, tmp, K_samples=3)
    torch.manual_seed(4321)
    kw_train, kw_test, start, grad_vars, optimizer = R.create_nerf(args)
    model = kw_train["network_fn"]
    net = model.module
    n = 16
    rays, (H, W, focal) = fern_rays(rng22, n)
    rays_t = torch.tensor(rays)
    target = torch.tensor(rng22.uniform(0, 1, (n, 3)), dtype=torch.float32)
    global_step = 0
    for step in range(2):
        t_rand = torch.tensor(rng22.uniform(0, 1, (n, 128)), dtype=torch.float32)
        ea = torch.tensor(rng22.standard_normal((3, 1)), dtype=torch.float32)
        er = torch.tensor(rng22.standard_normal((3, 3)), dtype=torch.float32)
        with ExplicitRandom(t_rand=t_rand, normals=[ea, er]):
            rgbs, disp, depth, extras = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., verbose=False, retraw=False, **kw_train)
        loss = reference_kde_nll(rgbs, target, 3) + 0.01 * extras["loss_entropy"].mean()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        global_step += 1
    tar_tmp = os.path.join(tmp, "g22_reference_ckpt.tar")                                  # (the archive records its own base name: same name, same bytes)
    torch.save({                                                                           # RUN:1095-1099, the N_importance == 0 branch
        'global_step': global_step,
        'network_fn_state_dict': kw_train['network_fn'].state_dict(),
        'optimizer_state_dict': optimizer.state_dict(),
    }, tar_tmp)
    with open(tar_tmp, "rb") as f:
        tar_bytes = f.read()
    if not only or "g22_reference_ckpt" in only:
        with open(os.path.join(out_dir, "g22_reference_ckpt.tar"), "wb") as f:
            f.write(tar_bytes)
    ea = torch.tensor(rng22.standard_normal((3, 1)), dtype=torch.float32)
    er = torch.tensor(rng22.standard_normal((3, 3)), dtype=torch.float32)
    net.sample_alpha, net.sample_rgb = ea.clone(), er.clone()                             # eval latents: plain attributes, not in the checkpoint (R9)
    with torch.no_grad():
        rgbs_e, disp_e, depth_e, _ = R.render(H, W, focal, chunk=8192, rays=rays_t, near=0., far=1., **kw_test)
    sd = model.state_dict()
    g22 = dict(netwidth=64, netdepth=4, K=3, H=H, W=W, focal=focal, rays=rays_t, global_step=global_step, sample_alpha=ea, sample_rgb=er,
               rgb_map_eval=rgbs_e, disp_map_eval=disp_e, depth_map_eval=depth_e, state_dict_keys=np.array(list(sd.keys())),
               optimizer_state_entries=len(optimizer.state_dict()["state"]),
               tar_sha256=np.array(__import__("hashlib").sha256(tar_bytes).hexdigest()), tar_bytes=len(tar_bytes))
    for k, v in sd.items():
        f = v.detach().reshape(-1)
        g22["numel." + k] = f.numel()
        g22["head." + k] = f[:4].clone()
        g22["sum." + k] = f.double().sum()
    out["g22_reference_ckpt"] = g22

    import hashlib
    import json
    man_path = os.path.join(out_dir, "MANIFEST.json")
    manifest = {}
    if only and os.path.exists(man_path):
        with open(man_path) as f:
            manifest = json.load(f)
    for name, d in out.items():
        if only and name not in only:
            continue
        arrays = t2n(d)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **arrays)
        manifest[name] = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] + ":" + str(v.dtype) + str(list(v.shape))
                          for k, v in sorted(arrays.items())}
        print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")
    with open(man_path, "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
