"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import argparse
import os

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


# ---------------------------------------------------------------------------------------------------------------------
# Gradient parity with ReLU-mask accounting.
# Two fp32 implementations round a ~0 pre-activation to different sides now and then (the 2^9-frequency positional
# encoding alone carries ~3e-5 of fp32 noise), and one flipped unit changes the piecewise-linear function that is being
# differentiated.  Instead of a blanket tolerance on the trunk gradients, the tests (1) read the masks the HIP forward
# actually took from its stash, (2) check against the oracle's pre-activations that every disagreement sits on a
# pre-activation smaller than the layer's own activation error (so it IS rounding noise) and that there are few,
# (3) differentiate the oracle on the HIP masks and hold EVERY gradient entry to the tight fp32 bound.
G_TIGHT = 2e-4            # of the tensor's largest entry


def q4_to_rows(flat, cols):
    """A wide stash stream in the Q4 layout (csrc/cfnerf_device.h: per 64-point tile [half i][n-tile][group g][lane = 32 h + c][e], element =
    A[64 t + 32 i + 8 g + 4 h + e][32 nt + c]) as the row-major [P, cols] matrix; `flat` is a device or host tensor of P * cols floats."""
    nt = cols // 32
    return flat.reshape(-1, 2, nt, 4, 2, 32, 4).permute(0, 1, 3, 4, 6, 2, 5).reshape(-1, cols)


Q4_STREAMS = {"h": 1, "g_h": 1, "g_feat": 1, "feat": 1, "v": 2, "g_v": 2}        # wide streams that take the Q4 layout -> netwidth / this = their columns


def stash_copy(net, name, layer, n, keep=None):
    """buffer `name` of the last STASH forward as a CPU tensor in ROW-MAJOR order (the Q4 streams of a whole-tile fp32 forward are
    un-permuted here); keep = (lo, hi): only that range of its floats crosses to the host"""
    import ctypes as C
    import hooks
    from cfnerf_amd import _lib as L
    fn = hooks.lib().cfnerf_debug_copy_stash            # (test library: the product .so exports include/cfnerf.h only)
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    out = torch.empty(n, device="cuda")
    r = fn(net.handle, name.encode(), layer, C.c_void_p(out.data_ptr()), n, L.stream())
    assert r == n, (name, layer, r, n)
    q4 = hooks.lib().cfnerf_debug_stash_q4
    q4.restype, q4.argtypes = C.c_int, [C.c_void_p]
    if name in Q4_STREAMS and q4(net.handle) == 1:
        out = q4_to_rows(out, net.W // Q4_STREAMS[name]).reshape(-1)
    return (out if keep is None else out[keep[0]:keep[1]]).cpu()


def hip_relu_masks(net, P, rows=None):
    """0/1 masks of the last STASH forward: trunk<i> [P,W], views [P,W/2] (post-ReLU activation > 0); rows = (lo, hi): of those points only"""
    W, D = net.W, net.D
    lo, hi = rows if rows is not None else (0, P)
    acts = {f"trunk{i}": stash_copy(net, "h", i, P * W, (lo * W, hi * W)).reshape(hi - lo, W) for i in range(D)}
    acts["views"] = stash_copy(net, "v", 0, P * (W // 2), (lo * (W // 2), hi * (W // 2))).reshape(hi - lo, W // 2)
    return acts, {k: (v > 0).float() for k, v in acts.items()}


def oracle_train_step_on_hip_masks(net, p, packed, target, cfg, ea, er, t_rand, beta1, lindisp=False, white_bkgd=False,
                                   max_flip_frac=2e-4, t_vals=None, f64=True, loss_grad_at=None):
    """(scalars, grads, ret, n_flips): the oracle's train step differentiated on the ReLU masks of the HIP forward that
    was just run on the same inputs, after checking that those masks differ from the oracle's own only by rounding.
    `f64` (default): the differentiation runs in float64, so the reference gradient carries no fp32 noise of its own (the
    transmittance adjoint divides by ~1e-10 factors: an fp32 oracle is as noisy there as the kernels) and the per-tensor
    bounds of grad_close_tight measure the HIP path alone.
    `loss_grad_at` (an rgb_map, e.g. the HIP forward's): the cotangent of the backward is the fp64 loss gradient evaluated at THAT rgb_map
    instead of at the oracle's own (returned as ret["d_loss_d_rgb_map"]) - the reference of a backward run in isolation from the loss's
    steepness (K = 2: tests/test_hip_train.py); every calibration term below is then taken with that same cotangent."""
    S = 128 if t_vals is None else int(t_vals.shape[0])
    P = packed.shape[0] * S
    acts, masks = hip_relu_masks(net, P)
    rec = {}
    with torch.no_grad(), O.relu_override(record=rec):
        O.render_rays(p, packed, cfg, ea, er, True, t_rand, lindisp, white_bkgd, t_vals=t_vals)
    n_flips, n_units = 0, 0
    for k, pre in rec.items():
        own = (pre > 0).float()
        flip = own != masks[k]
        err = float((acts[k] - pre.clamp(min=0)).abs().max())          # activation disagreement of this layer
        assert err <= 2e-3 * max(1.0, float(pre.abs().max())), (k, err)
        if flip.any():
            worst = float(pre[flip].abs().max())
            assert worst <= max(err, 1e-7) * 1.0001, f"{k}: a ReLU mask differs at |pre-activation| {worst:.3e}, beyond the layer's activation error {err:.3e}"
        n_flips += int(flip.sum())
        n_units += flip.numel()
    assert n_flips <= max(2, max_flip_frac * n_units), f"{n_flips} of {n_units} ReLU masks differ"
    if not f64:
        with O.relu_override(masks=masks):
            scal, grads, ret = O.train_step(p, packed, target, cfg, ea, er, t_rand, beta1, lindisp, white_bkgd, t_vals=t_vals)
        return scal, grads, ret, n_flips
    d = lambda t: None if t is None else t.double()
    with O.relu_override(masks=masks):
        # train_step of the oracle in fp64, written out so that the backward can be run twice (see below)
        q = {k: d(v).clone().requires_grad_(True) for k, v in p.items()}
        O.latent_tap = tap = {}
        try:
            ret = O.render_rays(q, d(packed), cfg, d(ea), d(er), True, d(t_rand), lindisp, white_bkgd, t_vals=d(t_vals))
        finally:
            O.latent_tap = None
        L = O.train_loss(ret["rgb_map"], d(target), ret["loss_entropy"], cfg.K_samples, beta1)
        keys = [k for k in q]
        if loss_grad_at is None:
            (G,) = torch.autograd.grad(L["loss"], ret["rgb_map"], retain_graph=True)
        else:
            r_at = d(loss_grad_at).detach().clone().requires_grad_(True)
            (G,) = torch.autograd.grad(O.train_loss(r_at, d(target), ret["loss_entropy"].detach(), cfg.K_samples, beta1)["loss"], r_at)
        outs, cots = [ret["rgb_map"]], [G]
        if beta1:
            outs.append(ret["loss_entropy"]); cots.append(torch.tensor(float(beta1), dtype=torch.float64))
        gl = torch.autograd.grad(outs, [q[k] for k in keys], cots, retain_graph=True, allow_unused=True)
        grads = dict(zip(keys, gl))
        scal = {k: float(v.detach()) for k, v in L.items()}
        scal["loss_entropy"] = float(ret["loss_entropy"].detach())
        ret = {k: v.detach() for k, v in ret.items() if v is not None}
        ret["d_loss_d_rgb_map"] = G.detach()
        # Calibration of the bound, per tensor and per case.  (1) the SAME differentiation in fp32: what a straightforward fp32
        # implementation of this math delivers on THESE inputs.  (2) the conditioning of the case: a correct fp32 forward returns
        # rgb_map within ~5e-7, which moves the loss gradient d loss / d rgb_map by ~2e-6 of its largest entry (measured: HIP 1.7e-6,
        # fp32 oracle 1.0e-6) - for K = 2..3 the KDE bandwidth is tiny and the loss is steep; how far THAT moves each parameter
        # gradient is measured by a second fp64 backward with the cotangent perturbed by 2e-6 max|G|.  The alpha path (transmittance
        # adjoint: differences of nearly equal terms) amplifies it up to 100x on random sample tables; measured with
        # tests/tools/alpha_grad_diag.py: the fused backward fed the EXACT fp64 loss gradient at the HIP forward's rgb_map shows the
        # same deviation, i.e. it is the true gradient at a forward point 5e-7 away, not an error of the backward.
        gen = torch.Generator().manual_seed(0)
        Gp = G + 2e-6 * float(G.abs().max()) * (torch.rand(G.shape, generator=gen, dtype=torch.float64) * 2 - 1)
        gp = torch.autograd.grad(outs, [q[k] for k in keys], [Gp] + cots[1:], retain_graph=True, allow_unused=True)
        if loss_grad_at is None:
            _, g32, _ = O.train_step(p, packed, target, cfg, ea, er, t_rand, beta1, lindisp, white_bkgd, t_vals=t_vals)
        else:                                            # the same fp32 differentiation, fed the same cotangent
            q32 = {k: v.clone().requires_grad_(True) for k, v in p.items()}
            r32 = O.render_rays(q32, packed, cfg, ea, er, True, t_rand, lindisp, white_bkgd, t_vals=t_vals)
            o32, c32 = [r32["rgb_map"]], [G.float()]
            if beta1:
                o32.append(r32["loss_entropy"]); c32.append(torch.tensor(float(beta1)))
            g32 = dict(zip(keys, torch.autograd.grad(o32, [q32[k] for k in keys], c32, allow_unused=True)))
        # Tensors of a few elements (the base Gaussians: each entry is ONE sum over all (point, latent) contributions, which largely
        # cancel) get no averaging over entries: with a single perturbation draw the ratio error / noise is a ratio of two
        # half-normal variables and exceeds 10 in ~6 % of the cases (found by the CFNERF_FUZZ_SEEDS soak: 9 % of 140 random
        # configurations failed on alpha_mean / alpha_std / rgb_mean alone).  Their noise is therefore the largest of 8 draws; the
        # backward to these tensors stops at the flows (no MLP), so the extra passes are cheap.
        small = [k for k, g in zip(keys, gl) if g is not None and g.numel() <= 4]
        extra = {k: 0.0 for k in small}
        for draw in range(1, 8):
            gen = torch.Generator().manual_seed(draw)
            Gq = G + 2e-6 * float(G.abs().max()) * (torch.rand(G.shape, generator=gen, dtype=torch.float64) * 2 - 1)
            gq = torch.autograd.grad(outs, [q[k] for k in small], [Gq] + cots[1:], retain_graph=True, allow_unused=True)
            for k, v in zip(small, gq):
                if v is not None:
                    extra[k] = max(extra[k], float((v - grads[k]).abs().max()))
        # ... and their entries CANCEL: d loss / d alpha_mean = sum over (point, latent) of c = d loss / d alpha0 (alpha0 = eps std + mean,
        # MOD:239; alpha_std: eps c), often to 1e-2 of sum |c|, while every c carries the alpha path's conditioning error (5e-5 .. 2e-4
        # relative, tests/tools/alpha_grad_diag.py).  The natural scale of the error is sum |c|, not the result: these tensors get an
        # absolute allowance of G_SUM_REL sum |c| (soak over 140 random configurations: largest observed error 3.8e-4 sum |c|; a
        # reduction that dropped one ray of N would be off by ~sum |c| / N).
        ca, cr = torch.autograd.grad(outs, [tap["alpha0"], tap["rgb0"]], cots, retain_graph=True)
        Kl = int(ea.shape[0])
        ca, cr = ca.reshape(-1, Kl, 1), cr.reshape(-1, Kl, 3)
        sum_abs = {"alpha_mean": ca.abs().sum((0, 1)), "rgb_mean": cr.abs().sum((0, 1)),
                   "alpha_std": (ca * d(ea)[None]).abs().sum((0, 1)), "rgb_std": (cr * d(er)[None]).abs().sum((0, 1))}
        gmax = max(float(g.abs().max()) for g in gl if g is not None)
        for k, g, g2 in zip(keys, gl, gp):
            if g is not None:
                g.global_scale = gmax                     # largest gradient entry of the whole step (grad_close_tight: absolute floor)
            if g is not None and k in sum_abs:
                g.sum_abs = sum_abs[k].detach().cpu().numpy()
        for k, g, g2 in zip(keys, gl, gp):
            if g is not None:
                m = g.abs().max().clamp_min(1e-300)
                g.fp32_noise = max(float((g32[k].double() - g).abs().max() / m), float((g2 - g).abs().max() / m), extra.get(k, 0.0) / float(m))
    return scal, grads, ret, n_flips


G_FLOOR, G_CAP, G_NOISE_X = 2e-5, 1e-3, 16.0
G_ABS_EPS = 1e-7           # absolute floor of every comparison, in units of the step's largest gradient entry
G_SUM_REL = 1e-3           # base-Gaussian gradients: allowance relative to the sum of the magnitudes of the terms they add up
G_ALPHA_X = 2.0            # the density ("alpha") path is the ill-conditioned one (transmittance adjoint): twice the noise multiple
G_CAP_OTHER = 2e-4         # every tensor OUTSIDE the alpha path: the calibration can only tighten the fixed bound of round 2, never loosen it


def kde_loss_gradient(rgbs, target, K, n_total=None):
    """d loss_nll / d rgb_map of the reference's KDE loss (RUN:1032-1042) as an EXPLICIT function of rgb_map [N,3,K], differentiable ONCE
    MORE with the bandwidth's own dependence on rgb_map included: the reference detaches the bandwidth from the loss's graph, so autograd's
    double-backward of O.train_loss misses the path through it, while the gradient VALUE moves with it.  Its Jacobian applied to the
    difference of two forwards is how far the loss gradient moves between them (tests/test_hip_train.py: the K = 2 link-by-link step)."""
    import math
    n, eps = K, 1e-05
    d = rgbs - target[..., None]
    bw = float(torch.pow(torch.tensor(0.8 / n), torch.tensor(-1 / 7)))           # (fp32 constants, as RUN:1036,1039 make them)
    c2pi = float(torch.pow(torch.tensor(2 * math.pi), -1.5))
    H = (torch.std(rgbs, -1) * n / (n - 1) * bw + eps)[..., None]
    r1 = torch.exp(-(d * d) / (2 * H * H))
    r2 = c2pi / H
    m = (r1 * r2).mean(-1, keepdim=True) + eps
    return (r1 * r2) * d / (H * H) / m / (K * 3.0 * (n_total or rgbs.shape[0]))


ONE_RAY_COVERAGE = 0.1     # a one-ray gradient that lost or doubled a k-part / a tile of the ray is off by >= 0.25 of its largest entry


def one_ray_conditioning(ret64, q64, keys, Gi, white_bkgd, draws=3, rel_step=2e-7, seed=0):
    """{tensor: c}: how far the fp64 one-ray gradient of sum(Gi * rgb_map) moves (of its largest entry; the largest of `draws` draws) when
    the forward's flow outputs `raw` move by rel_step of their magnitude - what two correct fp32 forwards differ by (measured: HIP raw against the
    fp64 chain 1.3e-7 .. 2.6e-7).  ret64 / q64: the fp64 oracle's render_rays of that ray (graph alive) and its parameter leaves."""
    raw, z = ret64["raw"], ret64["z_vals"].detach()
    gen = torch.Generator().manual_seed(seed)

    def params_grad(raw_at):
        x = raw_at.detach().clone().requires_grad_(True)
        o = O.raw2outputs(x, z, ret64["rays_d"], white_bkgd)
        (gx,) = torch.autograd.grad((o[0] * Gi.to(x.dtype)[None]).sum(), x)
        return dict(zip(keys, torch.autograd.grad(raw, [q64[k] for k in keys], gx, retain_graph=True, allow_unused=True)))
    g0 = params_grad(raw)
    out = {k: 0.0 for k in keys if g0[k] is not None and float(g0[k].abs().max()) > 0.0}
    for _ in range(draws):
        u = torch.rand(raw.shape, generator=gen, dtype=raw.dtype) * 2 - 1
        g1 = params_grad(raw.detach() * (1.0 + rel_step * u))
        for k in out:
            out[k] = max(out[k], float((g1[k] - g0[k]).abs().max() / g0[k].abs().max()))
    return out


def density_path_tol(n32, cond, key, floor=G_FLOOR):
    """Bound of ONE density-path tensor of a ONE-ray gradient (of the tensor's largest entry): max(floor, 8 x noise) - the colour branch's
    rule - capped at the coverage bound, with the noise of this tensor ON THIS RAY read robustly:
      n32[key]   what the SAME one-ray differentiation in fp32 on the CPU (the reference's arithmetic) loses against fp64,
      cond[key]  the conditioning: how far a 2e-7 move of the forward's `raw` moves this tensor in fp64 (one_ray_conditioning),
      a quarter of the worst n32 on the density path.
    Why three readings: the density tensors of a ray are all linear images of ONE vector, d loss / d raw[..., 3] per (sample, latent), whose
    entries carry the fp32 forward's conditioning (alpha = 1 - e of a nearly transparent sample keeps 3 digits: the kernel's formula
    evaluated in fp64 at its fp32 inputs is 1e-4 .. 5e-4 per entry from the exact answer, like the fp32 oracle and like the kernel -
    tests/tools/density_bisect.py, profiles/r06_density_bisect.txt).  How much of it survives in a tensor is ONE draw of a cancelling sum,
    and the ratio of two such draws (kernel / fp32 oracle) is heavy-tailed: the oracle's draw is now and then 20 - 50 x luckier on one
    tensor than on its neighbours (measured: alpha_mean 1.3e-6 next to alpha_std 1.8e-4)."""
    n_path = max(v for k, v in n32.items() if "alpha" in k)
    return min(max(floor, 8.0 * max(n32[key], cond.get(key, 0.0), 0.25 * n_path)), ONE_RAY_COVERAGE)


def grad_close_tight(g, ref, what, tol=None):
    """Every entry of a gradient tensor within `tol` of the tensor's largest entry (+ 1e-4 relative), and the tensor's RMS error
    within `tol` of its RMS.  `ref`: numpy, or a torch tensor from oracle_train_step_on_hip_masks - those carry `fp32_noise`, the
    error of the fp32 CPU oracle against the fp64 one on the same inputs, and the bound is G_NOISE_X times that, clipped to
    [G_FLOOR, G_CAP] (measured over the suite: HIP error / fp32-oracle error stays below G_NOISE_X; typical errors are 3e-7 .. 6e-6,
    the floor sits ~10x above them).  Without it (fixtures of the real reference, themselves fp32): the fixed G_TIGHT."""
    noise = getattr(ref, "fp32_noise", None)
    sum_abs = getattr(ref, "sum_abs", None)              # (base Gaussians: sum of the magnitudes of the terms each entry adds up)
    # absolute floor: G_ABS_EPS of the LARGEST gradient entry of the step (~2 eps32).  A tensor whose whole gradient is 1e-7 of the step's
    # (the soak hit density-path gradients of 1e-9 next to 1e-2 elsewhere: saturated rays) is the difference of fp32 terms far larger
    # than itself and cannot be resolved to 1e-3 of ITS largest entry by any fp32 pipeline.
    floor = G_ABS_EPS * float(getattr(ref, "global_scale", 0.0))
    if torch.is_tensor(ref):
        ref = ref.detach().cpu().double().numpy()
    ref = np.asarray(ref, dtype=np.float64)
    g = g.detach().cpu().double().numpy() if torch.is_tensor(g) else np.asarray(g, dtype=np.float64)
    nx = G_NOISE_X * (G_ALPHA_X if ("alpha" in what) else 1.0)
    cap = G_CAP if ("alpha" in what) else G_CAP_OTHER
    if tol is None:
        tol = G_TIGHT if noise is None else min(max(nx * noise, G_FLOOR), cap)
    tol_rms = tol if noise is None else min(max(nx * noise, 1.5 * G_FLOOR), cap)
    scale = max(float(np.abs(ref).max()), 1e-12)
    rms_rel = float(np.sqrt(((g - ref) ** 2).sum() / max(float((ref ** 2).sum()), 1e-300)))
    stats = os.environ.get("CFNERF_GRAD_STATS")          # development aid: log the measured error of every comparison
    if stats:
        import json
        with open(stats, "a") as f:
            f.write(json.dumps({"what": what, "rel_to_max": float(np.abs(g - ref).max() / scale), "scale": scale, "n": int(ref.size),
                                "rms_rel": rms_rel, "noise": noise, "tol": tol}) + "\n")
    # (there is no "record only" mode: a variable that switched the assertions off would make every gradient test pass vacuously
    # wherever it leaked to; tests/conftest.py refuses to start with the old CFNERF_GRAD_STATS_ONLY set)
    if sum_abs is not None:                              # a cancelling sum: absolute allowance of G_SUM_REL x sum |terms| per entry
        allow = G_SUM_REL * np.asarray(sum_abs, dtype=np.float64).reshape(ref.shape)
        err = np.abs(g - ref)
        if stats:
            with open(stats, "a") as f:
                f.write(json.dumps({"what": what + " (sum)", "err_over_sum_abs": float((err / np.maximum(allow / G_SUM_REL, 1e-300)).max())}) + "\n")
        assert np.all(err <= allow + floor + tol * scale + 1e-4 * np.abs(ref)), f"{what}: max err {err.max():.3e} beyond {tol:.1e} of the largest entry + {G_SUM_REL:.0e} sum|terms| ({allow.max():.3e})"
        return
    close(g, ref, atol=tol * scale + floor, rtol=1e-4, what=what)
    rms_ref = float(np.sqrt((ref ** 2).mean()))
    assert rms_rel * rms_ref <= tol_rms * rms_ref + floor, f"{what}: RMS error {rms_rel:.2e} of the tensor's RMS exceeds {tol_rms:.1e}"


def fuzz_case(seed, run=True, **force):
    """The seeded random draw of tests/test_hip_train.py::test_random_configurations_forward_and_gradients_vs_oracle, as a function so that
    the named regression cases and tests/tools/k2_grad_diag.py replay EXACTLY the draw a soak seed made.  Seeds >= 1000: every width;
    >= 2000: sample tables other than the reference's 128 entries; >= 3000: MANY rays through a small network (every workgroup walks
    several rays).  `force` overrides drawn values AFTER the draw (the stream of random numbers stays the seed's): W, D, K, ha, hr, N, S.
    Returns the configuration, the inputs and - with `run` - the model, the Trainer and its flat gradient of the step."""
    from cfnerf_amd import train as TR
    rng = np.random.default_rng(9000 + seed)
    W = int(rng.choice([64, 128, 192, 256, 320] if seed < 1000 else [64, 128, 192, 256, 320, 384, 448, 512]))
    D = int(rng.choice([4, 5, 6, 8]))
    K = int(rng.choice([2, 3, 4, 5, 6, 16, 32, 72]))
    ha = int(rng.choice([32, 64, 96, 128] if W <= 256 else [32, 64, 96]))
    hr = int(rng.choice([h for h in (32, 64, 96, 128) if W // 2 + h <= max(W, 128)]))
    N = int(rng.integers(3, 24))
    if seed >= 3000:
        W, D, K, ha, hr, N = 64, int(rng.choice([4, 5])), int(rng.choice([2, 3, 4])), 32, 32, int(rng.integers(600, 1500))
    ndc = bool(rng.integers(0, 2))
    lindisp = (not ndc) and bool(rng.integers(0, 2))
    wb = bool(rng.integers(0, 2))
    perturb = bool(rng.integers(0, 4))                   # mostly on
    nf = int(rng.choice([4, 4, 4, 3, 2]))
    W, D, K, ha, hr, N = (force.get("W", W), force.get("D", D), force.get("K", K), force.get("ha", ha), force.get("hr", hr), force.get("N", N))
    cfg = O.OracleCfg(netwidth=W, netdepth=D, K_samples=K, h_alpha_size=ha, h_rgb_size=hr, n_flows=nf)
    c = dict(seed=seed, W=W, D=D, K=K, ha=ha, hr=hr, nf=nf, N=N, ndc=ndc, lindisp=lindisp, wb=wb, perturb=perturb, cfg=cfg)
    if run:
        _, kw_train, _, model, p, _ = build_model(cfg, 700 + seed, no_ndc=not ndc, lindisp=lindisp, white_bkgd=wb)
        net = model.module
        if os.environ.get("CFNERF_FUZZ_PREC"):           # soak of the opt-in mode: CFNERF_FUZZ_PREC=bf16x3
            net.set_precision(os.environ["CFNERF_FUZZ_PREC"])
        c.update(kw_train=kw_train, model=model, net=net, p=p)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    near, far = (0., 1.) if ndc else (1.2, 8.0)
    S = 128 if seed < 2000 else int(rng.choice([16, 64, 100, 128, 130, 192, 257]))
    S = force.get("S", S)
    t_vals = None if S == 128 else torch.linspace(0., 1., S)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, S)), dtype=torch.float32) if perturb else None
    ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
    target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32)
    beta1 = float(rng.choice([0.0, 0.01, 0.1]))
    packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], ndc, near, far)
    c.update(S=S, rays=rays, H=H, Wd=Wd, focal=focal, near=near, far=far, t_vals=t_vals, t_rand=t_rand, ea=ea, er=er, target=target,
             beta1=beta1, packed=packed)
    if run:
        tr = TR.Trainer(net, beta1=beta1)
        dev = "cuda"
        c["fb_kw"] = dict(t_rand=None if t_rand is None else t_rand.to(dev), eps=torch.cat([er, ea], -1).to(dev), near=near, far=far, ndc=ndc,
                          lindisp=lindisp, white_bkgd=wb, perturb=1. if perturb else 0., t_vals=None if t_vals is None else t_vals.to(dev))
        grad = tr.forward_backward(H, Wd, focal, rays.to(dev), target.to(dev), **c["fb_kw"]).cpu()
        c.update(tr=tr, grad=grad)
    return c
