"""Development aid (round 5): is a gradient-BOUND failure of a soak draw (K = 2, many rays: seeds 5027 / 5029 of
test_random_configurations_forward_and_gradients_vs_oracle) a defect of the backward kernels or the conditioning of the case?

    python tests/tools/k2_grad_diag.py 5027 5029 [--force W=256,S=16] [--rays 12] [--json out.json]

Per seed, three separate comparisons against the fp64 oracle differentiated on the ReLU masks of the HIP forward:
 (1) FULL step: what the suite judges (HIP loss kernel's d_rgb at the HIP rgb_map  vs  fp64 loss gradient at the fp64 rgb_map).
 (2) ISOLATED backward: the fused backward is fed the fp64 loss gradient evaluated AT THE HIP FORWARD'S OWN rgb_map (cast to fp32), the
     fp64 oracle is differentiated with that same cotangent: the loss's steepness (the K-dependent KDE bandwidth, RUN:1032-1042) is out
     of the comparison, what is left is the backward kernels' own arithmetic.  The conditioning factor of every tensor is printed:
     kappa = (|dg| / |g|) / (|dG| / |G|) for a random perturbation dG of the cotangent, and how far the two loss gradients are apart.
 (3) PER RAY: a one-hot cotangent (one ray at a time: first / last of the batch, both sides of 64-ray and 4-ray boundaries, a few random
     ones) against the fp64 oracle on THAT ray alone - a dropped or duplicated partial of a cross-tile reduction is an O(1) error here."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from cfnerf_amd import _lib as L
from oracle import cfnerf_oracle as O
from util_hip import G_ALPHA_X, G_CAP, G_CAP_OTHER, G_FLOOR, G_NOISE_X, fuzz_case, hip_relu_masks, oracle_train_step_on_hip_masks

d = lambda t: None if t is None else t.double()


def rel_max(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def rel_rms(a, b):
    a, b = a.detach(), b.detach()
    return float(((a.double() - b.double()) ** 2).sum().sqrt() / (b.double() ** 2).sum().sqrt().clamp_min(1e-300))


def hip_bwd(net, d_rgb, d_ent):
    lib = L.lib()
    gen = lib.cfnerf_model_stash_generation(net.handle)
    gout = torch.empty(net.n_params, device="cuda")
    L.check(lib.cfnerf_render_bwd(net.handle, gen, L.ptr(d_rgb.float().cuda().contiguous()), None, None if d_ent is None else L.ptr(d_ent), L.ptr(gout),
                                  L.stream()), "cfnerf_render_bwd")
    return gout.cpu()


def diag(seed, force, n_rays):
    c = fuzz_case(seed, **force)
    net, p, cfg, tr = c["net"], c["p"], c["cfg"], c["tr"]
    N, S, K, beta1 = c["N"], c["S"], c["K"], c["beta1"]
    print(f"==== seed {seed}: W={c['W']} D={c['D']} K={K} N={N} S={S} ndc={c['ndc']} wb={c['wb']} perturb={c['perturb']} beta1={beta1} F={c['nf']}", flush=True)
    out = {"seed": seed, "cfg": {k: c[k] for k in ("W", "D", "K", "N", "S", "ha", "hr", "nf", "ndc", "lindisp", "wb", "perturb", "beta1")}}
    keys = [k for k in net.layout]
    # ---- (1) the suite's comparison
    scal, grads, ret, flips = oracle_train_step_on_hip_masks(net, p, c["packed"], c["target"], cfg, c["ea"], c["er"], c["t_rand"], beta1,
                                                             lindisp=c["lindisp"], white_bkgd=c["wb"], t_vals=c["t_vals"])
    full = {}
    for k in keys:
        if grads[k] is None or float(grads[k].abs().max()) == 0.0:
            continue
        off, cnt = net.layout[k]
        g = c["grad"][off:off + cnt].reshape(grads[k].shape)
        nx = G_NOISE_X * (G_ALPHA_X if "alpha" in k else 1.0)
        cap = G_CAP if "alpha" in k else G_CAP_OTHER
        noise = grads[k].fp32_noise
        full[k] = dict(err_max=rel_max(g, grads[k]), err_rms=rel_rms(g, grads[k]), noise=noise, tol=min(max(nx * noise, G_FLOOR), cap),
                       tol_rms=min(max(nx * noise, 1.5 * G_FLOOR), cap))
    out["full"] = full
    bad = [k for k, v in full.items() if (v["err_max"] > v["tol"] or v["err_rms"] > v["tol_rms"]) and not hasattr(grads[k], "sum_abs")]
    print(f"(1) full step, {flips} mask flips; tensors beyond the suite's bound: {bad}")
    for k in sorted(full, key=lambda k: -full[k]["err_rms"] / full[k]["tol_rms"])[:6]:
        v = full[k]
        print(f"    {k:34s} max {v['err_max']:.2e} (tol {v['tol']:.1e})  rms {v['err_rms']:.2e} (tol {v['tol_rms']:.1e})  fp32-oracle/conditioning noise {v['noise']:.1e}")
    # ---- (2) isolated backward: same cotangent on both sides
    masks = hip_relu_masks(net, N * S)[1]
    with O.relu_override(masks=masks):
        q = {k: d(v).clone().requires_grad_(True) for k, v in p.items()}
        r64 = O.render_rays(q, d(c["packed"]), cfg, d(c["ea"]), d(c["er"]), True, d(c["t_rand"]), c["lindisp"], c["wb"], t_vals=d(c["t_vals"]))
        q32 = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        r32 = O.render_rays(q32, c["packed"], cfg, c["ea"], c["er"], True, c["t_rand"], c["lindisp"], c["wb"], t_vals=c["t_vals"])
    rgb_h = tr.rgb_map.detach().cpu().double().requires_grad_(True)
    (Gh,) = torch.autograd.grad(O.train_loss(rgb_h, d(c["target"]), r64["loss_entropy"].detach(), K, beta1)["loss"], rgb_h)
    rgb_o = r64["rgb_map"].detach().clone().requires_grad_(True)
    (Go,) = torch.autograd.grad(O.train_loss(rgb_o, d(c["target"]), r64["loss_entropy"].detach(), K, beta1)["loss"], rgb_o)
    out["rgb_map_hip_vs_f64"] = rel_max(tr.rgb_map.cpu(), r64["rgb_map"].detach())
    out["loss_grad_at_hip_point_vs_at_f64_point"] = dict(max=rel_max(Gh, Go), rms=rel_rms(Gh, Go))
    out["loss_kernel_vs_f64_loss_grad_at_its_own_point"] = dict(max=rel_max(tr.d_rgb.cpu(), Gh), rms=rel_rms(tr.d_rgb.cpu(), Gh))
    print(f"(2) rgb_map HIP vs fp64: {out['rgb_map_hip_vs_f64']:.2e} of max;  d loss/d rgb_map at the HIP point vs at the fp64 point: max {rel_max(Gh, Go):.2e} rms {rel_rms(Gh, Go):.2e}"
          f";  loss kernel vs fp64 at its own point: max {rel_max(tr.d_rgb.cpu(), Gh):.2e}")
    Ghf = Gh.float()                                             # what the kernels can be given
    g_iso = hip_bwd(net, Ghf, tr.d_ent if beta1 else None)
    outs, cots = [r64["rgb_map"]], [Ghf.double()]
    outs32, cots32 = [r32["rgb_map"]], [Ghf]
    if beta1:
        outs.append(r64["loss_entropy"]); cots.append(torch.tensor(float(beta1), dtype=torch.float64))
        outs32.append(r32["loss_entropy"]); cots32.append(torch.tensor(float(beta1)))
    ref = dict(zip(keys, torch.autograd.grad(outs, [q[k] for k in keys], cots, retain_graph=True, allow_unused=True)))
    ref32 = dict(zip(keys, torch.autograd.grad(outs32, [q32[k] for k in keys], cots32, retain_graph=True, allow_unused=True)))
    gen = torch.Generator().manual_seed(0)
    dG = 1e-4 * float(Ghf.abs().max()) * (torch.rand(Ghf.shape, generator=gen, dtype=torch.float64) * 2 - 1)
    refp = dict(zip(keys, torch.autograd.grad(outs, [q[k] for k in keys], [Ghf.double() + dG] + cots[1:], retain_graph=True, allow_unused=True)))
    dG_rel = float((dG ** 2).sum().sqrt() / (Ghf.double() ** 2).sum().sqrt())
    iso = {}
    for k in keys:
        if ref[k] is None or float(ref[k].abs().max()) == 0.0:
            continue
        off, cnt = net.layout[k]
        g = g_iso[off:off + cnt].reshape(ref[k].shape)
        iso[k] = dict(err_max=rel_max(g, ref[k]), err_rms=rel_rms(g, ref[k]), cpu32_max=rel_max(ref32[k], ref[k]), cpu32_rms=rel_rms(ref32[k], ref[k]),
                      kappa=rel_rms(refp[k], ref[k]) / dG_rel)
    out["isolated"] = iso
    worst = sorted(iso, key=lambda k: -iso[k]["err_rms"])[:8]
    print(f"    isolated backward (fp64 loss gradient at the HIP rgb_map on both sides), worst tensors by RMS error (G_FLOOR = {G_FLOOR:.0e}):")
    for k in sorted(set(worst) | set(bad)):
        v = iso[k]
        print(f"    {k:34s} hip: max {v['err_max']:.2e} rms {v['err_rms']:.2e}   fp32 CPU oracle: max {v['cpu32_max']:.2e} rms {v['cpu32_rms']:.2e}   kappa {v['kappa']:.1f}")
    out["isolated_worst_max"] = max(v["err_max"] for k, v in iso.items())
    out["isolated_worst_rms"] = max(v["err_rms"] for k, v in iso.items())
    print(f"    all tensors: worst max-error {out['isolated_worst_max']:.2e}, worst RMS error {out['isolated_worst_rms']:.2e}")
    # ---- (3) one ray at a time
    rng = np.random.default_rng(seed)
    cand = [0, 1, 3, 4, 63, 64, 255, 256, N // 2, N - 2, N - 1] + [int(x) for x in rng.integers(0, N, 16)]
    rays_i = []
    for i in cand:
        if 0 <= i < N and i not in rays_i:
            rays_i.append(i)
    rays_i = rays_i[:n_rays]
    per_ray = {}
    tv64 = d(c["t_vals"])
    for i in rays_i:
        Gi = torch.tensor(rng.standard_normal((3, K)), dtype=torch.float32)
        G = torch.zeros(N, 3, K)
        G[i] = Gi
        g_hip = hip_bwd(net, G, None).double()
        m_i = {k: v[i * S:(i + 1) * S] for k, v in masks.items()}
        qi = {k: d(v).clone().requires_grad_(True) for k, v in p.items()}
        tr_i = None if c["t_rand"] is None else c["t_rand"][i:i + 1]
        with O.relu_override(masks=m_i):
            r = O.render_rays(qi, d(c["packed"][i:i + 1]), cfg, d(c["ea"]), d(c["er"]), True, d(tr_i), c["lindisp"], c["wb"], t_vals=tv64)
        (r["rgb_map"] * d(Gi)[None]).sum().backward()
        qj = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        with O.relu_override(masks=m_i):
            rj = O.render_rays(qj, c["packed"][i:i + 1], cfg, c["ea"], c["er"], True, tr_i, c["lindisp"], c["wb"], t_vals=c["t_vals"])
        (rj["rgb_map"] * Gi[None]).sum().backward()
        w, wo = ("", 0.0, 0.0), ("", 0.0, 0.0)                   # worst tensor overall / outside the density ("alpha") path
        for k in keys:
            gk = qi[k].grad
            if gk is None or float(gk.abs().max()) == 0.0:
                continue
            off, cnt = net.layout[k]
            e = rel_max(g_hip[off:off + cnt].reshape(gk.shape), gk)
            if e > w[1]:
                w = (k, e, rel_max(qj[k].grad, gk))
            if "alpha" not in k and e > wo[1]:
                wo = (k, e, rel_max(qj[k].grad, gk))
        per_ray[i] = dict(worst_tensor=w[0], err_max=w[1], cpu32_err_max=w[2], worst_other=wo[0], other_err_max=wo[1], other_cpu32_err_max=wo[2])
        print(f"(3) ray {i:5d}: worst tensor {w[0]:30s} error {w[1]:.2e} of its largest entry (fp32 CPU oracle on this ray: {w[2]:.2e});  outside the alpha path "
              f"{wo[0]:28s} {wo[1]:.2e} ({wo[2]:.2e})", flush=True)
    out["per_ray"] = per_ray
    # ---- (4) the worst rays of (3): is the one-ray error the composite adjoint's arithmetic, or the forward point it is evaluated at?
    #      raw_hip = what the HIP forward computed for the ray (1e-6 from the fp64 raw).  A = fp64 composite adjoint AT raw_hip,
    #      B = fp64 composite adjoint at the fp64 raw (the reference of (3)), H = the product's composite adjoint kernel at raw_hip.
    #      H vs A = arithmetic of the adjoint; A vs B = how far a 1e-6 move of the forward point moves the gradient (conditioning).
    #      Then the network backward in fp64 fed with A: the parameter gradient "at the HIP forward point", against the HIP gradient.
    import cfnerf_amd
    kw = dict(c["kw_train"])
    kw["perturb"] = 1. if c["perturb"] else 0.
    if c["t_vals"] is not None:
        kw["t_vals"] = c["t_vals"].cuda()
    with torch.no_grad():
        _, _, _, ex = cfnerf_amd.render(c["H"], c["Wd"], c["focal"], rays=c["rays"].cuda(), near=c["near"], far=c["far"],
                                        t_rand=c["t_rand"], eps_alpha=c["ea"], eps_rgb=c["er"], **kw)
    raw_hip_all = ex["raw"].detach().cpu()
    tr.forward_backward(c["H"], c["Wd"], c["focal"], c["rays"].cuda(), c["target"].cuda(), **c["fb_kw"])      # (the stash of the step again)
    zv_all = O.sample_z(c["packed"][:, 6:7], c["packed"][:, 7:8], O.t_vals_table() if c["t_vals"] is None else c["t_vals"], c["lindisp"], c["t_rand"])
    worst_rays = sorted(per_ray, key=lambda i: -per_ray[i]["err_max"])[:4]
    split = {}
    rng = np.random.default_rng(seed)
    for i in worst_rays:
        Gi = torch.tensor(np.random.default_rng(seed * 1000 + i).standard_normal((3, K)), dtype=torch.float32)
        m_i = {k: v[i * S:(i + 1) * S] for k, v in masks.items()}
        qi = {k: d(v).clone().requires_grad_(True) for k, v in p.items()}
        tr_i = None if c["t_rand"] is None else c["t_rand"][i:i + 1]
        with O.relu_override(masks=m_i):
            r = O.render_rays(qi, d(c["packed"][i:i + 1]), cfg, d(c["ea"]), d(c["er"]), True, d(tr_i), c["lindisp"], c["wb"], t_vals=tv64)
        raw64 = r["raw"]
        zi, di = d(zv_all[i:i + 1]), d(c["packed"][i:i + 1, 3:6])
        def comp_adj(raw_in):
            x = raw_in.detach().clone().requires_grad_(True)
            o = O.raw2outputs(x, zi.to(x.dtype), di.to(x.dtype), c["wb"])
            (gx,) = torch.autograd.grad((o[0] * Gi.to(x.dtype)[None]).sum(), x)
            return gx
        rh = raw_hip_all[i:i + 1].double().reshape(raw64.shape)
        A, B = comp_adj(rh), comp_adj(raw64)
        xh = raw_hip_all[i:i + 1].reshape(raw64.shape).cuda().requires_grad_(True)
        oh = cfnerf_amd.raw2outputs(xh, zv_all[i:i + 1].cuda(), c["packed"][i:i + 1, 3:6].cuda(), white_bkgd=c["wb"])
        (Hh,) = torch.autograd.grad((oh[0] * Gi.cuda()[None]).sum(), xh)
        g_hip = hip_bwd(net, torch.zeros(N, 3, K).index_put_((torch.tensor([i]),), Gi[None]), None).double()
        gA = dict(zip(keys, torch.autograd.grad(raw64, [qi[k] for k in keys], A, retain_graph=True, allow_unused=True)))
        gB = dict(zip(keys, torch.autograd.grad(raw64, [qi[k] for k in keys], B, retain_graph=True, allow_unused=True)))
        wk = per_ray[i]["worst_tensor"]
        off, cnt = net.layout[wk]
        gh = g_hip[off:off + cnt].reshape(gA[wk].shape)
        A32 = comp_adj(raw_hip_all[i:i + 1].reshape(raw64.shape))                 # torch's fp32 composite adjoint at the same raw
        print(f"    ray {i}: density channel of d_raw, relative to ITS OWN largest entry: kernel {rel_max(Hh.cpu()[..., 3], A[..., 3]):.1e}, torch fp32 {rel_max(A32[..., 3], A[..., 3]):.1e}"
              f"  (|d_raw density| max / |d_raw| max = {float(A[..., 3].abs().max() / A.abs().max()):.1e})")
        split[i] = dict(tensor=wk, raw_hip_vs_f64=rel_max(rh, raw64), adj_kernel_vs_f64_same_point=rel_max(Hh.cpu(), A),
                        adj_density_kernel=rel_max(Hh.cpu()[..., 3], A[..., 3]), adj_density_torch32=rel_max(A32[..., 3], A[..., 3]),
                        adj_f64_hip_point_vs_f64_point=rel_max(A, B), grad_hip_vs_f64_at_f64_point=rel_max(gh, gB[wk]),
                        grad_hip_vs_f64_at_hip_point=rel_max(gh, gA[wk]))
        v = split[i]
        print(f"(4) ray {i:5d} {wk:30s} raw HIP vs fp64 {v['raw_hip_vs_f64']:.1e};  composite adjoint: kernel vs fp64 at the SAME raw {v['adj_kernel_vs_f64_same_point']:.1e}, "
              f"fp64 at HIP raw vs fp64 at fp64 raw {v['adj_f64_hip_point_vs_f64_point']:.1e};  parameter gradient: HIP vs fp64 at the fp64 point {v['grad_hip_vs_f64_at_f64_point']:.1e}, "
              f"vs fp64 with the adjoint taken at the HIP raw {v['grad_hip_vs_f64_at_hip_point']:.1e}", flush=True)
    out["worst_rays_split"] = split
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("seeds", nargs="+", type=int)
    ap.add_argument("--force", default="")
    ap.add_argument("--rays", type=int, default=10)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    force = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.force.split(",") if kv}
    res = [diag(s, force, a.rays) for s in a.seeds]
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)
