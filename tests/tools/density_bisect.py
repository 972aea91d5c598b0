"""Development aid (round 6): WHERE does the fused backward lose precision on the density path of ONE ray?

    python tests/tools/density_bisect.py 5027:63,0 5102:64,0 [--json out.json]

The tail kernel is a function  (theta, raw, (e, T), z, rays, eps, base Gaussians, d loss / d rgb_map)  ->  g_theta [S,128], gms [8]  of
values the forward left in the stash.  For one ray with a one-hot cotangent this tool reads those inputs AND the kernel's outputs back
(tests/cfnerf_debug.h) and compares the density columns of g_theta (amor_diag1 / amor_diag2 / amor_b of flows_alpha: 12 columns) with
  E   : torch autograd in fp64 of the same function of the same fp32 inputs (the exact answer at the HIP forward point),
  T32 : torch autograd in fp32 of it (what the reference's arithmetic delivers),
per sample, as column sums (= the bias gradients: where a ray's terms cancel) and as the sums weighted with h_alpha (= the head weights).
Then the formula of rounds 1 - 5 (g T - suffix / x, the flow adjoint on the recomputed chain) is evaluated in fp64 with ONE ingredient at a
time replaced by what the kernel really has (the stashed fp32 e, the stashed fp32 T, fp32 colours, an fp32 suffix scan, ...): the
ingredient whose substitution reproduces the kernel's error is the defect - it was the product scan's T (profiles/r06_density_bisect.txt).
"alternative: ..." rows: candidate replacements fed the kernel's own stash; D32tree is what comp_adjoint_D (csrc/cfnerf_device.h) does now."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from cfnerf_amd import _lib as L
from util_hip import fuzz_case, stash_copy

F32, F64 = torch.float32, torch.float64
FORCE = {5101: dict(W=256, D=8, K=2, hr=64, S=64), 5102: dict(W=256, D=8, K=2, hr=64, S=64)}     # (the K2_DRAWS of tests/test_hip_train.py)


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def hip_bwd(net, d_rgb):
    lib = L.lib()
    gout = torch.empty(net.n_params, device="cuda")
    dd = d_rgb.float().cuda().contiguous()
    L.check(lib.cfnerf_render_bwd(net.handle, lib.cfnerf_model_stash_generation(net.handle), L.ptr(dd), None, None, L.ptr(gout), L.stream()), "cfnerf_render_bwd")
    torch.cuda.synchronize()
    return gout.cpu()


def read_ray(net, i, N, S, K):
    P, nch = N * S, (S + 63) // 64
    row = lambda name, w: stash_copy(net, name, 0, P * w, keep=(i * S * w, (i + 1) * S * w)).reshape(S, w)
    out = dict(theta=row("theta", 128), g_theta=row("g_theta", 128), z=row("z", 1).reshape(S))
    out["rays"] = stash_copy(net, "rays", 0, N * 11, keep=(i * 11, (i + 1) * 11))
    n_t = N * nch
    raw = stash_copy(net, "raw", 0, n_t * 64 * K * 4, keep=(i * nch * K * 256, (i + 1) * nch * K * 256))
    out["raw"] = raw.reshape(nch, K, 64, 4).permute(0, 2, 1, 3).reshape(nch * 64, K, 4)[:S]
    at = stash_copy(net, "at", 0, n_t * 64 * K * 2, keep=(i * nch * K * 128, (i + 1) * nch * K * 128))
    out["at"] = at.reshape(nch, K, 64, 2).permute(0, 2, 1, 3).reshape(nch * 64, K, 2)[:S]
    return out


def flow_chain(th_a, a0, tanh=torch.tanh):
    """alpha flows: th_a [S,12] = d1[4] | d2[4] | b[4] (diagonals post-tanh), a0 [K] -> a_4 [S,K] and the per-step (a_in, t)"""
    S = th_a.shape[0]
    a = a0[None, :].expand(S, a0.shape[0])
    keep = []
    for f in range(4):
        t = tanh(th_a[:, 4 + f, None] * a + th_a[:, 8 + f, None])
        keep.append((a, t))
        a = th_a[:, f, None] * t + a
    return a, keep


def density_loss(th_a, eps_a, mean, std, c, z, dnorm, G, wb):
    """sum(G * rgb_map) of one ray as a function of the density-flow parameters (colours c [S,K,3] are constants here)"""
    dt = th_a.dtype
    a, _ = flow_chain(th_a, eps_a * std + mean)
    dz = torch.cat([z[1:] - z[:-1], torch.tensor([1e1], dtype=dt)])
    dist = dz * dnorm
    alpha = 1.0 - torch.exp(-torch.nn.functional.softplus(a) * dist[:, None])
    x = 1.0 - alpha + 1e-10
    T = torch.cumprod(torch.cat([torch.ones(1, x.shape[1], dtype=dt), x], 0), 0)[:-1]
    w = alpha * T
    out = (w[:, :, None] * c).sum(0)                         # [K,3]
    if wb:
        out = out + (1.0 - w.sum(0))[:, None]
    return (out * G.T.to(dt)).sum(), dict(a=a, alpha=alpha, T=T, x=x, dist=dist, w=w)


def autograd_ref(d, G, eps_a, mean, std, wb, dt):
    th = d["theta"][:, 96:108].to(dt).clone().requires_grad_(True)
    m = torch.tensor(float(mean), dtype=dt, requires_grad=True)
    s = torch.tensor(float(std), dtype=dt, requires_grad=True)
    c = torch.sigmoid(d["raw"][:, :, :3].to(dt))
    rd = d["rays"][3:6].to(dt)
    dnorm = torch.sqrt((rd * rd).sum())
    Lv, fw = density_loss(th, eps_a.to(dt), m, s, c, d["z"].to(dt), dnorm, G, wb)
    gth, gm, gs = torch.autograd.grad(Lv, [th, m, s])
    thd = th.detach()
    g = gth.clone()
    g[:, :8] = g[:, :8] * (1.0 - thd[:, :8] ** 2)            # the diagonals chain through their tanh (MOD:341-348): g_theta is pre-tanh
    return g, float(gm), float(gs), {k: v.detach() for k, v in fw.items()}


def kernel_formula(d, G, eps_a, mean, std, wb, sub):
    """The tail kernel's own formula (cfnerf_tail.hip) in fp64, with the ingredients named in `sub` replaced by what the kernel has.
    sub: set of {"e", "T", "c32", "raw3", "suffix32", "comp32", "flow32"}"""
    dt = F64
    th = d["theta"][:, 96:108].to(dt)
    S, K = d["raw"].shape[0], d["raw"].shape[1]
    z = d["z"].to(dt)
    rd = d["rays"][3:6].to(dt)
    dnorm = torch.sqrt((rd * rd).sum())
    dz = torch.cat([z[1:] - z[:-1], torch.tensor([1e1], dtype=dt)])
    dist = dz * dnorm
    a0 = eps_a.to(dt) * std + mean
    a4, keep = flow_chain(th, a0)
    raw3 = d["raw"][:, :, 3].to(dt) if "raw3" in sub else a4     # the kernel takes softplus' from the STASHED raw
    sp = torch.nn.functional.softplus(a4)
    e_ex = torch.exp(-sp * dist[:, None])
    e = d["at"][:, :, 0].to(dt) if "e" in sub else e_ex
    alpha = 1.0 - e
    x = (1.0 - alpha) + 1e-10
    if "T" in sub:
        T = d["at"][:, :, 1].to(dt)
    else:
        xe = (1.0 - (1.0 - e_ex)) + 1e-10
        T = torch.cumprod(torch.cat([torch.ones(1, K, dtype=dt), xe], 0), 0)[:-1]
    c = torch.sigmoid(d["raw"][:, :, :3].to(dt))
    if "c32" in sub:
        c = torch.sigmoid(d["raw"][:, :, :3]).to(dt)
    Gt = G.T.to(dt)                                              # [K,3]
    if "comp32" in sub:                                          # the whole composite adjoint in fp32 operations (sequential suffix)
        f = lambda t: t.float()
        g = (f(Gt)[None] * f(c)).sum(-1)
        if wb:
            g = g - f(Gt).sum(-1)[None]
        w = f(alpha) * f(T)
        gw = g * w
        suffix = torch.cat([torch.flip(torch.cumsum(torch.flip(gw, [0]), 0), [0])[1:], torch.zeros(1, K)], 0)
        dalpha = g * f(T) - suffix / f(x)
        sg = torch.sigmoid(f(raw3))
        ga = (((dalpha * f(e)) * f(dist)[:, None]) * sg).to(dt)
    else:
        g = (Gt[None] * c).sum(-1)                               # [S,K]
        if wb:
            g = g - Gt.sum(-1)[None]
        w = alpha * T
        gw = g * w
        if "suffix32" in sub:
            gw32 = gw.float()
            suffix = torch.cat([torch.flip(torch.cumsum(torch.flip(gw32, [0]), 0), [0])[1:], torch.zeros(1, K)], 0).to(dt)
        else:
            suffix = torch.cat([torch.flip(torch.cumsum(torch.flip(gw, [0]), 0), [0])[1:], torch.zeros(1, K, dtype=dt)], 0)
        dalpha = g * T - suffix / x
        sg = torch.sigmoid(raw3)
        ga = dalpha * e * dist[:, None] * sg
    # flow adjoint (alpha part of flows_adjoint), on the chain recomputed from theta
    fd = F32 if "flow32" in sub else dt
    thf = th.to(fd)
    _, keepf = flow_chain(thf, a0.to(fd))
    gaf = ga.to(fd)
    gth = torch.zeros(S, 12, dtype=fd)
    for f in (3, 2, 1, 0):
        ai, t = keepf[f]
        d1, d2 = thf[:, f, None], thf[:, 4 + f, None]
        gth[:, f] = (gaf * t).sum(1)
        gpa = (gaf * d1) * (1.0 - t * t)
        gth[:, 8 + f] = gpa.sum(1)
        gth[:, 4 + f] = (gpa * ai).sum(1)
        gaf = gaf + gpa * d2
    gm, gs = gaf.sum(), (gaf * eps_a.to(fd)[None]).sum()
    gth = gth.to(dt)
    gth[:, :8] = gth[:, :8] * (1.0 - th[:, :8] ** 2)
    return gth, float(gm), float(gs), dict(ga=ga, dalpha=dalpha if "comp32" not in sub else dalpha.to(dt))


def alt_formula(d, G, eps_a, mean, std, wb, how):
    """Candidate replacements of the composite adjoint, fed what the kernel has (stashed fp32 e and T, the stashed raw, fp32 colours); the flow
    adjoint after it in fp64 (it is not the problem).  how:
      "That32"  : g T^ - suffix / x with T^ = the SEQUENTIAL fp32 running product of the x the forward used (torch's cumprod order)
      "D64"     : d alpha_s = T_s D_s,  D_s = (g_s - g_{s+1}) + x_{s+1} D_{s+1}  (D = g_s - R_{s+1}: the cancelled quantity carried itself), fp64
      "D32seq"  : the same in fp32 operations, sample after sample
      "D32tree" : the same in fp32 operations as a 64-lane Hillis-Steele suffix scan of affine maps per chunk + a carry (a wave's order)"""
    dt = F64
    th = d["theta"][:, 96:108].to(dt)
    S, K = d["raw"].shape[0], d["raw"].shape[1]
    z32 = d["z"]
    rd32 = d["rays"][3:6]
    dnorm32 = torch.sqrt((rd32[0] * rd32[0] + rd32[1] * rd32[1]) + rd32[2] * rd32[2])
    dist32 = torch.cat([z32[1:] - z32[:-1], torch.tensor([1e1])]) * dnorm32
    a0 = eps_a.to(dt) * std + mean
    e32, T32 = d["at"][:, :, 0], d["at"][:, :, 1]
    alpha32 = 1.0 - e32
    x32 = (1.0 - alpha32) + 1e-10
    c32 = torch.sigmoid(d["raw"][:, :, :3])
    Gt32 = G.T.float()
    g32 = ((Gt32[None, :, 0] * c32[:, :, 0] + Gt32[None, :, 1] * c32[:, :, 1]) + Gt32[None, :, 2] * c32[:, :, 2])
    if wb:
        g32 = g32 - ((Gt32[:, 0] + Gt32[:, 1]) + Gt32[:, 2])[None]
    sg32 = torch.sigmoid(d["raw"][:, :, 3])
    if how == "That32":
        That = torch.ones(S, K)
        for s in range(1, S):
            That[s] = That[s - 1] * x32[s - 1]
        gw = g32 * (alpha32 * That)
        suffix = torch.zeros(S, K)
        for s in range(S - 2, -1, -1):
            suffix[s] = suffix[s + 1] + gw[s + 1]
        dalpha = g32 * That - suffix / x32
    else:
        wd = dt if how == "D64" else F32
        g, x, T = g32.to(wd), x32.to(wd), T32.to(wd)
        gn = torch.cat([g[1:], torch.zeros(1, K, dtype=wd)], 0)              # g_{s+1}, 0 behind the last sample
        xn = torch.cat([x[1:], torch.zeros(1, K, dtype=wd)], 0)              # x_{s+1}
        b = (g - gn) - 1e-10 * gn                                            # (the 1e-10 of RUN:443: alpha_{s+1} = 1 - x_{s+1} + 1e-10)
        D = torch.zeros(S, K, dtype=wd)
        if how in ("D64", "D32seq"):
            nxt = torch.zeros(K, dtype=wd)
            for s in range(S - 1, -1, -1):
                D[s] = b[s] + xn[s] * nxt
                nxt = D[s]
        else:
            nch = (S + 63) // 64
            carry = torch.zeros(K, dtype=wd)
            for ch in range(nch - 1, -1, -1):
                lo, hi = ch * 64, min(S, ch * 64 + 64)
                A = torch.ones(64, K, dtype=wd); B = torch.zeros(64, K, dtype=wd)
                A[:hi - lo] = xn[lo:hi]; B[:hi - lo] = b[lo:hi]
                dd = 1
                while dd < 64:
                    A2, B2 = A.clone(), B.clone()
                    B2[:64 - dd] = B[:64 - dd] + A[:64 - dd] * B[dd:]
                    A2[:64 - dd] = A[:64 - dd] * A[dd:]
                    A, B = A2, B2
                    dd *= 2
                Dc = B + A * carry[None]
                D[lo:hi] = Dc[:hi - lo]
                carry = Dc[0]
        dalpha = T * D
    ga = (((dalpha.float() * e32) * dist32[:, None]) * sg32).to(dt) if how != "D64" else dalpha * e32.to(dt) * dist32.to(dt)[:, None] * sg32.to(dt)
    thf = th
    _, keepf = flow_chain(thf, a0)
    gaf = ga
    gth = torch.zeros(S, 12, dtype=dt)
    for f in (3, 2, 1, 0):
        ai, t = keepf[f]
        d1, d2 = thf[:, f, None], thf[:, 4 + f, None]
        gth[:, f] = (gaf * t).sum(1)
        gpa = (gaf * d1) * (1.0 - t * t)
        gth[:, 8 + f] = gpa.sum(1)
        gth[:, 4 + f] = (gpa * ai).sum(1)
        gaf = gaf + gpa * d2
    gm, gs = gaf.sum(), (gaf * eps_a.to(dt)[None]).sum()
    gth[:, :8] = gth[:, :8] * (1.0 - th[:, :8] ** 2)
    return gth, float(gm), float(gs)


def summarize(name, g, gm, gs, E, Em, Es, ha):
    """errors of a candidate (g [S,12], mean, std) against the exact (E, Em, Es), relative to the largest exact entry of each group"""
    per = rel(g, E)
    col = rel(g.double().sum(0), E.sum(0))
    wg = rel(torch.einsum("sc,sj->cj", g.double(), ha.double()), torch.einsum("sc,sj->cj", E, ha.double()))
    return dict(name=name, per_sample=per, bias_sums=col, head_weights=wg, alpha_mean=abs(gm - Em) / max(abs(Em), 1e-300), alpha_std=abs(gs - Es) / max(abs(Es), 1e-300))


def fmt(r):
    return (f"    {r['name']:44s} per-sample {r['per_sample']:.1e}  bias sums {r['bias_sums']:.1e}  head weights {r['head_weights']:.1e}  "
            f"alpha_mean {r['alpha_mean']:.1e}  alpha_std {r['alpha_std']:.1e}")


def diag(seed, rays_i, force):
    c = fuzz_case(seed, **force)
    net, N, S, K = c["net"], c["N"], c["S"], c["K"]
    print(f"==== seed {seed}: W={c['W']} D={c['D']} K={K} N={N} S={S} ndc={c['ndc']} wb={c['wb']} perturb={c['perturb']}", flush=True)
    flat = net.flat.detach().cpu()
    mean, std = float(flat[0]), float(flat[1])
    eps_a = c["ea"].reshape(K)
    P = N * S
    HA = c["ha"]
    res = []
    for i in rays_i:
        rng = np.random.default_rng(seed * 1000 + i)
        Gi = torch.tensor(rng.standard_normal((3, K)), dtype=F32)
        G = torch.zeros(N, 3, K)
        G[i] = Gi
        g_flat = hip_bwd(net, G)
        d = read_ray(net, i, N, S, K)
        ha = stash_copy(net, "ha", 0, P * HA, keep=(i * S * HA, (i + 1) * S * HA)).reshape(S, HA)
        E, Em, Es, fw = autograd_ref(d, Gi, eps_a, mean, std, c["wb"], F64)
        T32, Tm, Ts, _ = autograd_ref(d, Gi, eps_a, mean, std, c["wb"], F32)
        H = d["g_theta"][:, 96:108]
        off_m, _ = net.layout["alpha_mean"]
        off_s, _ = net.layout["alpha_std"]
        print(f"  ray {i}: stash e vs exact {rel(d['at'][:, :, 0], fw['alpha'].neg().add(1.0)):.1e}, T vs exact {rel(d['at'][:, :, 1], fw['T']):.1e} "
              f"(of the largest entry); raw3 stash vs exact chain {rel(d['raw'][:, :, 3], fw['a']):.1e}; opacity of the ray {float(1 - fw['x'].prod(0).min()):.4f}")
        rows = [summarize("HIP tail kernel (g_theta, flat gms)", H, float(g_flat[off_m]), float(g_flat[off_s]), E, Em, Es, ha),
                summarize("torch fp32 autograd of the same function", T32, Tm, Ts, E, Em, Es, ha)]
        for sub in ([], ["e"], ["T"], ["e", "T"], ["raw3"], ["c32"], ["e", "T", "raw3", "c32"], ["e", "T", "raw3", "c32", "suffix32"],
                    ["e", "T", "raw3", "c32", "comp32"], ["flow32"], ["e", "T", "raw3", "c32", "comp32", "flow32"]):
            g, gm, gs, _ = kernel_formula(d, Gi, eps_a, mean, std, c["wb"], set(sub))
            rows.append(summarize("kernel formula, fp64 but: " + ("nothing" if not sub else " ".join(sub)), g, gm, gs, E, Em, Es, ha))
        for how in ("That32", "D64", "D32seq", "D32tree"):
            g, gm, gs = alt_formula(d, Gi, eps_a, mean, std, c["wb"], how)
            rows.append(summarize("alternative: " + how, g, gm, gs, E, Em, Es, ha))
        # downstream of g_theta: the final flat gradients of the density heads against the fp64 sums of the kernel's OWN g_theta rows
        for key, cols in (("flows_alpha.amor_diag1.0.bias", slice(96, 100)), ("flows_alpha.amor_diag2.0.bias", slice(100, 104)), ("flows_alpha.amor_b.bias", slice(104, 108))):
            off, cnt = net.layout[key]
            own = d["g_theta"][:, cols].double().sum(0)
            print(f"    {key:34s} flat gradient vs fp64 sum of the kernel's own rows: {rel(g_flat[off:off + cnt], own):.1e}")
        for r in rows:
            print(fmt(r), flush=True)
        res.append(dict(seed=seed, ray=i, rows=rows))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="+", help="seed:ray,ray,...")
    ap.add_argument("--force", default="")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    force = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.force.split(",") if kv}
    out = []
    for case in a.cases:
        seed, rays = case.split(":")
        out += diag(int(seed), [int(x) for x in rays.split(",")], {**FORCE.get(int(seed), {}), **force})
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
