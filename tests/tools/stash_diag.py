"""Development aid: compare the stashed activations / backward intermediates with the oracle's autograd."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.nn.functional as F
import cfnerf_amd
from cfnerf_amd import train as TR, _lib as L
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

def stash(net, name, layer, n):
    import hooks
    lib = hooks.lib()
    lib.cfnerf_debug_copy_stash.restype = C.c_int64
    lib.cfnerf_debug_copy_stash.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    out = torch.empty(n, device="cuda")
    r = lib.cfnerf_debug_copy_stash(net.handle, name.encode(), layer, C.c_void_p(out.data_ptr()), n, None)
    assert r == n, (name, r, n)
    torch.cuda.synchronize()
    return out.cpu()

def run(W, K, N, beta1=0.05):
    cfg = O.OracleCfg(netwidth=W, K_samples=K)
    _, kw_train, _, model, p, _ = build_model(cfg, 500 + W + K)
    net = model.module
    rng = np.random.default_rng(W + N)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
    target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32)
    tr = TR.Trainer(net, beta1=beta1)
    tr.forward_backward(H, Wd, focal, rays.cuda(), target.cuda(), t_rand=t_rand.cuda(), eps=torch.cat([er, ea], -1).cuda())
    P = N * 128
    # oracle in fp64 with retained intermediate grads
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
    packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.).double()
    ro, rd, vd = packed[:, 0:3], packed[:, 3:6], packed[:, 8:11]
    z = O.sample_z(packed[:, 6:7], packed[:, 7:8], O.t_vals_table(torch.float64), False, t_rand.double())
    pts = ro[..., None, :] + rd[..., None, :] * z[..., :, None]
    x = torch.cat([O.embed(pts.reshape(-1, 3), 10), O.embed(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
    ha, hr, acts = O.mlp_encode(p64, x, cfg, return_acts=True)
    for a in acts["trunk"] + [acts["feature"], acts["views"]]:
        a.retain_grad()
    ha.retain_grad(); hr.retain_grad()
    # rest of forward by hand (same as nerf_flows_forward but reusing ha/hr)
    BN = P
    eaD, erD = ea.double(), er.double()
    a_mean, a_std = p64["alpha_mean"], p64["alpha_std"]
    alpha0 = (eaD[None].expand(BN, K, 1) * a_std[None, None, :] + a_mean[None, None, :]).reshape(-1, 1)
    rgb0 = (erD[None].expand(BN, K, 3) * p64["rgb_std"][None, None, :] + p64["rgb_mean"][None, None, :]).reshape(-1, 3)
    haK = ha[:, None, :].expand(BN, K, 32).reshape(-1, 32); hrK = hr[:, None, :].expand(BN, K, 64).reshape(-1, 64)
    z_a, ld_a = O.sylvester_flow(p64, "flows_alpha", alpha0, haK, 4, False)
    z_r, ld_r = O.sylvester_flow(p64, "flows_rgb", rgb0, hrK, 4, False)
    zka, zkr = z_a.reshape(BN, K, 1), z_r.reshape(BN, K, 3)
    raw = torch.cat([zkr, zka], -1)
    ld_a = ld_a.reshape(BN, K) + (zka.sum(-1) - F.softplus(zka).sum(-1))
    ld_r = ld_r.reshape(BN, K) + (zkr.sum(-1) - 2 * F.softplus(zkr).sum(-1))
    a0 = alpha0.reshape(BN, K, 1); r0 = rgb0.reshape(BN, K, 3)
    base_a = -0.5 * (a_std.log() * 2 + (a0 - a_mean) ** 2 / a_std ** 2)
    base_r = -0.5 * (p64["rgb_std"].log() * 2 + (r0 - p64["rgb_mean"]) ** 2 / p64["rgb_std"] ** 2)
    ent = base_a.mean() - ld_a.mean() + base_r.mean() - ld_r.mean()
    rgb_map, disp, wts, depth = O.raw2outputs(raw.reshape(N, 128, K, 4), z, rd)
    Ls = O.train_loss(rgb_map, target.double(), ent, K, beta1)
    Ls["loss"].backward()
    def rel(a, b):
        b = b.detach().double().reshape(-1); a = a.double().reshape(-1)
        return float((a - b).abs().max() / (b.abs().max() + 1e-30)), int((a - b).abs().argmax())
    print(f"W={W}: tensor, rel max err (vs fp64 oracle), flat argmax, P*W={P*W}")
    for l in range(8):
        h = acts["trunk"][l]
        e1, i1 = rel(stash(net, "h", l, P * W), h)
        gpre = h.grad * (h > 0)
        e2, i2 = rel(stash(net, "g_h", l, P * W), gpre)
        print(f"  h[{l}] {e1:.2e} @row {i1 // W}   g_h[{l}] {e2:.2e} @row {i2 // W} col {i2 % W}")
    print("  feat", rel(stash(net, "feat", 0, P * W), acts["feature"]), " g_feat", rel(stash(net, "g_feat", 0, P * W), acts["feature"].grad))
    v = acts["views"]
    print("  v", rel(stash(net, "v", 0, P * W // 2), v), " g_v", rel(stash(net, "g_v", 0, P * W // 2), v.grad * (v > 0)))
    print("  ha", rel(stash(net, "ha", 0, P * 32), ha), " g_ha", rel(stash(net, "g_ha", 0, P * 32), ha.grad))
    print("  hr", rel(stash(net, "hr", 0, P * 64), hr), " g_hr", rel(stash(net, "g_hr", 0, P * 64), hr.grad))
    # row-wise error profile of g_h[6]
    l = 6 if W == 256 else 7
    h = acts["trunk"][l]; gpre = (h.grad * (h > 0)).detach()
    g = stash(net, "g_h", l, P * W).double().reshape(P, W)
    rowerr = (g - gpre).abs().max(1).values / gpre.abs().max()
    bad = (rowerr > 1e-5).nonzero().reshape(-1)
    print(f"  g_h[{l}]: {bad.numel()} rows with rel err > 1e-5; first rows: {bad[:20].tolist()}; row%64: {sorted(set((bad % 64).tolist()))[:40]}")
    if bad.numel():
        r = int(bad[0]); cols = ((g[r] - gpre[r]).abs() > 1e-6 * gpre.abs().max()).nonzero().reshape(-1)
        print(f"  row {r}: {cols.numel()} bad cols, e.g. {cols[:16].tolist()}; hip {g[r, cols[:4]].tolist()} ref {gpre[r, cols[:4]].tolist()}")
        hm = stash(net, "h", l, P * W).reshape(P, W)
        print(f"  mask agreement on that row: hip(h>0) vs ref(h>0) mismatches: {int(((hm[r] > 0) != (h[r] > 0)).sum())}")

if __name__ == "__main__":
    run(256, 4, 48)
    run(64, 4, 32)
