"""Development aid: where does the alpha-path gradient error of an ill-conditioned case come from?  Splits the backward at `raw`
(the unfused seam): d loss / d raw from cfnerf_composite_bwd vs the fp64 oracle, then cfnerf_network_bwd fed with the ORACLE's
d_raw, so the composite adjoint and the flow adjoint / reductions are judged separately."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfnerf_amd
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays, hip_relu_masks

S, W, K, N = 100, 128, 3, 14
cfg = O.OracleCfg(netwidth=W, K_samples=K)
_, kw_train, _, model, p, _ = build_model(cfg, 40 + S)
net = model.module
rng = np.random.default_rng(S * 7 + W)
rays, (H, Wd, focal) = fern_rays(rng, N)
tv = torch.sort(torch.tensor(rng.uniform(0, 1, S), dtype=torch.float32)).values
t_rand = torch.tensor(rng.uniform(0, 1, (N, S)), dtype=torch.float32)
ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32)
beta1 = 0.02
packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.)
tr = TR.Trainer(net, beta1=beta1)
g_fused = tr.forward_backward(H, Wd, focal, rays.cuda(), target.cuda(), t_rand=t_rand.cuda(), eps=torch.cat([er, ea], -1).cuda(), t_vals=tv.cuda()).cpu().clone()
acts, masks = hip_relu_masks(net, N * S)


def oracle(dt):
    d = lambda t: t.to(dt)
    q = {k: d(v).clone().requires_grad_(True) for k, v in p.items()}
    with O.relu_override(masks=masks):
        ret = O.render_rays(q, d(packed), cfg, d(ea), d(er), True, d(t_rand), False, False, t_vals=d(tv))
    ret["raw"].retain_grad()
    L = O.train_loss(ret["rgb_map"], d(target), ret["loss_entropy"], K, beta1)
    L["loss"].backward()
    return q, ret


q64, r64 = oracle(torch.float64)
q32, r32 = oracle(torch.float32)
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())

# d_raw: oracle32 vs 64
print("d_raw cpu32-vs-64 (rel to max):", rel(r32["raw"].grad, r64["raw"].grad), " alpha channel:", rel(r32["raw"].grad[..., 3], r64["raw"].grad[..., 3]))
# HIP composite_bwd on the HIP raw with the oracle64's d_rgb
rawh = r32["raw"].detach().cuda().requires_grad_(True)           # same raw on both sides: isolates the adjoint

zv = O.sample_z(packed[:, 6:7], packed[:, 7:8], tv, False, t_rand)
outs = cfnerf_amd.raw2outputs(rawh, zv.cuda(), packed[:, 3:6].cuda())
# cotangent of rgb_map from the fp64 loss
rgb64 = r64["rgb_map"].detach().clone().requires_grad_(True)
L = O.train_loss(rgb64, target.double(), r64["loss_entropy"].detach(), K, beta1)["loss"]
(G,) = torch.autograd.grad(L, rgb64)
(d_raw_h,) = torch.autograd.grad((outs[0] * G.float().cuda()).sum(), rawh)
# oracle64 composite adjoint on the SAME raw
r_in = r32["raw"].detach().double().requires_grad_(True)
o64 = O.raw2outputs(r_in, zv.double(), packed[:, 3:6].double())
(d_raw_64,) = torch.autograd.grad((o64[0] * G).sum(), r_in)
r_in32 = r32["raw"].detach().clone().requires_grad_(True)
o32 = O.raw2outputs(r_in32, zv, packed[:, 3:6])
(d_raw_32,) = torch.autograd.grad((o32[0] * G.float()).sum(), r_in32)
for c, name in ((slice(0, 3), "rgb"), (3, "alpha")):
    print(f"composite adjoint, {name} channels: hip-vs-64 {rel(d_raw_h.cpu()[..., c], d_raw_64[..., c]):.2e}   cpu32-vs-64 {rel(d_raw_32[..., c], d_raw_64[..., c]):.2e}"
          f"   sum over points: hip {float(d_raw_h.cpu()[..., c].double().sum()):.6e} cpu32 {float(d_raw_32[..., c].double().sum()):.6e} f64 {float(d_raw_64[..., c].sum()):.6e}")
# network backward fed with the oracle's exact d_raw
x = None
net.flat.grad = None
pts = packed[:, None, 0:3] + packed[:, None, 3:6] * zv[..., None]
e = torch.cat([O.embed(pts.reshape(-1, 3), 10), O.embed(packed[:, None, 8:11].expand(pts.shape).reshape(-1, 3), 4)], -1)
raw_n, ent_n = net(e.cuda(), False, False, eps_alpha=ea, eps_rgb=er)
(raw_n * d_raw_64.float().reshape(raw_n.shape).cuda()).sum().backward()
gn = net.flat.grad.cpu()
# oracle: same cotangent through the network only
def net_oracle(dt):
    qq = {k: v.to(dt).clone().requires_grad_(True) for k, v in p.items()}
    with O.relu_override(masks=masks):
        rw, en = O.nerf_flows_forward(qq, e.to(dt), ea.to(dt), er.to(dt), cfg, False)
    (rw * d_raw_64.to(dt).reshape(rw.shape)).sum().backward()
    return qq
n64, n32 = net_oracle(torch.float64), net_oracle(torch.float32)
print("network backward with the oracle's d_raw (no entropy):")
for key in ("alpha_mean", "alpha_std", "rgb_mean", "flows_alpha.amor_diag1.0.bias", "flows_alpha.amor_b.bias", "h_alpha_linear.bias", "flows_rgb.amor_b.bias", "h_rgb_linear.bias"):
    off, cnt = net.layout[key]
    print(f"  {key:34s} hip-vs-64 {rel(gn[off:off + cnt].reshape(n64[key].grad.shape), n64[key].grad):.2e}   cpu32-vs-64 {rel(n32[key].grad, n64[key].grad):.2e}")
print("full fused step:")
for key in ("alpha_mean", "alpha_std", "flows_alpha.amor_diag1.0.bias", "h_alpha_linear.bias"):
    off, cnt = net.layout[key]
    print(f"  {key:34s} hip-vs-64 {rel(g_fused[off:off + cnt].reshape(q64[key].grad.shape), q64[key].grad):.2e}   cpu32-vs-64 {rel(q32[key].grad, q64[key].grad):.2e}")
# ---- which input of the fused backward carries the error?
print("d_rgb (loss kernel) vs fp64:", rel(tr.d_rgb.cpu(), G), "  per-ray worst relative:", float(((tr.d_rgb.cpu().double() - G).abs().amax((1, 2)) / G.abs().amax((1, 2))).max()))
G32 = None
rgb32 = r32["rgb_map"].detach().clone().requires_grad_(True)
(G32,) = torch.autograd.grad(O.train_loss(rgb32, target, r32["loss_entropy"].detach(), K, beta1)["loss"], rgb32)
print("d_rgb cpu32 vs fp64:", rel(G32, G), "  per-ray worst relative:", float(((G32.double() - G).abs().amax((1, 2)) / G.abs().amax((1, 2))).max()))
print("rgb_map hip vs 64:", rel(tr.rgb_map.cpu(), r64["rgb_map"]), " cpu32 vs 64:", rel(r32["rgb_map"], r64["rgb_map"]))
# the fused backward driven by the fp64 loss gradient instead of the loss kernel's
import ctypes as C
from cfnerf_amd import _lib as L
lib = L.lib()
tr.forward_backward(H, Wd, focal, rays.cuda(), target.cuda(), t_rand=t_rand.cuda(), eps=torch.cat([er, ea], -1).cuda(), t_vals=tv.cuda())
gen = lib.cfnerf_model_stash_generation(net.handle)
for name, cot in (("loss kernel d_rgb", tr.d_rgb), ("fp64 d_rgb", G.float().cuda().contiguous()), ("cpu32 d_rgb", G32.cuda().contiguous())):
    gout = torch.empty(net.n_params, device="cuda")
    L.check(lib.cfnerf_render_bwd(net.handle, gen, L.ptr(cot), None, L.ptr(tr.d_ent), L.ptr(gout), L.stream()), "bwd")
    gout = gout.cpu()
    msg = []
    for key in ("alpha_mean", "alpha_std", "flows_alpha.amor_diag1.0.bias", "h_alpha_linear.bias", "rgb_mean"):
        off, cnt = net.layout[key]
        msg.append(f"{key.split('.')[-2] if '.' in key else key}: {rel(gout[off:off + cnt].reshape(q64[key].grad.shape), q64[key].grad):.2e}")
    print(f"fused backward with {name:18s}:", "  ".join(msg))
rgbh = tr.rgb_map.detach().cpu().double().requires_grad_(True)
(Gh,) = torch.autograd.grad(O.train_loss(rgbh, target.double(), r64["loss_entropy"].detach(), K, beta1)["loss"], rgbh)
print("loss kernel d_rgb vs fp64 loss gradient AT THE HIP rgb_map:", rel(tr.d_rgb.cpu(), Gh), " cpu32's own:", rel(G32, torch.autograd.grad(O.train_loss((r := r32["rgb_map"].detach().double().requires_grad_(True)), target.double(), r64["loss_entropy"].detach(), K, beta1)["loss"], r)[0]))
gout = torch.empty(net.n_params, device="cuda")
L.check(lib.cfnerf_render_bwd(net.handle, gen, L.ptr(Gh.float().cuda().contiguous()), None, L.ptr(tr.d_ent), L.ptr(gout), L.stream()), "bwd")
gout = gout.cpu()
for key in ("alpha_mean", "alpha_std", "flows_alpha.amor_diag1.0.bias", "h_alpha_linear.bias", "rgb_mean"):
    off, cnt = net.layout[key]
    print(f"  fused backward with the fp64 loss gradient at the HIP rgb_map: {key:32s} {rel(gout[off:off + cnt].reshape(q64[key].grad.shape), q64[key].grad):.2e}")
