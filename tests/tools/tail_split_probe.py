"""Development aid: the fused tail kernel (adjoint of composite + flows in one wave per ray) against its two halves as separate
kernels (cfnerf_composite_bwd: one wave per ray, 70 registers; flows_bwd_kernel: lane = point), at BASELINE config sizes."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfnerf_amd
from cfnerf_amd import _lib as L, train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

lib = L.lib()


def ev_time(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (N, K, W) in ((1024, 4, 256), (1024, 16, 256), (512, 32, 512), (1024, 64, 256)):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    _, kw_train, _, model, p, _ = build_model(cfg, 1)
    net = model.module
    rays, (H, Wd, focal) = fern_rays(np.random.default_rng(0), N)
    rays = rays.cuda(); target = torch.rand(N, 3, device="cuda")
    tr = TR.Trainer(net, beta1=0.01)
    t_rand = torch.rand(N, 128, device="cuda"); eps = torch.randn(K, 4, device="cuda")
    tr.forward_backward(H, Wd, focal, rays, target, t_rand=t_rand, eps=eps)
    lib.cfnerf_timing_enable(net.handle, 1)
    tr.forward_backward(H, Wd, focal, rays, target, t_rand=t_rand, eps=eps); torch.cuda.synchronize()
    t_tail = lib.cfnerf_timing_last_ms(net.handle, 1) * 1e3
    lib.cfnerf_timing_enable(net.handle, 0)
    # the two halves on the same sizes
    S = 128
    raw = torch.randn(N, S, K, 4, device="cuda"); z = torch.sort(torch.rand(N, S, device="cuda"), -1).values; d = torch.randn(N, 3, device="cuda")
    d_rgb = torch.randn(N, 3, K, device="cuda"); d_raw = torch.empty_like(raw)
    t_comp = ev_time(lambda: lib.cfnerf_composite_bwd(L.ptr(raw), L.ptr(z), L.ptr(d), N, S, K, 0, L.ptr(d_rgb), None, None, None, L.ptr(d_raw), L.stream()))
    x = torch.randn(N * S, 90, device="cuda") * 0.5
    net.ensure_workspace(1, N * S, K)
    rawp = torch.empty(N * S, K, 4, device="cuda"); ent = torch.zeros(1, device="cuda")
    L.check(lib.cfnerf_network_fwd(net.handle, L.ptr(x), L.ptr(eps), N * S, K, L.F_TRAIN | L.F_STASH, L.ptr(rawp), L.ptr(ent), L.stream()), "fwd")
    gen = lib.cfnerf_model_stash_generation(net.handle)
    grad = torch.empty(net.n_params, device="cuda"); de = torch.tensor([0.01], device="cuda")
    lib.cfnerf_timing_enable(net.handle, 1)
    L.check(lib.cfnerf_network_bwd(net.handle, gen, L.ptr(d_raw.reshape(N * S, K, 4)), L.ptr(de), L.ptr(grad), L.stream()), "bwd"); torch.cuda.synchronize()
    t_flows = lib.cfnerf_timing_last_ms(net.handle, 1) * 1e3          # flows_bwd + reduce_gms
    lib.cfnerf_timing_enable(net.handle, 0)
    print(f"N={N} K={K} W={W}: fused tail (+reduce_gms) {t_tail:7.1f} us | composite_bwd {t_comp:7.1f} us + flows_bwd (+reduce_gms) {t_flows:7.1f} us = {t_comp + t_flows:7.1f} us")
    net.release_workspace(); del tr, net, model
