"""Quick timing of the fused forward (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfnerf_amd
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

def run(W, K, N, train, iters=10):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, 1)
    rng = np.random.default_rng(0)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    rays = rays.cuda()
    t_rand = torch.rand(N, 128, device="cuda")
    kw = kw_train if train else kw_test
    with torch.no_grad():
        for _ in range(3):
            cfnerf_amd.render(H, Wd, focal, rays=rays, t_rand=t_rand if train else None, **kw)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(iters):
            cfnerf_amd.render(H, Wd, focal, rays=rays, t_rand=t_rand if train else None, **kw)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / iters
    flops = N * 128 * (2 * (609152 + 4992) if W == 256 else 2 * (2348800 + 5376))
    print(f"W={W} K={K} N={N} train={train}: {dt*1e3:.3f} ms  {N/dt:,.0f} rays/s  {flops/dt/1e12:.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    run(256, 4, 1024, False); run(256, 4, 1024, True); run(256, 4, 8192, False); run(256, 32, 8192, False)
    run(512, 32, 4096, False); run(256, 4, 65536, False)
