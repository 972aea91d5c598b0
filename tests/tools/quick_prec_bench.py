"""Development aid: fused forward timing in both arithmetic modes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfnerf_amd
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays
for W, K, N in ((256, 4, 1024), (256, 4, 16384), (256, 32, 8192), (512, 32, 4096)):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, 1)
    rays, (H, Wd, focal) = fern_rays(np.random.default_rng(0), N)
    rays = rays.cuda()
    for mode in ("fp32", "bf16x3"):
        model.module.set_precision(mode)
        with torch.no_grad():
            for _ in range(3): cfnerf_amd.render(H, Wd, focal, rays=rays, **kw_test)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(10): cfnerf_amd.render(H, Wd, focal, rays=rays, **kw_test)
            torch.cuda.synchronize(); dt = (time.time() - t0) / 10
        fl = N * 128 * (2 * (609152 + 4992) if W == 256 else 2 * (2348800 + 5376))
        print(f"W={W} K={K} N={N} {mode:7s}: {dt*1e3:8.3f} ms  {N/dt:12,.0f} rays/s  {fl/dt/1e12:7.1f} TFLOP/s (fp32-equivalent)", flush=True)
