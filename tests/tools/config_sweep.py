"""Development aid: throughput of the other BASELINE configs on one GPU (C3 train, C4 per-GPU shard, C5 eval)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfnerf_amd
from cfnerf_amd import train as TR, evaluate as E
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

def train_cfg(name, W, K, N, ndc, near, far, ha=32, iters=20):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=ha)
    _, kw, _, model, _, _ = build_model(cfg, 0, no_ndc=not ndc)
    rng = np.random.default_rng(0)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    rays = rays.cuda(); target = torch.rand(N, 3, device="cuda")
    tr = TR.Trainer(model, beta1=0.01)
    def step():
        tr.step(H, Wd, focal, rays, target, t_rand=torch.rand(N, 128, device="cuda"), eps=torch.randn(K, 4, device="cuda"), near=near, far=far, ndc=ndc)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / iters
    print(f"{name}: train W={W} K={K} N={N}: {dt*1e3:.2f} ms/step  {N/dt:,.0f} rays/s  stash {tr.net and 0 or 0}", flush=True)

def eval_cfg(name, W, K, H, Wd, focal, iters=3):
    cfg = O.OracleCfg(netwidth=W, K_samples=K)
    _, kw, kwt, model, _, _ = build_model(cfg, 0, white_bkgd=True, no_ndc=True)
    th, ph = np.deg2rad(30.0), np.deg2rad(-30.0)
    c2w = torch.tensor([[np.cos(th), -np.sin(th) * np.sin(ph), np.sin(th) * np.cos(ph), 4 * np.sin(th) * np.cos(ph)],
                        [0, np.cos(ph), np.sin(ph), 4 * np.sin(ph)],
                        [-np.sin(th), -np.cos(th) * np.sin(ph), np.cos(th) * np.cos(ph), 4 * np.cos(th) * np.cos(ph)]], dtype=torch.float32)
    for fused in (True, False):
        def run():
            if fused:
                return E.render_uncertainty(H, Wd, focal, c2w, model, near=2., far=6., ndc=False, white_bkgd=True)
            with torch.no_grad():
                return cfnerf_amd.render(H, Wd, focal, c2w=c2w, near=2., far=6., **kwt)
        run(); torch.cuda.synchronize(); t0 = time.time()
        for _ in range(iters): run()
        torch.cuda.synchronize(); dt = (time.time() - t0) / iters
        print(f"{name}: eval {H}x{Wd} K={K} {'fused K-stats only' if fused else 'per-K maps'}: {dt*1e3:.1f} ms/image  {H*Wd/dt:,.0f} rays/s", flush=True)

if __name__ == "__main__":
    train_cfg("C2", 256, 4, 1024, True, 0., 1.)
    train_cfg("C3 (africa-like)", 256, 8, 4096, False, 1.2, 8.0)
    train_cfg("C4 (per-GPU shard)", 256, 16, 1024, True, 0., 1.)
    train_cfg("authors' W=512 K=32 N=512", 512, 32, 512, False, 1.2, 8.0, ha=64, iters=5)
    eval_cfg("C5", 256, 32, 800, 800, 1111.1)
