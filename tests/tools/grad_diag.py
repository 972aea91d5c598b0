"""Development aid: per-tensor gradient error of the HIP train step vs the fp32 and fp64 CPU oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfnerf_amd
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

def run(W, K, N, beta1=0.05):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    _, kw_train, _, model, p, _ = build_model(cfg, 500 + W + K)
    net = model.module
    rng = np.random.default_rng(W + N)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
    target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32)
    tr = TR.Trainer(net, beta1=beta1)
    grad = tr.forward_backward(H, Wd, focal, rays.cuda(), target.cuda(), t_rand=t_rand.cuda(), eps=torch.cat([er, ea], -1).cuda()).cpu()
    packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.)
    scal, g32, _ = O.train_step(p, packed, target, cfg, ea, er, t_rand, beta1)
    p64 = {k: v.double() for k, v in p.items()}
    scal64, g64, _ = O.train_step(p64, packed.double(), target.double(), cfg, ea.double(), er.double(), t_rand.double(), beta1)
    print(f"W={W} K={K} N={N} loss hip {float(tr.scalars[0]):.7f} cpu32 {scal['loss']:.7f} cpu64 {scal64['loss']:.7f}")
    print(f"{'param':34s} {'max|g64|':>10s} {'hip-64':>9s} {'cpu32-64':>9s} {'hip-cpu32':>9s}   (errors relative to max|g64|)")
    for key, (off, cnt) in net.layout.items():
        if g64[key] is None:
            continue
        r = g64[key].numpy().reshape(-1); s = np.abs(r).max() + 1e-30
        h = grad[off:off + cnt].double().numpy(); c = g32[key].double().numpy().reshape(-1)
        print(f"{key:34s} {s:10.3e} {np.abs(h - r).max() / s:9.2e} {np.abs(c - r).max() / s:9.2e} {np.abs(h - c).max() / s:9.2e}")

if __name__ == "__main__":
    run(256, 4, 48)
    run(64, 4, 32)
