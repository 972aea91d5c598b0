"""Development aid: gradient of one narrow (1 x 8) weight-gradient tile vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays
W, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = O.OracleCfg(netwidth=W, K_samples=K)
_, kw, _, model, p, _ = build_model(cfg, 5)
net = model.module
rng = np.random.default_rng(1)
rays, (H, Wd, focal) = fern_rays(rng, N)
t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32); er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32)
tr = TR.Trainer(net, beta1=0.01)
g = tr.forward_backward(H, Wd, focal, rays.cuda(), target.cuda(), t_rand=t_rand.cuda(), eps=torch.cat([er, ea], -1).cuda()).cpu()
torch.cuda.synchronize()
packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.)
_, grads, _ = O.train_step(p, packed, target, cfg, ea, er, t_rand, 0.01)
for key, (off, cnt) in net.layout.items():
    if grads[key] is None: continue
    a, b = g[off:off + cnt].reshape(grads[key].shape), grads[key]
    print(f"{key:34s} |hip| {float(a.abs().max()):.3e} |ref| {float(b.abs().max()):.3e} maxdiff {float((a - b).abs().max()):.3e} ratio {float((a*b).sum()/(b*b).sum().clamp(min=1e-30)):.4f}")
