"""One-off race soak: the same train step (forward + loss + backward, identical inputs) N times per configuration; every gradient,
output and scalar must be BIT-identical to the first run (no float atomics anywhere: a difference is a race).  Also the eval render.
    python tests/tools/repro_soak.py [iterations=300]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfnerf_amd
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for W, K, N, prec in [(256, 4, 1024, "fp32"), (256, 16, 1000, "fp32"), (512, 8, 512, "fp32"), (128, 3, 777, "fp32"), (64, 5, 2000, "fp32"), (320, 4, 600, "fp32"),
                      (256, 4, 1024, "bf16x3"), (512, 4, 300, "bf16x3")]:
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    _, kw_train, kw_test, model, _, _ = build_model(cfg, 5)
    net = model.module
    net.set_precision(prec)
    rng = np.random.default_rng(W + K)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    rays = rays.cuda()
    target = torch.tensor(rng.uniform(0, 1, (N, 3)), dtype=torch.float32).cuda()
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32).cuda()
    eps = torch.tensor(rng.standard_normal((K, 4)), dtype=torch.float32).cuda()
    tr = TR.Trainer(net, beta1=0.01)
    ref = None
    diffs = 0
    for i in range(iters):
        g = tr.forward_backward(H, Wd, focal, rays, target, t_rand=t_rand, eps=eps).clone()
        cur = (g, tr.rgb_map.clone(), tr.scalars.clone())
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
            diffs += 1
    with torch.no_grad():
        e0 = cfnerf_amd.render(H, Wd, focal, rays=rays, **kw_test)[0].clone()
        ed = sum(0 if torch.equal(e0, cfnerf_amd.render(H, Wd, focal, rays=rays, **kw_test)[0]) else 1 for _ in range(iters // 3))
    print(f"W={W} K={K} N={N} {prec}: {iters} train steps, {diffs} differ from the first; {iters // 3} eval renders, {ed} differ", flush=True)
    bad += diffs + ed
    net.release_workspace()
print("repro soak:", "OK" if bad == 0 else f"{bad} NON-REPRODUCIBLE RESULTS")
sys.exit(1 if bad else 0)
