"""One-off: 60 create / train step / eval render / destroy cycles of models of three widths; the device memory free afterwards must equal
the free memory before (handles, workspaces, packed weights and event pools are all released).   python tests/tools/leak_check.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, gc
import cfnerf_amd
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays
import contextlib, io
def once(i):
    cfg = O.OracleCfg(netwidth=[64,128,256][i%3], K_samples=4)
    with contextlib.redirect_stdout(io.StringIO()):
        _, kw_train, kw_test, model, _, _ = build_model(cfg, i)
    rng = np.random.default_rng(i)
    rays, (H, W, focal) = fern_rays(rng, 256)
    tr = TR.Trainer(model.module, beta1=0.01)
    tr.step(H, W, focal, rays.cuda(), torch.rand(256, 3, device="cuda"))
    with torch.no_grad(): cfnerf_amd.render(H, W, focal, rays=rays.cuda(), **kw_test)
    torch.cuda.synchronize()
for i in range(3): once(i)
gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
f0 = torch.cuda.mem_get_info()[0]
for i in range(60): once(i)
gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
f1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MiB, after 60 create/train/eval/destroy cycles %.1f MiB, delta %.1f MiB" % (f0/2**20, f1/2**20, (f0-f1)/2**20))
