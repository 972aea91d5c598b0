import ctypes as C, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfnerf_amd
from cfnerf_amd import _lib as L
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays
cfg = O.OracleCfg(netwidth=256, K_samples=4)
_, kw_train, kw_test, model, _, _ = build_model(cfg, 0)
rays, (H, W, focal) = fern_rays(np.random.default_rng(0), 1024)
rays = rays.cuda()
if os.environ.get("TL_PREC"): model.module.set_precision(os.environ["TL_PREC"])
TRAIN = os.environ.get("TL_MODE", "eval") == "train"
if TRAIN:
    from cfnerf_amd import train as TR
    tr = TR.Trainer(model.module, beta1=0.01)
    target = torch.rand(1024, 3, device="cuda")
for _ in range(3):
    if TRAIN: tr.forward_backward(H, W, focal, rays, target)
    else:
        with torch.no_grad(): cfnerf_amd.render(H, W, focal, rays=rays, **kw_test)
torch.cuda.synchronize()
print("variant:", "train" if TRAIN else "eval")
lib = L.lib()
big = (C.c_ulonglong * 4096)()
lib.cfnerf_debug_read_dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.cfnerf_debug_read_dbg(big, 4096) == 0
m = np.ctypeslib.as_array(big).astype(np.int64)
t = m[2048:2048 + 2 + 8 * 5].astype(np.float64); t = (t - t[0]) / 100.0
d = np.diff(t)
print("sampling->encode", d[0])
tr = d[1:1 + 40].reshape(8, 5) if len(d) >= 41 else None
lay = np.diff(np.concatenate([[t[1]], t[2:42]])).reshape(8, 5)
print("per layer [wait, mfma(wave0), barrier1 wait, store(wave0), barrier2 wait] us:")
print(np.round(lay, 2))
print("mean", np.round(lay[1:].mean(0), 2))

# every recorded tile of the two sampled workgroups (47 marks per tile with CFN_TIMESTAMP_FINE and D = 8):
# [sampling, encode, sum over layers of (wait, mfma wave0, barrier1, store wave0, barrier2), heads, views, h_rgb, theta, flows] in us
for sel in range(2):
    mk = m[2048 + sel * 700: 2048 + (sel + 1) * 700].astype(np.float64)
    n = int((mk > 0).sum())
    mk = mk[:n] / 100.0
    print("workgroup", sel, "marks", n)
    prev_end = None
    for t0 in range(0, n - 46, 47):
        q = mk[t0:t0 + 47]
        lay = np.diff(q[1:42]).reshape(8, 5)
        rest = np.diff(q[41:47])
        gap = (q[0] - prev_end) if prev_end is not None else 0.0
        prev_end = q[46]
        print("  tile", t0 // 47, "samp+gap %.1f" % gap, "enc %.1f" % (q[1] - q[0]), "layers[wait mfma b1 store b2]", np.round(lay.sum(0), 1),
              "heads views hrgb theta flows", np.round(rest, 1), "total %.1f" % (q[46] - q[0] + gap))
