"""Development aid: per-workgroup start / end timestamps and placement (HW_ID, XCC_ID) of the fused forward kernel,
from a library built with -DCFN_TIMESTAMP (CFNERF_LIB=...).  Prints how co-resident workgroups pair up and overlap."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
TRAIN = os.environ.get("TL_MODE", "eval") == "train"          # TL_MODE=train: the stashing train variant (forward of a Trainer step)
if TRAIN:
    from cfnerf_amd import train as TR
    tr = TR.Trainer(model.module, beta1=0.01)
    target = torch.rand(1024, 3, device="cuda")
for _ in range(3):
    if TRAIN:
        tr.forward_backward(H, W, focal, rays, target)
    else:
        with torch.no_grad():
            cfnerf_amd.render(H, W, focal, rays=rays, **kw_test)
torch.cuda.synchronize()
print("variant:", "train (stash)" if TRAIN else "eval")
lib = L.lib()
n = 512
buf = (C.c_ulonglong * (4 * n))()
lib.cfnerf_debug_read_dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.cfnerf_debug_read_dbg(buf, 4 * n) == 0
a = np.ctypeslib.as_array(buf).reshape(n, 4).astype(np.int64)
t0 = a[:, 0].min()
start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0          # us (100 MHz clock)
hw, xcc = a[:, 2], a[:, 3] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 20 + cu
print("kernel span us", end.max(), "start spread", start.max())
from collections import defaultdict
d = defaultdict(list)
for b in range(n):
    d[int(key[b])].append(b)
sizes = defaultdict(int)
for k, v in d.items(): sizes[len(v)] += 1
print("workgroups per CU histogram:", dict(sizes), "distinct CUs", len(d))
pairs = [v for v in d.values() if len(v) == 2]
print("first pairs (blockIdx):", pairs[:8])
diff = np.array([abs(v[0] - v[1]) for v in pairs]); print("pair index distance: unique", np.unique(diff)[:10])
lo = np.array([min(end[v[0]], end[v[1]]) for v in pairs]); hi = np.array([max(end[v[0]], end[v[1]]) for v in pairs])
print("end of earlier WG of a pair (mean us)", lo.mean(), " later WG", hi.mean())
first = np.array([end[min(v)] for v in pairs]); second = np.array([end[max(v)] for v in pairs])
print("end of lower-index WG", first.mean(), " higher-index WG", second.mean())

# phase marks of the co-resident pair (WG 0, WG grid/2), per tile: sampling, encoding, 8 x [mfma start, mfma end,
# epilogue+barrier end], heads, views, h_rgb, theta, flows+composite  = 31 marks
big = (C.c_ulonglong * 4096)()
assert lib.cfnerf_debug_read_dbg(big, 4096) == 0
m = np.ctypeslib.as_array(big).astype(np.int64)
NM = 31
names = ["sampling", "encode"] + [f"L{l}.{x}" for l in range(8) for x in ("wait", "mfma", "epi")] + ["heads", "views", "h_rgb", "theta", "flows+comp"]
for nm, base in (("A (WG 0)", 2048), ("B (WG grid/2)", 2048 + 700)):
    t = m[base:base + 4 * NM].astype(np.float64)
    if t[0] <= 0:
        continue
    t = (t - t[0]) / 100.0
    d = np.diff(np.concatenate([[t[0]], t]))                       # duration that ENDS at each mark
    per = d.reshape(4, NM)
    print(nm, "tile period us:", [round(float(t[(i + 1) * NM - 1] - (t[i * NM - 1] if i else 0)), 1) for i in range(4)])
    avg = per[1:].mean(0)                                          # skip the first tile (cold)
    trunk = avg[2:26].reshape(8, 3)
    print("  sampling %.1f  encode %.1f | trunk per layer: wait %.1f mfma %.1f epi+barrier %.1f | heads %.1f views %.1f h_rgb %.1f theta %.1f flows+composite %.1f" %
          (avg[0], avg[1], trunk[:, 0].mean(), trunk[:, 1].mean(), trunk[:, 2].mean(), avg[26], avg[27], avg[28], avg[29], avg[30]))
    print("  trunk total %.1f  tail total %.1f  head(sampling+encode) %.1f" % (avg[2:26].sum(), avg[26:].sum(), avg[:2].sum()))
