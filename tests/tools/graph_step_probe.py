"""Development probe: is the train step launch-bound?  Times N eager Trainer.step calls against N replays of the same step captured
in a HIP graph (torch.cuda.CUDAGraph on the launch stream; explicit t_rand / eps, so nothing inside the capture touches the host)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

N, K = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = O.OracleCfg(netwidth=256, K_samples=K)
_, _, _, model, _, _ = build_model(cfg, 1)
rng = np.random.default_rng(0)
rays, (H, W, focal) = fern_rays(rng, N)
rays = rays.cuda()
target = torch.rand(N, 3, device="cuda")
t_rand = torch.rand(N, 128, device="cuda")
eps = torch.randn(K, 4, device="cuda")
tr = TR.Trainer(model, beta1=0.01)
step = lambda: tr.step(H, W, focal, rays, target, t_rand=t_rand, eps=eps)
for _ in range(5): step()
torch.cuda.synchronize()
def timed(fn, n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager  ms/step", round(timed(step), 4))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        step()
    print("graph  ms/step", round(timed(g.replay), 4))
    print("eager  ms/step", round(timed(step), 4))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
