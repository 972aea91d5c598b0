"""PSNR on a PROCEDURAL scene (synthetic stand-in: no LLFF-fern data exists in this environment).

Analytic emissive-absorbing Gaussian blobs are rendered by dense quadrature (torch, data generation only) into
20 views on a sphere (17 train / 3 held out, like llffhold=8); the HIP path trains on them through the device ray
pool + fused Trainer and is evaluated with the fused uncertainty render.  Prints one JSON line per precision mode."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cfnerf_amd
from cfnerf_amd import train as TR, evaluate as E
from oracle import cfnerf_oracle as O            # deterministic weight generator only
from util_hip import build_model

sys.path.insert(0, os.path.join(ROOT, "tools"))
import procedural_scene as PS

DEV = "cuda"
H, W, FOCAL, NEAR, FAR = PS.H, PS.W, PS.FOCAL, PS.NEAR, PS.FAR
pose_spherical = PS.pose_spherical


def scene(rng, n_blobs=6):
    return PS.blobs(rng, DEV, n_blobs)


def render_truth(sc, c2w, n_quad=512):
    return PS.render_truth(sc, c2w, DEV, n_quad)


def psnr_of(model, poses, images, idx):
    return PS.held_out_psnr(model, poses, images, idx, DEV)


def main(steps=3000, N_rand=1024, K=4):
    rng = np.random.default_rng(7)
    sc = scene(rng)
    poses = [pose_spherical(th, -20.0 - 10.0 * (i % 3), 4.0) for i, th in enumerate(np.linspace(-60, 60, 20))]
    images = torch.stack([render_truth(sc, p) for p in poses]).cpu()
    i_test = [2, 10, 18]
    i_train = [i for i in range(20) if i not in i_test]
    poses_t = torch.stack(poses)
    for mode in ("fp32", "bf16x3"):
        cfg = O.OracleCfg(netwidth=256, K_samples=K)
        torch.manual_seed(0)
        _, kw_train, kw_test, model, _, _ = build_model(cfg, 0, no_ndc=True)
        net = model.module
        net.reset_parameters()                      # nn.Linear-style init (the parity tests use the oracle's generator)
        net.set_precision(mode)
        pool = cfnerf_amd.RayPool(images, poses_t, H, W, FOCAL, i_train, N_rand, generator=torch.Generator(device=DEV).manual_seed(1))
        tr = TR.Trainer(net, lrate=5e-4, lrate_decay=250, beta1=0.01)
        g = torch.Generator(device=DEV).manual_seed(2)
        curve = {0: psnr_of(model, poses, images, i_test)}
        torch.cuda.synchronize(); t0 = time.time()
        for it in range(1, steps + 1):
            rays, target = pool.next_batch()
            tr.step(H, W, FOCAL, rays, target.contiguous(), t_rand=torch.rand(N_rand, 128, device=DEV, generator=g),
                    eps=torch.randn(K, 4, device=DEV, generator=g), near=NEAR, far=FAR, ndc=False)
            if it in (250, 1000, steps):
                torch.cuda.synchronize()
                curve[it] = psnr_of(model, poses, images, i_test)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(json.dumps({"scene": "procedural gaussian blobs (synthetic stand-in for LLFF-fern)", "precision": mode, "views": "17 train / 3 held out",
                          "image": [H, W], "N_rand": N_rand, "K": K, "steps": steps, "held_out_psnr_by_step": curve,
                          "train_psnr_last_batch": float(tr.scalars[3]), "wall_s_incl_eval": round(dt, 2)}), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)
