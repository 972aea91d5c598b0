"""PSNR-vs-step of the HIP path next to the CPU oracle, SAME scene / weights / batches / jitter / eps  (SURVEY 8d:
"train both reference (CPU, few hundred steps) and build, compare PSNR-vs-step curves").

The scene is the procedural stand-in of psnr_procedural.py (no LLFF-fern data exists here).  The two runs start from
identical weights and see identical rays, targets, t_rand and eps every step, so the only difference is the
arithmetic: the curves must coincide up to fp32 re-association noise amplified by training.  As the CONTROL for how
large that amplification is, the oracle is run a second time from weights perturbed by one fp32 ulp (x(1 +- 2^-23)):
the HIP curve should sit as close to the oracle as the oracle sits to its own perturbed twin.  The oracle runs on
the host cores (N_rand is reduced so a few hundred steps finish in minutes).  Prints one JSON line.

    python tests/tools/psnr_curve_vs_oracle.py [steps=200] [N_rand=256] > gpurun_out/psnr_curve.json
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import cfnerf_amd
from cfnerf_amd import train as TR
from oracle import cfnerf_oracle as O
from util_hip import build_model
import psnr_procedural as PP

DEV = "cuda"
H, W, FOCAL, NEAR, FAR = PP.H, PP.W, PP.FOCAL, PP.NEAR, PP.FAR


def oracle_psnr(p, cfg, pose, image, ea, er):
    with torch.no_grad():
        out = O.render(p, H, W, FOCAL, cfg, ea, er, False, c2w=pose, ndc=False, near=NEAR, far=FAR)
    return float(-10 * torch.log10(torch.mean((out["rgb_map"].mean(-1) - image) ** 2)))


def hip_psnr(kw_test, pose, image):
    kw = dict(kw_test); kw.update(near=NEAR, far=FAR, ndc=False)
    with torch.no_grad():
        rgb, _, _, _ = cfnerf_amd.render(H, W, FOCAL, c2w=pose.to(DEV), **kw)
    return float(-10 * torch.log10(torch.mean((rgb.mean(-1) - image.to(DEV)) ** 2)))


def main(steps=200, N_rand=256, K=4, every=25, threads=32):
    torch.set_num_threads(threads)
    rng = np.random.default_rng(7)
    sc = PP.scene(rng)
    poses = [PP.pose_spherical(th, -20.0 - 10.0 * (i % 3), 4.0) for i, th in enumerate(np.linspace(-60, 60, 20))]
    images = torch.stack([PP.render_truth(sc, p) for p in poses]).cpu()
    i_test = 10
    i_train = [i for i in range(20) if i not in (2, 10, 18)]
    cfg = O.OracleCfg(netwidth=256, K_samples=K)
    torch.manual_seed(0)
    _, _, kw_test, model, _, _ = build_model(cfg, 0, no_ndc=True)
    net = model.module
    net.reset_parameters()
    shapes = O.param_shapes(cfg)
    sd = net.state_dict()
    p = {k: sd[k].detach().cpu().clone().reshape(shapes[k]) for k in shapes}
    # the eval eps of the HIP module (R9: last row 0) drives both evaluations
    ev = net.eval_eps().cpu()                                           # [K,4] = rgb3, alpha1
    ea_eval, er_eval = ev[:, 3:4].clone(), ev[:, 0:3].clone()
    tr = TR.Trainer(net, lrate=5e-4, lrate_decay=250, beta1=0.01)
    state, state2 = {}, {}
    gp = torch.Generator().manual_seed(11)
    p2 = {k: v * (1 + (torch.randint(0, 2, v.shape, generator=gp).float() * 2 - 1) * 2.0 ** -23) for k, v in p.items()}
    g = torch.Generator().manual_seed(3)
    # all rays of the training views, host side (the oracle needs them on the CPU anyway)
    ro_all, rd_all, tg_all = [], [], []
    for v in i_train:
        ro, rd = O.get_rays(H, W, FOCAL, poses[v])
        ro_all.append(ro.reshape(-1, 3)); rd_all.append(rd.reshape(-1, 3)); tg_all.append(images[v].reshape(-1, 3))
    ro_all, rd_all, tg_all = torch.cat(ro_all), torch.cat(rd_all), torch.cat(tg_all)
    curve = {"step": [], "hip_psnr": [], "oracle_psnr": [], "oracle_1ulp_psnr": [], "hip_loss": [], "oracle_loss": [], "oracle_1ulp_loss": []}

    def record(it, lh, lo, lo2):
        curve["step"].append(it)
        curve["hip_psnr"].append(hip_psnr(kw_test, poses[i_test], images[i_test]))
        curve["oracle_psnr"].append(oracle_psnr(p, cfg, poses[i_test], images[i_test], ea_eval, er_eval))
        curve["oracle_1ulp_psnr"].append(oracle_psnr(p2, cfg, poses[i_test], images[i_test], ea_eval, er_eval))
        curve["hip_loss"].append(lh); curve["oracle_loss"].append(lo); curve["oracle_1ulp_loss"].append(lo2)

    record(0, None, None, None)
    t_cpu = 0.0
    for it in range(1, steps + 1):
        sel = torch.randint(0, ro_all.shape[0], (N_rand,), generator=g)
        ro, rd, tg = ro_all[sel], rd_all[sel], tg_all[sel]
        t_rand = torch.rand(N_rand, 128, generator=g)
        eps = torch.randn(K, 4, generator=g)
        sc_h = tr.step(H, W, FOCAL, (ro.to(DEV), rd.to(DEV)), tg.to(DEV).contiguous(), t_rand=t_rand.to(DEV), eps=eps.to(DEV),
                       near=NEAR, far=FAR, ndc=False)
        t0 = time.time()
        packed = O.pack_rays(H, W, FOCAL, ro, rd, False, NEAR, FAR)
        scal, grads, _ = O.train_step(p, packed, tg, cfg, eps[:, 3:4], eps[:, 0:3], t_rand, 0.01)
        p = O.adam_step(p, grads, state, it, O.lr_schedule(5e-4, 250, it - 1))
        t_cpu += time.time() - t0
        scal2, grads2, _ = O.train_step(p2, packed, tg, cfg, eps[:, 3:4], eps[:, 0:3], t_rand, 0.01)
        p2 = O.adam_step(p2, grads2, state2, it, O.lr_schedule(5e-4, 250, it - 1))
        if it % every == 0 or it == steps:
            record(it, float(sc_h[0]), scal["loss"], scal2["loss"])
    d = np.abs(np.array(curve["hip_psnr"]) - np.array(curve["oracle_psnr"]))
    d2 = np.abs(np.array(curve["oracle_1ulp_psnr"]) - np.array(curve["oracle_psnr"]))
    print(json.dumps({"scene": "procedural gaussian blobs (synthetic stand-in for LLFF-fern)", "N_rand": N_rand, "K": K, "steps": steps,
                      "held_out_view": i_test, "image": [H, W], "max_abs_psnr_diff_db": float(d.max()),
                      "control_max_abs_psnr_diff_db_oracle_vs_1ulp_perturbed_oracle": float(d2.max()),
                      "oracle_cpu_s_per_step": round(t_cpu / steps, 3), "oracle_threads": threads, **curve}), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if len(a) > 0 else 200, int(a[1]) if len(a) > 1 else 256)
