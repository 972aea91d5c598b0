"""Shared by the CPU (oracle) and GPU (HIP) tests of fixture G19 - the REFERENCE's own 120-step training curve on a tiny
procedural scene (tests/golden/make_golden.py: psnr_curve_fixture).  The per-step draws are re-derived from the fixture's
seed in the generator's order; `sel_checksum` guards that order."""
import numpy as np
import torch


def draws(step_rng, n_pool, n, K):
    sel = step_rng.integers(0, n_pool, n)
    t_rand = torch.tensor(step_rng.uniform(0, 1, (n, 128)), dtype=torch.float32)
    ea = torch.tensor(step_rng.standard_normal((K, 1)), dtype=torch.float32)
    er = torch.tensor(step_rng.standard_normal((K, 3)), dtype=torch.float32)
    return sel, t_rand, ea, er


def views(g):
    i_test = [int(i) for i in g["i_test"]]
    i_train = [i for i in range(int(g["n_views"])) if i not in i_test]
    return i_train, i_test


def lr_of_step(lrate, lrate_decay, s):
    """rate the reference's optimiser holds DURING 0-based step s: the initial rate at s = 0, afterwards what RUN:1073-1077 wrote
    back after step s - 1 with global_step = s - 1"""
    return lrate * (0.1 ** (max(s - 1, 0) / (lrate_decay * 1000)))


def psnr_of(pred_mean, image):
    return float(-10.0 * torch.log10(torch.mean((pred_mean - image) ** 2)))


def hip_curve(g, precision=None, dev="cuda"):
    """The HIP path over the fixture: same weights, batches, jitter and latents as the reference's run.  Returns
    (|loss - ref| / |ref| per step, |train PSNR - ref| per step in dB, {checkpoint step: held-out PSNR - ref [3]})."""
    import cfnerf_amd
    from cfnerf_amd import train as TR
    from oracle import cfnerf_oracle as O
    from util_hip import build_model
    H, W, focal, near, far = int(g["H"]), int(g["W"]), float(g["focal"]), float(g["near"]), float(g["far"])
    K, n, every = int(g["K"]), int(g["n_rand"]), int(g["every"])
    cfg = O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=K)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, int(g["seed"]), no_ndc=True)
    net = model.module
    if precision:
        net.set_precision(precision)
    poses, images = torch.tensor(g["poses"]), torch.tensor(g["images"])
    i_train, i_test = views(g)
    ro_all, rd_all, tg_all = [], [], []
    for v in i_train:                                   # the product's own ray kernel (get_rays formula HLP:288-297)
        ro, rd = cfnerf_amd.get_rays(H, W, focal, poses[v].to(dev))
        ro_all.append(ro.reshape(-1, 3)); rd_all.append(rd.reshape(-1, 3)); tg_all.append(images[v].reshape(-1, 3).to(dev))
    ro_all, rd_all, tg_all = torch.cat(ro_all), torch.cat(rd_all), torch.cat(tg_all)
    net.sample_alpha, net.sample_rgb = torch.tensor(g["eps_alpha_eval"]).clone(), torch.tensor(g["eps_rgb_eval"]).clone()
    kw = dict(kw_test); kw.update(near=near, far=far, ndc=False)

    def held_out():
        out = []
        for v in i_test:
            with torch.no_grad():
                rgb, _, _, _ = cfnerf_amd.render(H, W, focal, c2w=poses[v].to(dev), **kw)
            out.append(psnr_of(rgb.mean(-1).cpu(), images[v]))
        return np.array(out)

    tr = TR.Trainer(net, lrate=float(g["lrate"]), lrate_decay=int(g["lrate_decay"]), beta1=float(g["beta1"]))
    rng = np.random.default_rng(int(g["step_seed"]))
    held = {0: held_out() - g["psnr_test"][0]}
    dl, dp, sel_sum = [], [], 0
    for s in range(int(g["n_steps"])):
        sel, t_rand, ea, er = draws(rng, ro_all.shape[0], n, K)
        sel_sum += int(sel.sum())
        sel_t = torch.tensor(sel, device=dev)
        sc = tr.step(H, W, focal, (ro_all[sel_t], rd_all[sel_t]), tg_all[sel_t].contiguous(), t_rand=t_rand.to(dev),
                     eps=torch.cat([er, ea], -1).to(dev), near=near, far=far, ndc=False).cpu()
        dl.append(abs(float(sc[0]) - float(g["loss"][s])) / abs(float(g["loss"][s])))
        dp.append(abs(float(sc[3]) - float(g["psnr_train"][s])))
        if (s + 1) % every == 0:
            held[s + 1] = held_out() - g["psnr_test"][(s + 1) // every]
    assert sel_sum == int(g["sel_checksum"]), "the per-step draws left the generator's order"
    return np.array(dl), np.array(dp), held


# bounds of the -m gpu test: (first step, last step, relative loss error, train-batch PSNR error in dB); measured (MI355X, fp32 / bf16x3):
# 1.1e-7 / 2.9e-6 dB over steps 0-10, 4e-6 / 6e-5 (8e-6 / 9e-5) over 10-40, 3e-4 / 1.4e-3 (2.8e-4 / 1e-3) over 40-120 - the trajectories
# separate as fp32 re-association noise is amplified by training, so the bounds widen with the step
CURVE_BOUNDS = ((0, 10, 1e-5, 1e-4), (10, 40, 2e-4, 2e-3), (40, 120, 5e-3, 2e-2))
HELD_OUT_BOUNDS = {0: 1e-4, 40: 2e-3, 80: 5e-3, 120: 1e-2}        # dB; measured 1e-6, 7e-5, 1.2e-4, 3.3e-4 (bf16x3: 9.6e-4 at step 120)
