"""-m gpu: the OPT-IN split-bf16 ("bf16x3") arithmetic mode of the fused forward is held to the SAME parity tolerances
as the default exact-fp32 MFMA mode (golden vectors from the reference + CPU oracle)."""
import numpy as np
import pytest
import torch

import cfnerf_amd
from oracle import cfnerf_oracle as O
from util_hip import ATOL_DISP, build_model, close, fern_rays

pytestmark = pytest.mark.gpu
T = lambda a: torch.tensor(np.asarray(a))
DEV = "cuda"


@pytest.mark.parametrize("tag", ["w64", "w256", "w128k5"])
def test_model_forward_bf16x3_vs_reference_golden(golden, tag):
    g = golden(f"g123_model_{tag}")
    cfg = O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=int(g["K"]), h_alpha_size=int(g["h_alpha_size"]), h_rgb_size=int(g["h_rgb_size"]))
    _, kw, _, model, p, _ = build_model(cfg, int(g["seed"]))
    net = model.module
    net.set_precision("bf16x3")
    x = T(g["x90"]).to(DEV)
    net.sample_alpha = T(g["eps_alpha"]).clone()
    net.sample_rgb = T(g["eps_rgb"]).clone()
    with torch.no_grad():
        raw_e, _ = net(x, False, True)
        raw_t, ent = net(x, False, False, eps_alpha=T(g["eps_alpha"]), eps_rgb=T(g["eps_rgb"]))
    close(raw_e, g["raw_eval"], what="raw_eval")
    close(raw_t, g["raw_train"], what="raw_train")
    close(ent.reshape(-1)[0], g["loss_entropy"], what="loss_entropy")


@pytest.mark.parametrize("tag", ["w64_ndc", "w64_nondc_lindisp_wb", "w256_ndc"])
def test_render_bf16x3_vs_reference_golden(golden, tag):
    g = golden(f"g57_render_{tag}")
    cfg = O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=int(g["K"]))
    over = dict(no_ndc=not bool(g["ndc"]), lindisp=bool(g["lindisp"]), white_bkgd=bool(g["white_bkgd"]))
    _, kw_train, kw_test, model, p, _ = build_model(cfg, int(g["seed"]), **over)
    model.module.set_precision("bf16x3")
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    with torch.no_grad():
        rgbs, disp, depth, extras = cfnerf_amd.render(H, W, focal, rays=T(g["rays"]).to(DEV), near=float(g["near"]), far=float(g["far"]),
                                                      t_rand=T(g["t_rand"]), eps_alpha=T(g["eps_alpha"]), eps_rgb=T(g["eps_rgb"]), **kw_train)
    close(extras["raw"], g["raw"], what="raw")
    close(rgbs, g["rgb_map"], what="rgb_map")
    close(depth, g["depth_map"], what="depth_map")
    close(disp, g["disp_map"], atol=ATOL_DISP, rtol=1e-3, what="disp_map")
    close(extras["loss_entropy"].mean(), g["loss_entropy"], what="loss_entropy")


@pytest.mark.parametrize("W,K,N", [(256, 4, 96), (512, 8, 24)])
def test_render_bf16x3_vs_oracle_and_vs_fp32_mode(W, K, N):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    _, kw_train, _, model, p, _ = build_model(cfg, 300 + W + K)
    rng = np.random.default_rng(W + K + N)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
    out = {}
    for mode in ("fp32", "bf16x3"):
        model.module.set_precision(mode)
        with torch.no_grad():
            out[mode] = cfnerf_amd.render(H, Wd, focal, rays=rays.to(DEV), t_rand=t_rand, eps_alpha=ea, eps_rgb=er, **kw_train)
    r = O.render(p, H, Wd, focal, cfg, ea, er, True, rays=(rays[0], rays[1]), t_rand=t_rand)
    close(out["bf16x3"][3]["raw"], r["raw"], what="raw vs oracle")
    close(out["bf16x3"][0], r["rgb_map"], what="rgb_map vs oracle")
    close(out["bf16x3"][2], r["depth_map"], what="depth vs oracle")
    assert not torch.equal(out["bf16x3"][0], out["fp32"][0])          # it really is a different arithmetic path
    close(out["bf16x3"][0], out["fp32"][0], what="rgb_map vs fp32 mode")


def test_train_step_with_bf16x3_forward_matches_reference_gradients(golden):
    """Forward and backward-data in bf16x3, weight-gradient GEMMs in fp32: the gradients still meet the reference within
    the train tolerances."""
    from cfnerf_amd import train as TR
    from test_hip_train import grad_close, mask_corrected
    g = golden("g57_render_w64_ndc")
    cfg = O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=int(g["K"]))
    _, kw_train, _, model, p, _ = build_model(cfg, int(g["seed"]))
    net = model.module
    net.set_precision("bf16x3")
    tr = TR.Trainer(net, beta1=float(g["beta1"]))
    eps = torch.cat([T(g["eps_rgb"]), T(g["eps_alpha"])], -1).to(DEV)
    grad = tr.forward_backward(int(g["H"]), int(g["W"]), float(g["focal"]), T(g["rays"]).to(DEV), T(g["target"]).to(DEV),
                               t_rand=T(g["t_rand"]).to(DEV), eps=eps)
    close(tr.scalars[0].cpu(), g["loss"], atol=1e-5, rtol=1e-4, what="loss")
    rays = T(g["rays"])
    packed = O.pack_rays(int(g["H"]), int(g["W"]), float(g["focal"]), rays[0], rays[1], True, 0., 1.)
    corr, _ = mask_corrected(net, p, packed, T(g["target"]), cfg, T(g["eps_alpha"]), T(g["eps_rgb"]), T(g["t_rand"]), float(g["beta1"]))
    for key, (off, cnt) in net.layout.items():
        if ("grad." + key) in g:
            grad_close(grad[off:off + cnt].cpu().numpy().reshape(g["grad." + key].shape), g["grad." + key] + corr[key], "grad " + key)


def test_bf16x3_accuracy_class_vs_fp64():
    """Error of each arithmetic mode against the fp64 oracle on the network outputs: the split-bf16 mode must stay in the
    same accuracy class as exact-fp32 MFMA (within 4x of its error) and far inside the parity tolerance."""
    cfg = O.OracleCfg(netwidth=256, K_samples=4)
    _, kw_train, _, model, p, _ = build_model(cfg, 3)
    rng = np.random.default_rng(0)
    N = 64
    rays, (H, Wd, focal) = fern_rays(rng, N)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((4, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((4, 3)), dtype=torch.float32)
    p64 = {k: v.double() for k, v in p.items()}
    packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.)
    with torch.no_grad():
        ref = O.render_rays(p64, packed.double(), cfg, ea.double(), er.double(), True, t_rand=t_rand.double())
    err = {}
    for mode in ("fp32", "bf16x3"):
        model.module.set_precision(mode)
        with torch.no_grad():
            out = cfnerf_amd.render(H, Wd, focal, rays=rays.to(DEV), t_rand=t_rand, eps_alpha=ea, eps_rgb=er, **kw_train)
        err[mode] = {"raw": float((out[3]["raw"].cpu().double() - ref["raw"]).abs().max()),
                     "rgb": float((out[0].cpu().double() - ref["rgb_map"]).abs().max())}
    # the fp32 noise floor of this path is set by the 2^9-frequency encoding of fp32 points (~1e-5 on raw)
    assert err["bf16x3"]["raw"] <= 4 * err["fp32"]["raw"] + 1e-6, err
    assert err["bf16x3"]["rgb"] <= 4 * err["fp32"]["rgb"] + 1e-7, err
    assert err["bf16x3"]["rgb"] < 1e-5, err
