"""-m gpu: the HIP path (through the C ABI and the reference-shaped Python mirror) against
(1) golden vectors captured from the real reference and (2) the CPU oracle on seeded inputs."""
import numpy as np
import pytest
import torch

import cfnerf_amd
from oracle import cfnerf_oracle as O
from util_hip import ATOL_DISP, build_model, close, fern_rays, make_args

pytestmark = pytest.mark.gpu
T = lambda a: torch.tensor(np.asarray(a))
DEV = "cuda"


def cfg_from(g):
    return O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=int(g["K"]), netdepth=int(g.get("netdepth", 8)), n_flows=int(g.get("n_flows", 4)),
                       h_alpha_size=int(g.get("h_alpha_size", 32)), h_rgb_size=int(g.get("h_rgb_size", 64)),
                       multires=int(g.get("multires", 10)), multires_views=int(g.get("multires_views", 4)))


def test_native_library_is_loaded():
    from cfnerf_amd import _lib
    assert _lib.lib().cfnerf_version() >= 100
    with open("/proc/self/maps") as f:
        assert "libcfnerf_hip.so" in f.read()


# ---------------------------------------------------------------- NeRF_Flows.forward (MOD:188-291)
@pytest.mark.parametrize("tag", ["w64", "w256", "w64k1", "w128k5"])
def test_model_forward_vs_reference_golden(golden, tag):
    g = golden(f"g123_model_{tag}")
    cfg = cfg_from(g)
    _, kw, _, model, p, _ = build_model(cfg, int(g["seed"]))
    net = model.module
    x = T(g["x90"]).to(DEV)
    net.sample_alpha = T(g["eps_alpha"]).clone()          # R9: eval latents are plain attributes
    net.sample_rgb = T(g["eps_rgb"]).clone()
    with torch.no_grad():
        raw_e, aux = net(x, False, True)
        raw_t, ent = net(x, False, False, eps_alpha=T(g["eps_alpha"]), eps_rgb=T(g["eps_rgb"]))
    close(raw_e, g["raw_eval"], what="raw_eval")
    assert float(aux.abs().max()) == 0.0 and aux.shape == raw_e.shape
    close(raw_t, g["raw_train"], what="raw_train")
    assert list(ent.shape) == list(g["loss_entropy_shape"])
    close(ent.reshape(-1)[0], g["loss_entropy"], what="loss_entropy")


@pytest.mark.parametrize("W,K,P", [(256, 4, 1000), (256, 32, 129), (512, 8, 200), (128, 3, 64), (64, 1, 1),
                                   # every netwidth that is a multiple of 64 (RUN:584 takes any), K up to 128 (RUN:631 defaults to 64)
                                   (192, 4, 150), (320, 5, 130), (384, 3, 70), (448, 2, 65), (256, 128, 70), (64, 100, 33)])
def test_model_forward_vs_oracle(W, K, P):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    _, kw, _, model, p, _ = build_model(cfg, 100 + W + K)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(P, 90, generator=g) * 2 - 1
    ea, er = torch.randn(K, 1, generator=g), torch.randn(K, 3, generator=g)
    with torch.no_grad():
        raw_t, ent = model.module(x.to(DEV), False, False, eps_alpha=ea, eps_rgb=er)
    raw_o, ent_o = O.nerf_flows_forward(p, x, ea, er, cfg, is_test=False)
    close(raw_t, raw_o, what="raw")
    close(ent.reshape(-1)[0], ent_o, what="entropy")


@pytest.mark.parametrize("W,ha,hr,K,N", [(256, 96, 128, 4, 20), (64, 128, 96, 3, 12), (512, 96, 128, 2, 6), (192, 96, 96, 5, 9), (256, 64, 96, 4, 10),
                                         (128, 128, 32, 2, 8)])
def test_head_sizes_96_and_128_vs_oracle(W, ha, hr, K, N):
    """--h_alpha_size / --h_rgb_size (RUN:617-620 take any value): 32, 64, 96, 128 - model forward and a full render vs the oracle"""
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=ha, h_rgb_size=hr)
    _, kw_train, _, model, p, _ = build_model(cfg, 300 + ha + hr)
    g = torch.Generator().manual_seed(ha)
    x = torch.rand(130, 90, generator=g) * 2 - 1
    ea, er = torch.randn(K, 1, generator=g), torch.randn(K, 3, generator=g)
    with torch.no_grad():
        raw_t, ent = model.module(x.to(DEV), False, False, eps_alpha=ea, eps_rgb=er)
    raw_o, ent_o = O.nerf_flows_forward(p, x, ea, er, cfg, is_test=False)
    close(raw_t, raw_o, what="raw")
    close(ent.reshape(-1)[0], ent_o, what="entropy")
    rng = np.random.default_rng(N)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    with torch.no_grad():
        rgbs, disp, depth, extras = cfnerf_amd.render(H, Wd, focal, rays=rays.to(DEV), t_rand=t_rand, eps_alpha=ea, eps_rgb=er, **kw_train)
    r = O.render(p, H, Wd, focal, cfg, ea, er, True, rays=(rays[0], rays[1]), t_rand=t_rand)
    close(rgbs, r["rgb_map"], what="rgb_map")
    close(depth, r["depth_map"], what="depth_map")


@pytest.mark.parametrize("F,W,K,N", [(1, 64, 3, 10), (2, 256, 4, 14), (3, 128, 5, 9), (3, 512, 16, 5)])
def test_n_flows_other_than_four_vs_oracle(F, W, K, N):
    """--n_flows (RUN:622) 1 .. 3: the kernels are built for four flow steps; a shorter stack runs as four with the missing steps'
    parameters zero in every packed operand (identity steps, log-det 0) - model forward, entropy and a full render vs the oracle"""
    cfg = O.OracleCfg(netwidth=W, K_samples=K, n_flows=F, h_alpha_size=64 if W == 512 else 32)
    _, kw_train, _, model, p, _ = build_model(cfg, 400 + F)
    assert model.module.n_flows == F and tuple(model.module.view("flows_rgb.amor_d.weight").shape) == (9 * F, cfg.h_rgb_size)
    g = torch.Generator().manual_seed(F)
    x = torch.rand(150, 90, generator=g) * 2 - 1
    ea, er = torch.randn(K, 1, generator=g), torch.randn(K, 3, generator=g)
    with torch.no_grad():
        raw_t, ent = model.module(x.to(DEV), False, False, eps_alpha=ea, eps_rgb=er)
        raw_e, _ = model.module(x.to(DEV), False, True, eps_alpha=ea, eps_rgb=er)
    raw_o, ent_o = O.nerf_flows_forward(p, x, ea, er, cfg, is_test=False)
    close(raw_t, raw_o, what="raw")
    close(raw_e, raw_o, what="raw (eval branch)")
    close(ent.reshape(-1)[0], ent_o, what="entropy")
    rng = np.random.default_rng(N)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    with torch.no_grad():
        rgbs, disp, depth, extras = cfnerf_amd.render(H, Wd, focal, rays=rays.to(DEV), t_rand=t_rand, eps_alpha=ea, eps_rgb=er, **kw_train)
    r = O.render(p, H, Wd, focal, cfg, ea, er, True, rays=(rays[0], rays[1]), t_rand=t_rand)
    close(rgbs, r["rgb_map"], what="rgb_map")
    close(depth, r["depth_map"], what="depth_map")
    close(extras["loss_entropy"].mean(), r["loss_entropy"], what="loss_entropy")
    with pytest.raises(RuntimeError, match="n_flows"):
        cfnerf_amd.create_nerf(make_args(O.OracleCfg(netwidth=64, n_flows=5)))


# ---------------------------------------------------------------- raw2outputs (RUN:411-454)
@pytest.mark.parametrize("wb", [False, True])
def test_composite_vs_reference_golden(golden, wb):
    g = golden("g4_composite")
    s = "wb" if wb else "nb"
    rgb, disp, w, depth = cfnerf_amd.raw2outputs(T(g["raw"]).to(DEV), T(g["z_vals"]).to(DEV), T(g["rays_d"]).to(DEV), 0, wb)
    close(rgb, g[f"rgb_map_{s}"], what="rgb_map")
    close(w, g[f"weights_{s}"], what="weights")
    close(depth, g[f"depth_{s}"], what="depth")
    close(disp, g[f"disp_{s}"], atol=ATOL_DISP, rtol=1e-3, what="disp")


@pytest.mark.parametrize("N,S,K", [(5, 128, 4), (3, 70, 3), (2, 2, 1), (7, 200, 32), (1, 64, 5), (6, 129, 8), (3, 65, 13), (2, 300, 17), (5, 128, 64),
                                   (2, 63, 128), (9, 2, 4), (1, 1000, 2)])
def test_composite_vs_oracle_ragged(N, S, K):
    g = torch.Generator().manual_seed(N * 1000 + S + K)
    raw = torch.randn(N, S, K, 4, generator=g) * 3
    z = torch.sort(torch.rand(N, S, generator=g) * 4 + 0.5, -1).values
    d = torch.randn(N, 3, generator=g)
    for wb in (False, True):
        rgb, disp, w, depth = cfnerf_amd.raw2outputs(raw.to(DEV), z.to(DEV), d.to(DEV), 0, wb)
        rgb_o, disp_o, w_o, depth_o = O.raw2outputs(raw, z, d, wb)
        close(rgb, rgb_o, what="rgb")
        close(w, w_o, what="weights")
        close(depth, depth_o, what="depth")
        close(disp, disp_o, atol=ATOL_DISP, rtol=1e-3, what="disp")


# ---------------------------------------------------------------- render (RUN:103-170) on a ray batch
@pytest.mark.parametrize("tag", ["w64_ndc", "w64_nondc_lindisp_wb", "w256_ndc"])
def test_render_batch_vs_reference_golden(golden, tag):
    g = golden(f"g57_render_{tag}")
    cfg = cfg_from(g)
    over = dict(no_ndc=not bool(g["ndc"]), lindisp=bool(g["lindisp"]), white_bkgd=bool(g["white_bkgd"]))
    _, kw_train, kw_test, model, p, _ = build_model(cfg, int(g["seed"]), **over)
    net = model.module
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    rays = T(g["rays"]).to(DEV)
    near, far = float(g["near"]), float(g["far"])
    with torch.no_grad():
        rgbs, disp, depth, extras = cfnerf_amd.render(H, W, focal, chunk=8192, rays=rays, near=near, far=far,
                                                      verbose=False, retraw=False, t_rand=T(g["t_rand"]),
                                                      eps_alpha=T(g["eps_alpha"]), eps_rgb=T(g["eps_rgb"]), **kw_train)
    assert sorted(extras.keys()) == list(g["train_extras_keys"])
    close(extras["pts"], g["pts"], atol=1e-6, rtol=1e-6, what="pts")
    close(extras["raw"], g["raw"], what="raw")
    close(rgbs, g["rgb_map"], what="rgb_map")
    close(depth, g["depth_map"], what="depth_map")
    close(disp, g["disp_map"], atol=ATOL_DISP, rtol=1e-3, what="disp_map")
    assert extras["loss_entropy"].numel() == int(g["loss_entropy_numel"])
    close(extras["loss_entropy"].mean(), g["loss_entropy"], what="loss_entropy")
    # eval branch: fixed latents, last = 0, perturb off (RUN:403-407)
    net.sample_alpha = T(g["eps_alpha"]).clone()
    net.sample_rgb = T(g["eps_rgb"]).clone()
    with torch.no_grad():
        rgbs_e, disp_e, depth_e, extras_e = cfnerf_amd.render(H, W, focal, chunk=8192, rays=rays, near=near, far=far, **kw_test)
    assert sorted(extras_e.keys()) == list(g["eval_extras_keys"])
    close(rgbs_e, g["rgb_map_eval"], what="rgb_map_eval")
    close(depth_e, g["depth_map_eval"], what="depth_map_eval")
    close(disp_e, g["disp_map_eval"], atol=ATOL_DISP, rtol=1e-3, what="disp_map_eval")


@pytest.mark.parametrize("tag", ["w64_nondc_lindisp_wb", "w256_ndc"])
def test_render_batch_with_hardware_flow_math_vs_reference_golden(golden, tag):
    """cfnerf_model_set_flow_math: the K flows + composite on the hardware transcendentals (what the fused kernels take by themselves
    from 16 latents on), FORCED here on the K = 4 fixtures of the real reference: same outputs within the same bounds, not the same bits."""
    g = golden(f"g57_render_{tag}")
    cfg = cfg_from(g)
    over = dict(no_ndc=not bool(g["ndc"]), lindisp=bool(g["lindisp"]), white_bkgd=bool(g["white_bkgd"]))
    _, kw_train, kw_test, model, p, _ = build_model(cfg, int(g["seed"]), **over)
    net = model.module
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    kw = dict(rays=T(g["rays"]).to(DEV), near=float(g["near"]), far=float(g["far"]), t_rand=T(g["t_rand"]), eps_alpha=T(g["eps_alpha"]),
              eps_rgb=T(g["eps_rgb"]))
    out = {}
    for mode in ("libm", "fast", "auto"):
        net.set_flow_math(mode)
        with torch.no_grad():
            out[mode] = cfnerf_amd.render(H, W, focal, **kw, **kw_train)
    rgbs, disp, depth, extras = out["fast"]
    close(extras["raw"], g["raw"], what="raw")
    close(rgbs, g["rgb_map"], what="rgb_map")
    close(depth, g["depth_map"], what="depth_map")
    close(disp, g["disp_map"], atol=ATOL_DISP, rtol=1e-3, what="disp_map")
    close(extras["loss_entropy"].mean(), g["loss_entropy"], what="loss_entropy")
    assert not torch.equal(out["fast"][3]["raw"], out["libm"][3]["raw"])                  # a different arithmetic path ...
    assert torch.equal(out["auto"][3]["raw"], out["libm"][3]["raw"]) and torch.equal(out["auto"][0], out["libm"][0])    # ... that K = 4 does not take by default
    with pytest.raises(KeyError):
        net.set_flow_math("fastest")


def test_flow_math_default_switches_at_16_latents():
    """auto = libm below 16 latent samples, hardware forms from 16 on; both against the oracle at K = 32 (the authors' recipe)"""
    cfg = O.OracleCfg(netwidth=128, K_samples=32)
    _, kw_train, _, model, p, _ = build_model(cfg, 66)
    net = model.module
    rng = np.random.default_rng(2)
    rays, (H, Wd, focal) = fern_rays(rng, 12)
    t_rand = torch.tensor(rng.uniform(0, 1, (12, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((32, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((32, 3)), dtype=torch.float32)
    r = O.render(p, H, Wd, focal, cfg, ea, er, True, rays=(rays[0], rays[1]), t_rand=t_rand)
    out = {}
    for mode in ("auto", "libm", "fast"):
        net.set_flow_math(mode)
        with torch.no_grad():
            out[mode] = cfnerf_amd.render(H, Wd, focal, rays=rays.to(DEV), t_rand=t_rand, eps_alpha=ea, eps_rgb=er, **kw_train)
        close(out[mode][3]["raw"], r["raw"], what=f"raw [{mode}]")
        close(out[mode][0], r["rgb_map"], what=f"rgb_map [{mode}]")
        close(out[mode][2], r["depth_map"], what=f"depth_map [{mode}]")
        close(out[mode][3]["loss_entropy"].mean(), r["loss_entropy"], what=f"entropy [{mode}]")
    assert torch.equal(out["auto"][0], out["fast"][0]) and not torch.equal(out["auto"][0], out["libm"][0])


def test_render_config1_k1_vs_reference_golden(golden):
    """G14 = BASELINE config 1 through render() (RUN:103-170): K = 1 latent sample, N_rand = 256 fern-shaped rays, default
    width, forward only (the reference's K = 1 train loss is NaN) - train-mode and eval-mode render."""
    g = golden("g14_render_c1_k1")
    cfg = cfg_from(g)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, int(g["seed"]))
    net = model.module
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    rays = T(g["rays"]).to(DEV)
    with torch.no_grad():
        rgbs, disp, depth, extras = cfnerf_amd.render(H, W, focal, chunk=8192, rays=rays, near=0., far=1., t_rand=T(g["t_rand"]),
                                                      eps_alpha=T(g["eps_alpha"]), eps_rgb=T(g["eps_rgb"]), **kw_train)
    assert list(rgbs.shape) == [256, 3, 1] and list(disp.shape) == [256, 1] and list(extras["raw"].shape) == [256, 128, 1, 4]
    close(rgbs, g["rgb_map"], what="rgb_map")
    close(depth, g["depth_map"], what="depth_map")
    close(disp, g["disp_map"], atol=ATOL_DISP, rtol=1e-3, what="disp_map")
    close(extras["raw"][:4], g["raw_first4"], what="raw")
    assert list(extras["loss_entropy"].shape) == list(g["loss_entropy_shape"])
    close(extras["loss_entropy"].reshape(-1)[0], g["loss_entropy"], what="loss_entropy")
    net.sample_alpha, net.sample_rgb = T(g["eps_alpha"]).clone(), T(g["eps_rgb"]).clone()
    with torch.no_grad():
        rgbs_e, disp_e, depth_e, extras_e = cfnerf_amd.render(H, W, focal, chunk=8192, rays=rays, near=0., far=1., **kw_test)
    close(rgbs_e, g["rgb_map_eval"], what="rgb_map_eval")
    close(depth_e, g["depth_map_eval"], what="depth_map_eval")
    close(disp_e, g["disp_map_eval"], atol=ATOL_DISP, rtol=1e-3, what="disp_map_eval")
    # and the same batch in two chunks through batchify_rays (RUN:88-100): bit-identical maps
    with torch.no_grad():
        packed = torch.empty(256, 11, device=DEV)
        from cfnerf_amd import _lib as L
        ro, rd = rays[0].contiguous(), rays[1].contiguous()
        L.check(L.lib().cfnerf_rays_setup(H, W, float(focal), None, L.ptr(ro), L.ptr(rd), 256, 0, 1, 0., 1., L.ptr(packed), L.stream()), "rays_setup")
        kw = {k: v for k, v in kw_test.items() if k not in ("use_viewdirs", "ndc")}
        two = cfnerf_amd.batchify_rays(packed, chunk=100, **kw)
    assert torch.equal(two["rgb_map"], rgbs_e) and torch.equal(two["depth_map"], depth_e)


def test_render_c2w_vs_reference_golden(golden):
    g = golden("g6_render_c2w")
    cfg = cfg_from(g)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, int(g["seed"]))
    net = model.module
    net.sample_alpha = T(g["eps_alpha"]).clone()
    net.sample_rgb = T(g["eps_rgb"]).clone()
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    with torch.no_grad():
        rgbs, disp, depth, _ = cfnerf_amd.render(H, W, focal, chunk=8192, c2w=T(g["c2w"]), near=0., far=1., **kw_test)
    assert list(rgbs.shape) == [H, W, 3, 4] and list(disp.shape) == [H, W, 4]
    close(rgbs, g["rgb_map"], what="rgb_map")
    close(depth, g["depth_map"], what="depth_map")
    close(disp, g["disp_map"], atol=ATOL_DISP, rtol=1e-3, what="disp_map")
    with torch.no_grad():   # RUN:139-141
        rgbs, disp, depth, _ = cfnerf_amd.render(H, W, focal, chunk=8192, c2w=T(g["c2w"]), c2w_staticcam=T(g["c2w_static"]),
                                                 near=0., far=1., **kw_test)
    close(rgbs, g["rgb_map_static"], what="rgb_map (staticcam)")
    close(depth, g["depth_map_static"], what="depth_map (staticcam)")
    close(disp, g["disp_map_static"], atol=ATOL_DISP, rtol=1e-3, what="disp_map (staticcam)")


def test_get_rays_and_ndc_rays_helpers_vs_reference_golden(golden):
    """the standalone ray helpers (HLP:288-297, 360-377) are the product's kernels, not torch restatements: GPU tensors out,
    the reference's values (fixture G6), a near plane other than 1 against the oracle, CPU tensors rejected"""
    g = golden("g6_render_c2w")
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    for pose in (T(g["c2w"]), T(g["c2w"]).to(DEV)):                # a host pose goes straight into the call, a device pose is fetched
        ro, rd = cfnerf_amd.get_rays(H, W, focal, pose)
        assert ro.is_cuda and list(ro.shape) == [H, W, 3]
        close(ro, g["rays_o"], atol=0, rtol=0, what="rays_o")
        close(rd, g["rays_d"], atol=1e-6, rtol=1e-6, what="rays_d")
    no, nd = cfnerf_amd.ndc_rays(H, W, focal, 1., ro, rd)
    close(no, g["ndc_o"], atol=2e-6, rtol=1e-5, what="ndc_o")
    close(nd, g["ndc_d"], atol=2e-6, rtol=1e-5, what="ndc_d")
    no, nd = cfnerf_amd.ndc_rays(H, W, focal, 0.5, ro, rd)
    eo, ed = O.ndc_rays(H, W, focal, 0.5, ro.cpu(), rd.cpu())
    close(no, eo, atol=2e-6, rtol=1e-5, what="ndc_o near=0.5")
    close(nd, ed, atol=2e-6, rtol=1e-5, what="ndc_d near=0.5")
    with pytest.raises(RuntimeError):
        cfnerf_amd.ndc_rays(H, W, focal, 1., ro.cpu(), rd.cpu())


@pytest.mark.parametrize("W,K,N,ndc", [(256, 4, 96, True), (256, 8, 40, False), (512, 16, 24, True), (128, 2, 33, True),
                                       (192, 3, 20, True), (320, 4, 12, False), (384, 2, 10, True), (448, 3, 9, True), (128, 128, 5, True),
                                       (256, 96, 4, False),
                                       # from 16 latents on a wave takes its latents two at a time (flows_fwd2) and a left-over one alone: odd counts per
                                       # wave, waves with different counts (4 waves: K = 17, 18, 19, 23; 8 waves at W = 512: K = 20, 24)
                                       (128, 17, 6, True), (64, 18, 7, False), (128, 19, 5, True), (64, 23, 5, True), (512, 24, 4, True), (512, 20, 4, False)])
def test_render_vs_oracle(W, K, N, ndc):
    cfg = O.OracleCfg(netwidth=W, K_samples=K, h_alpha_size=64 if W == 512 else 32)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, 300 + W + K, no_ndc=not ndc)
    rng = np.random.default_rng(W + K + N)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    near, far = (0., 1.) if ndc else (1.2, 8.0)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
    with torch.no_grad():
        rgbs, disp, depth, extras = cfnerf_amd.render(H, Wd, focal, rays=rays.to(DEV), near=near, far=far, t_rand=t_rand,
                                                      eps_alpha=ea, eps_rgb=er, **kw_train)
    r = O.render(p, H, Wd, focal, cfg, ea, er, True, rays=(rays[0], rays[1]), ndc=ndc, near=near, far=far, t_rand=t_rand)
    close(extras["raw"], r["raw"], what="raw")
    close(rgbs, r["rgb_map"], what="rgb_map")
    close(depth, r["depth_map"], what="depth_map")
    close(disp, r["disp_map"], atol=ATOL_DISP, rtol=1e-3, what="disp_map")
    close(extras["loss_entropy"].mean(), r["loss_entropy"], what="entropy")


def test_ragged_sample_table():
    """S not a multiple of the 64-row tile, S < 64, and a custom t_vals table (the kernel takes it as input)."""
    cfg = O.OracleCfg(netwidth=64, K_samples=3)
    _, kw_train, _, model, p, _ = build_model(cfg, 41)
    rng = np.random.default_rng(5)
    for S in (70, 17, 129):
        N = 9
        rays, (H, Wd, focal) = fern_rays(rng, N)
        tv = torch.sort(torch.tensor(rng.uniform(0, 1, S), dtype=torch.float32)).values
        t_rand = torch.tensor(rng.uniform(0, 1, (N, S)), dtype=torch.float32)
        ea = torch.tensor(rng.standard_normal((3, 1)), dtype=torch.float32)
        er = torch.tensor(rng.standard_normal((3, 3)), dtype=torch.float32)
        packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.)
        kw = dict(kw_train)
        kw.pop("use_viewdirs"); kw["N_samples"] = S
        with torch.no_grad():
            ret = cfnerf_amd.render_rays(packed.to(DEV), t_vals=tv, t_rand=t_rand, eps_alpha=ea, eps_rgb=er, retweights=True, **kw)
        r = O.render_rays(p, packed, cfg, ea, er, True, t_rand=t_rand, t_vals=tv)
        close(ret["raw"], r["raw"], what=f"raw S={S}")
        close(ret["rgb_map"], r["rgb_map"], what=f"rgb S={S}")
        close(ret["weights"], r["weights"], what=f"weights S={S}")
        close(ret["depth_map"], r["depth_map"], what=f"depth S={S}")


def test_unfused_query_path_matches_fused():
    """A caller-supplied network_query_fn (run_network -> NeRF_Flows kernel -> composite kernel) vs the fused launch."""
    cfg = O.OracleCfg(netwidth=256, K_samples=4)
    args, kw_train, kw_test, model, p, _ = build_model(cfg, 77)
    rng = np.random.default_rng(3)
    rays, (H, Wd, focal) = fern_rays(rng, 50)
    t_rand = torch.tensor(rng.uniform(0, 1, (50, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((4, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((4, 3)), dtype=torch.float32)
    fused_q = kw_train["network_query_fn"]
    custom = lambda *a, **k: fused_q(*a, **k)          # same query, but not marked fusable
    with torch.no_grad():
        a = cfnerf_amd.render(H, Wd, focal, rays=rays.to(DEV), t_rand=t_rand, eps_alpha=ea, eps_rgb=er, **kw_train)
        kw2 = dict(kw_train, network_query_fn=custom)
        b = cfnerf_amd.render(H, Wd, focal, rays=rays.to(DEV), t_rand=t_rand, eps_alpha=ea, eps_rgb=er, **kw2)
    close(b[3]["raw"], a[3]["raw"], what="raw")
    close(b[0], a[0], what="rgb_map")
    close(b[2], a[2], what="depth")


# ---------------------------------------------------------------- full-size, size-independent properties
def test_full_size_properties_config2():
    """BASELINE config 2 sizes (N_rand 1024, S 128, K 4, W 256): properties that need no oracle run."""
    cfg = O.OracleCfg(netwidth=256, K_samples=4)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, 9)
    rng = np.random.default_rng(11)
    N = 1024
    rays, (H, Wd, focal) = fern_rays(rng, N)
    rays = rays.to(DEV)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, 128)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((4, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((4, 3)), dtype=torch.float32)
    kw = dict(t_rand=t_rand, eps_alpha=ea, eps_rgb=er)
    with torch.no_grad():
        full = cfnerf_amd.render(H, Wd, focal, rays=rays, **kw, **kw_train)
        # (1) rays are independent: two half batches reproduce the full batch bit-for-bit
        h0 = cfnerf_amd.render(H, Wd, focal, rays=rays[:, :400], t_rand=t_rand[:400], eps_alpha=ea, eps_rgb=er, **kw_train)
        h1 = cfnerf_amd.render(H, Wd, focal, rays=rays[:, 400:], t_rand=t_rand[400:], eps_alpha=ea, eps_rgb=er, **kw_train)
        assert torch.equal(torch.cat([h0[0], h1[0]]), full[0])
        assert torch.equal(torch.cat([h0[2], h1[2]]), full[2])
        # (2) latent samples are independent: permuting eps permutes the K axis bit-for-bit
        perm = torch.tensor([2, 0, 3, 1])
        pk = cfnerf_amd.render(H, Wd, focal, rays=rays, t_rand=t_rand, eps_alpha=ea[perm], eps_rgb=er[perm], **kw_train)
        assert torch.equal(pk[0], full[0][..., perm])
        assert torch.equal(pk[3]["raw"], full[3]["raw"][:, :, perm])
        # (3) composite invariants: rgb in [0,1], depth within [near, far] of the NDC volume
        rgb, depth = full[0], full[2]
        assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 + 1e-5
        assert float(depth.min()) >= 0.0 and float(depth.max()) <= 1.0 + 1e-5
    assert torch.isfinite(full[3]["loss_entropy"]).all()


def test_empty_batch_and_bad_arguments():
    cfg = O.OracleCfg(netwidth=64, K_samples=2)
    _, kw_train, _, model, _, _ = build_model(cfg, 1)
    kw = {k: v for k, v in kw_train.items() if k != "use_viewdirs"}
    ret = cfnerf_amd.render_rays(torch.zeros(0, 11, device=DEV), **kw)
    assert ret["rgb_map"].shape == (0, 3, 2)
    with pytest.raises(ValueError, match="N_samples"):
        cfnerf_amd.render_rays(torch.zeros(4, 11, device=DEV), **dict(kw, N_samples=64))
    with pytest.raises(NotImplementedError):
        cfnerf_amd.render_rays(torch.zeros(4, 11, device=DEV), **dict(kw, N_importance=64))
    with pytest.raises(ValueError):
        model.module(torch.zeros(3, 63, device=DEV), False, True)


# ---------------------------------------------------------------- EXTENSION: coarse + fine sampling (not in the reference)
@pytest.mark.parametrize("perturb", [0., 1.])
def test_hierarchical_extension_vs_own_restatement(perturb):
    """Parity here is UNPINNED by the reference (it has no second pass): the oracle restates
    nerf-pytorch's sample_pdf; this only shows the HIP resampler + explicit-depth launch agree with it."""
    cfg = O.OracleCfg(netwidth=128, K_samples=3)
    _, kw_train, kw_test, model, p, _ = build_model(cfg, 123)
    rng = np.random.default_rng(8)
    N, S, Ni = 21, 64, 128
    rays, (H, Wd, focal) = fern_rays(rng, N)
    packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.)
    tvc = torch.linspace(0., 1., steps=S)
    t_rand = torch.tensor(rng.uniform(0, 1, (N, S)), dtype=torch.float32) if perturb else None
    u = torch.tensor(rng.uniform(0, 1, (N, Ni)), dtype=torch.float32) if perturb else torch.linspace(0., 1., steps=Ni).expand(N, Ni).contiguous()
    ea = torch.tensor(rng.standard_normal((3, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((3, 3)), dtype=torch.float32)
    kw = {k: v for k, v in kw_test.items() if k not in ("use_viewdirs", "N_samples", "N_importance", "perturb")}
    with torch.no_grad():
        ret = cfnerf_amd.render_rays(packed.to(DEV), N_samples=S, N_importance=Ni, perturb=perturb, hierarchical_extension=True,
                                     t_rand=t_rand, u_fine=u, eps_alpha=ea, eps_rgb=er, **kw)
    r = O.render_rays_hierarchical(p, packed, cfg, ea, er, False, tvc, u, t_rand=t_rand)
    assert ret["z_vals"].shape == (N, S + Ni)
    assert bool((ret["z_vals"][:, 1:] >= ret["z_vals"][:, :-1]).all())
    close(ret["rgb0"], r["rgb0"], what="coarse rgb")
    close(ret["z_vals"], r["z_vals"], atol=2e-5, rtol=1e-4, what="merged depths")
    close(ret["rgb_map"], r["rgb_map"], atol=1e-4, rtol=1e-3, what="fine rgb")          # looser: depths move by ~1e-5
    close(ret["depth_map"], r["depth_map"], atol=1e-4, rtol=1e-3, what="fine depth")
    with pytest.raises(NotImplementedError):                                            # default stays reference-faithful
        cfnerf_amd.render_rays(packed.to(DEV), N_samples=S, N_importance=Ni, perturb=0., **kw)


# ---------------------------------------------------------------- shapes off the beaten path
@pytest.mark.parametrize("D,W,K,N,S", [(6, 128, 2, 5, 128), (4, 64, 64, 3, 128), (8, 256, 7, 1, 128), (8, 64, 3, 4, 300), (3, 64, 2, 6, 33)])
def test_generic_depth_k_limit_single_ray_long_tables(D, W, K, N, S):
    """netdepth != 8 (skip at D/2, RUN:327), K at the supported maximum, one ray, tables longer / shorter than a tile."""
    cfg = O.OracleCfg(netdepth=D, netwidth=W, K_samples=K)
    _, kw_train, _, model, p, _ = build_model(cfg, 700 + D + K, netdepth=D)
    rng = np.random.default_rng(D * 100 + K)
    rays, (H, Wd, focal) = fern_rays(rng, N)
    packed = O.pack_rays(H, Wd, focal, rays[0], rays[1], True, 0., 1.)
    tv = O.t_vals_table() if S == 128 else torch.sort(torch.tensor(rng.uniform(0, 1, S), dtype=torch.float32)).values
    t_rand = torch.tensor(rng.uniform(0, 1, (N, S)), dtype=torch.float32)
    ea = torch.tensor(rng.standard_normal((K, 1)), dtype=torch.float32)
    er = torch.tensor(rng.standard_normal((K, 3)), dtype=torch.float32)
    kw = {k: v for k, v in kw_train.items() if k != "use_viewdirs"}
    kw["N_samples"] = S
    with torch.no_grad():
        ret = cfnerf_amd.render_rays(packed.to(DEV), t_vals=tv, t_rand=t_rand, eps_alpha=ea, eps_rgb=er, retweights=True, **kw)
    r = O.render_rays(p, packed, cfg, ea, er, True, t_rand=t_rand, t_vals=tv)
    close(ret["raw"], r["raw"], what="raw")
    close(ret["rgb_map"], r["rgb_map"], what="rgb_map")
    close(ret["weights"], r["weights"], what="weights")
    close(ret["loss_entropy"].mean(), r["loss_entropy"], what="entropy")


def test_k_above_the_limit_and_bad_widths_are_rejected():
    import argparse
    from util_hip import make_args
    cfg = O.OracleCfg(netwidth=64, K_samples=129)
    args = make_args(cfg)
    kw, _, _, _, _ = cfnerf_amd.create_nerf(args)
    with pytest.raises(RuntimeError, match="K_samples"):
        cfnerf_amd.render_rays(torch.zeros(2, 11, device=DEV) + 0.5, **{k: v for k, v in kw.items() if k != "use_viewdirs"})
    with pytest.raises(RuntimeError, match="netwidth"):
        cfnerf_amd.create_nerf(make_args(O.OracleCfg(netwidth=96)))
    with pytest.raises(RuntimeError, match="netwidth"):
        cfnerf_amd.create_nerf(make_args(O.OracleCfg(netwidth=576)))
    with pytest.raises(RuntimeError, match="h_alpha_size"):
        cfnerf_amd.create_nerf(make_args(O.OracleCfg(netwidth=64, h_alpha_size=160)))
    with pytest.raises(RuntimeError, match="h_rgb_size"):
        cfnerf_amd.create_nerf(make_args(O.OracleCfg(netwidth=128, h_rgb_size=96)))      # 64 + 96 > max(128, 128)


# ---------------------------------------------------------------- standalone boundary kernels
def test_embedder_kernel_vs_reference_golden(golden):
    g = golden("g8_encoder")
    f10, d10 = cfnerf_amd.get_embedder(10, 0)
    f4, d4 = cfnerf_amd.get_embedder(4, 0)
    assert (d10, d4) == (int(g["d10"]), int(g["d4"]))
    x = T(g["x"]).to(DEV)
    close(f10(x), g["e10"], atol=2e-6, rtol=1e-5, what="embed multires 10")     # sin(512 x): one ulp of the argument = 3e-5 rad
    close(f4(x), g["e4"], atol=1e-6, rtol=1e-6, what="embed multires 4")
    assert f10(x.reshape(4, 4, 3)).shape == (4, 4, 63)


def test_sample_points_kernel_vs_reference_golden(golden):
    g = golden("g57_render_w64_ndc")
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    rays = T(g["rays"])
    packed = O.pack_rays(H, W, focal, rays[0], rays[1], True, 0., 1.).to(DEV)
    from cfnerf_amd import _lib as L
    N, S = packed.shape[0], 128
    z = torch.empty(N, S, device=DEV)
    pts = torch.empty(N, S, 3, device=DEV)
    tv = cfnerf_amd.t_vals_table(DEV)
    tr = T(g["t_rand"]).to(DEV)
    L.check(L.lib().cfnerf_sample_points(L.ptr(packed), L.ptr(tv), L.ptr(tr), 0, N, S, L.ptr(z), L.ptr(pts), L.stream()), "sample_points")
    close(pts, g["pts"], atol=1e-6, rtol=1e-6, what="pts")


@pytest.mark.parametrize("N,S,lindisp,jitter", [(1, 128, False, True), (7, 64, False, True), (5, 70, True, True), (9, 2, False, False),
                                                 (3, 200, True, False), (130, 65, False, True), (2, 1000, False, True)])
def test_sample_points_kernel_vs_oracle_ragged(N, S, lindisp, jitter):
    """RUN:510-534 standalone (one wave per ray, pts rows through LDS): every chunking of S, ray counts that do not fill a workgroup,
    lindisp, with and without the stratified jitter; z bit for bit the oracle's fp32 arithmetic up to the last place of the lerp."""
    from cfnerf_amd import _lib as L
    rng = np.random.default_rng(N * 31 + S)
    ro = torch.tensor(rng.standard_normal((N, 3)), dtype=torch.float32)
    rd = torch.tensor(rng.standard_normal((N, 3)), dtype=torch.float32)
    near = torch.tensor(rng.uniform(0.5, 1.5, (N, 1)), dtype=torch.float32)
    far = near + torch.tensor(rng.uniform(1.0, 5.0, (N, 1)), dtype=torch.float32)
    packed = torch.cat([ro, rd, near, far, rd / rd.norm(dim=-1, keepdim=True)], -1).contiguous()
    tv = torch.linspace(0., 1., steps=S)
    tr = torch.tensor(rng.uniform(0, 1, (N, S)), dtype=torch.float32) if jitter else None
    z_o = O.sample_z(near, far, tv, lindisp, tr)
    pts_o = ro[:, None, :] + rd[:, None, :] * z_o[..., None]
    z = torch.full((N, S), float("nan"), device=DEV)
    pts = torch.full((N, S, 3), float("nan"), device=DEV)
    packed_d, tv_d, tr_d = packed.to(DEV), tv.to(DEV), (tr.to(DEV) if jitter else None)      # (kept alive: the calls take raw pointers)
    L.check(L.lib().cfnerf_sample_points(L.ptr(packed_d), L.ptr(tv_d), L.ptr(tr_d) if jitter else None, L.F_LINDISP if lindisp else 0,
                                         N, S, L.ptr(z), L.ptr(pts), L.stream()), "sample_points")
    close(z, z_o, atol=1e-6, rtol=1e-6, what="z_vals")
    close(pts, pts_o, atol=2e-6, rtol=2e-6, what="pts")
    z2 = torch.full((N, S), float("nan"), device=DEV)          # z only (pts == NULL)
    L.check(L.lib().cfnerf_sample_points(L.ptr(packed_d), L.ptr(tv_d), L.ptr(tr_d) if jitter else None, L.F_LINDISP if lindisp else 0,
                                         N, S, L.ptr(z2), None, L.stream()), "sample_points")
    assert torch.equal(z2, z)


def test_seeded_create_nerf_matches_reference_construction(golden):
    """torch.manual_seed(1234); create_nerf(args) -> the reference's initial state_dict and eval latents (G11)."""
    g = golden("g11_seeded_init")
    cfg = O.OracleCfg(netwidth=256, K_samples=4)
    torch.manual_seed(1234)
    kw_train, _, _, _, _ = cfnerf_amd.create_nerf(make_args(cfg))
    net = kw_train["network_fn"].module
    assert np.array_equal(net.sample_alpha.numpy(), g["w256.sample_alpha"]) and np.array_equal(net.sample_rgb.numpy(), g["w256.sample_rgb"])
    sd = net.state_dict()
    for key in list(g):
        if key.startswith("w256.head."):
            k = key[len("w256.head."):]
            f = sd[k].reshape(-1).cpu()
            assert np.array_equal(f[:8].numpy(), g[key]), k
            assert float(f.double().sum()) == float(g["w256.sum." + k]), k


def test_seeded_implicit_draws_match_reference(golden):
    """G12: with no explicit t_rand / eps the host mirror draws them from the seeded CPU generator in the
    reference's order (RUN:524 -> MOD:234 -> MOD:246): same torch.manual_seed, same render."""
    g = golden("g12_seeded_draws")
    cfg = cfg_from(g)
    _, kw_train, _, model, p, _ = build_model(cfg, int(g["seed"]))
    rays = T(g["rays"]).to(DEV)
    torch.manual_seed(int(g["torch_seed"]))
    with torch.no_grad():
        rgbs, disp, depth, extras = cfnerf_amd.render(int(g["H"]), int(g["W"]), float(g["focal"]), chunk=8192, rays=rays,
                                                      near=0., far=1., **kw_train)
    close(rgbs, g["rgb_map"], what="rgb_map")
    close(depth, g["depth_map"], what="depth_map")
    close(disp, g["disp_map"], atol=ATOL_DISP, rtol=1e-3, what="disp_map")
    close(extras["raw"][:4], g["raw_first4"], what="raw")
    close(extras["loss_entropy"].mean(), g["loss_entropy"], what="loss_entropy")
