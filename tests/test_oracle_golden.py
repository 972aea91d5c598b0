"""Pin the CPU oracle (oracle/cfnerf_oracle.py) against golden vectors captured from the real
reference (tests/golden/make_golden.py).  fp32 oracle vs fp32 reference: the two run the same
ATen ops in (nearly) the same order, so the tolerance here is far tighter than the HIP one."""
import numpy as np
import pytest
import torch

from oracle import cfnerf_oracle as O

T = lambda a: torch.tensor(np.asarray(a))


def close(a, b, atol=2e-6, rtol=2e-5, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert np.all(err <= tol), f"{what}: max err {err.max():.3e} (tol {tol.flat[err.argmax()]:.3e})"


def cfg_from(g):
    return O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=int(g["K"]), netdepth=int(g.get("netdepth", 8)), n_flows=int(g.get("n_flows", 4)),
                       h_alpha_size=int(g.get("h_alpha_size", 32)), h_rgb_size=int(g.get("h_rgb_size", 64)),
                       multires=int(g.get("multires", 10)), multires_views=int(g.get("multires_views", 4)))


def test_encoder(golden):
    g = golden("g8_encoder")
    assert int(g["d10"]) == 63 and int(g["d4"]) == 27
    close(O.embed(T(g["x"]), 10), g["e10"], atol=0, rtol=0, what="embed10")   # bit-exact: same torch ops
    close(O.embed(T(g["x"]), 4), g["e4"], atol=0, rtol=0, what="embed4")


@pytest.mark.parametrize("tag", ["w64", "w256", "w64k1", "w128k5"])
def test_model_forward(golden, tag):
    g = golden(f"g123_model_{tag}")
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    x = T(g["x90"])
    ea, er = T(g["eps_alpha"]), T(g["eps_rgb"])
    h_alpha, h_rgb = O.mlp_encode(p, x, cfg)
    close(h_alpha, g["h_alpha"], what="h_alpha")
    close(h_rgb, g["h_rgb"], what="h_rgb")
    # eval: the reference zeroes the last latent sample itself (MOD:199,205)
    ea0, er0 = ea.clone(), er.clone()
    ea0[-1] = 0
    er0[-1] = 0
    raw_e, ent_e = O.nerf_flows_forward(p, x, ea0, er0, cfg, is_test=True)
    assert ent_e is None and float(g["aux_eval_absmax"]) == 0.0
    close(raw_e, g["raw_eval"], what="raw_eval")
    raw_t, ent = O.nerf_flows_forward(p, x, ea, er, cfg, is_test=False)
    close(raw_t, g["raw_train"], what="raw_train")
    close(ent, g["loss_entropy"], atol=2e-6, rtol=1e-5, what="loss_entropy")
    assert list(g["loss_entropy_shape"]) == [x.shape[0], cfg.K_samples, 1]


@pytest.mark.parametrize("tag", ["w64", "w256", "w128k5"])
def test_flow_units(golden, tag):
    g = golden(f"g123_model_{tag}")
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    za, lda = O.sylvester_flow(p, "flows_alpha", T(g["z0a"]), T(g["ha"]), cfg.n_flows, False)
    zr, ldr = O.sylvester_flow(p, "flows_rgb", T(g["z0r"]), T(g["hr"]), cfg.n_flows, False)
    close(za, g["za_train"], what="za")
    close(lda, g["lda_train"], what="lda")
    close(zr, g["zr_train"], what="zr")
    close(ldr, g["ldr_train"], what="ldr")
    close(O.sylvester_flow(p, "flows_alpha", T(g["z0a"]), T(g["ha"]), cfg.n_flows, True)[0], g["za_eval"])
    close(O.sylvester_flow(p, "flows_rgb", T(g["z0r"]), T(g["hr"]), cfg.n_flows, True)[0], g["zr_eval"])
    r1, r2, b = O.flow_encode(p, "flows_rgb", T(g["hr"]), 3, cfg.n_flows)
    close(r1, g["r1"], what="r1")
    close(r2, g["r2"], what="r2")
    close(b, g["b"], what="b")


@pytest.mark.parametrize("wb", [False, True])
def test_composite(golden, wb):
    g = golden("g4_composite")
    s = "wb" if wb else "nb"
    rgb_map, disp, w, depth = O.raw2outputs(T(g["raw"]), T(g["z_vals"]), T(g["rays_d"]), wb)
    close(rgb_map, g[f"rgb_map_{s}"], atol=0, rtol=0, what="rgb_map")
    close(disp, g[f"disp_{s}"], atol=0, rtol=0, what="disp")
    close(w, g[f"weights_{s}"], atol=0, rtol=0, what="weights")
    close(depth, g[f"depth_{s}"], atol=0, rtol=0, what="depth")
    # edge rays: the opaque ray's weights sum to 1; the empty ray accumulates ~nothing
    assert np.allclose(g[f"weights_{s}"][0].sum(0), 1.0, atol=1e-5)
    assert np.abs(g[f"weights_{s}"][1]).max() < 1e-12


def _render_inputs(g):
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    rays = T(g["rays"])
    kw = dict(ndc=bool(g["ndc"]), near=float(g["near"]), far=float(g["far"]), lindisp=bool(g["lindisp"]),
              white_bkgd=bool(g["white_bkgd"]))
    return cfg, p, rays, kw


@pytest.mark.parametrize("tag", ["w64_ndc", "w64_nondc_lindisp_wb", "w256_ndc"])
def test_render_batch(golden, tag):
    g = golden(f"g57_render_{tag}")
    cfg, p, rays, kw = _render_inputs(g)
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    ea, er = T(g["eps_alpha"]), T(g["eps_rgb"])
    r = O.render(p, H, W, focal, cfg, ea, er, True, rays=(rays[0], rays[1]), t_rand=T(g["t_rand"]), **kw)
    close(r["pts"], g["pts"], atol=1e-6, what="pts")
    close(r["raw"], g["raw"], atol=5e-6, rtol=5e-5, what="raw")
    close(r["rgb_map"], g["rgb_map"], atol=5e-6, rtol=5e-5, what="rgb_map")
    close(r["depth_map"], g["depth_map"], atol=5e-6, rtol=5e-5, what="depth_map")
    close(r["disp_map"], g["disp_map"], atol=5e-5, rtol=5e-5, what="disp_map")
    close(r["loss_entropy"], g["loss_entropy"], atol=5e-6, rtol=1e-5, what="entropy")
    assert list(g["train_extras_keys"]) == ["loss_entropy", "pts", "raw"]
    assert list(g["eval_extras_keys"]) == []
    ea0, er0 = ea.clone(), er.clone()
    ea0[-1] = 0
    er0[-1] = 0
    e = O.render(p, H, W, focal, cfg, ea0, er0, False, rays=(rays[0], rays[1]), t_rand=None, **kw)
    close(e["rgb_map"], g["rgb_map_eval"], atol=5e-6, rtol=5e-5, what="rgb_map_eval")
    close(e["depth_map"], g["depth_map_eval"], atol=5e-6, rtol=5e-5, what="depth_eval")
    close(e["disp_map"], g["disp_map_eval"], atol=5e-5, rtol=5e-5, what="disp_eval")


@pytest.mark.parametrize("tag", ["w64_ndc", "w64_nondc_lindisp_wb", "w256_ndc"])
def test_train_step(golden, tag):
    g = golden(f"g57_render_{tag}")
    cfg, p, rays, kw = _render_inputs(g)
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    packed = O.pack_rays(H, W, focal, rays[0], rays[1], kw["ndc"], kw["near"], kw["far"])
    scal, grads, ret = O.train_step(p, packed, T(g["target"]), cfg, T(g["eps_alpha"]), T(g["eps_rgb"]),
                                    T(g["t_rand"]), float(g["beta1"]), kw["lindisp"], kw["white_bkgd"])
    close(scal["loss_nll"], g["loss_nll"], rtol=2e-5, what="nll")
    close(scal["loss"], g["loss"], rtol=2e-5, what="loss")
    close(scal["mse"], g["mse"], rtol=2e-5, what="mse")
    close(scal["psnr"], g["psnr"], rtol=2e-5, what="psnr")
    dead = set(g["dead_params"].tolist())
    # R13: the reference leaves exactly these without a gradient
    assert dead == {"alpha_linear.weight", "alpha_linear.bias", "alpha_std_linear.weight", "alpha_std_linear.bias"}
    n_checked = 0
    for k in p:
        if ("grad." + k) in g:
            ref = g["grad." + k]
            scale = max(1e-7, float(np.abs(ref).max()))
            close(grads[k], ref, atol=2e-4 * scale, rtol=1e-3, what="grad " + k)
            n_checked += 1
        elif ("gradrows." + k) in g:
            ref = g["gradrows." + k]
            scale = max(1e-7, float(np.abs(ref).max()))
            close(grads[k][:2], ref, atol=2e-4 * scale, rtol=1e-3, what="gradrows " + k)
            close(grads[k].double().norm(), g["gradnorm." + k], rtol=1e-4, what="gradnorm " + k)
            n_checked += 1
        else:
            assert k in dead, k
            assert grads[k] is None
    assert n_checked >= 30
    # flows_alpha.amor_d is fully masked for z = 1 (R13): zero gradient, not None
    assert float(np.abs(g["grad.flows_alpha.amor_d.weight"]).max()) == 0.0
    # one Adam step (RUN:339,1067)
    state = {}
    newp = O.adam_step({k: v.clone() for k, v in p.items()}, grads, state, step=1, lr=5e-4)
    for k in p:
        if ("adam1." + k) in g:
            close(newp[k], g["adam1." + k], atol=2e-6, rtol=1e-5, what="adam " + k)


def test_render_c2w(golden):
    g = golden("g6_render_c2w")
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    ro, rd = O.get_rays(H, W, focal, T(g["c2w"]))
    close(ro, g["rays_o"], atol=0, rtol=0)
    close(rd, g["rays_d"], atol=1e-7)
    no, nd = O.ndc_rays(H, W, focal, 1.0, ro, rd)
    close(no, g["ndc_o"], atol=1e-6)
    close(nd, g["ndc_d"], atol=1e-6)
    ea, er = T(g["eps_alpha"]).clone(), T(g["eps_rgb"]).clone()
    ea[-1] = 0
    er[-1] = 0
    e = O.render(p, H, W, focal, cfg, ea, er, False, c2w=T(g["c2w"]), near=0., far=1.)
    assert list(e["rgb_map"].shape) == [H, W, 3, 4]
    close(e["rgb_map"], g["rgb_map"], atol=5e-6, rtol=5e-5)
    close(e["depth_map"], g["depth_map"], atol=5e-6, rtol=5e-5)
    close(e["disp_map"], g["disp_map"], atol=5e-5, rtol=5e-5)
    # c2w_staticcam (RUN:139-141): view directions of c2w on the static camera's rays
    e = O.render(p, H, W, focal, cfg, ea, er, False, c2w=T(g["c2w"]), near=0., far=1., c2w_staticcam=T(g["c2w_static"]))
    close(e["rgb_map"], g["rgb_map_static"], atol=5e-6, rtol=5e-5)
    close(e["depth_map"], g["depth_map_static"], atol=5e-6, rtol=5e-5)
    close(e["disp_map"], g["disp_map_static"], atol=5e-5, rtol=5e-5)
    assert (T(g["rgb_map_static"]) - T(g["rgb_map"])).abs().max() > 1e-4     # not degenerate: 10x the tolerance above


def test_oracle_fp64_vs_fp32_drift(golden):
    """The reference's own fp32-vs-fp64 drift bounds what 'fp32 tolerance' can mean."""
    g = golden("g57_render_w64_ndc")
    cfg, p, rays, kw = _render_inputs(g)
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    p64 = {k: v.double() for k, v in p.items()}
    r = O.render(p64, H, W, focal, cfg, T(g["eps_alpha"]).double(), T(g["eps_rgb"]).double(), True,
                 rays=(rays[0].double(), rays[1].double()), t_rand=T(g["t_rand"]).double(), **kw)
    close(r["rgb_map"], g["rgb_map"], atol=1e-5, rtol=1e-4, what="rgb_map fp64 vs ref fp32")
    close(r["depth_map"], g["depth_map"], atol=1e-5, rtol=1e-4, what="depth fp64 vs ref fp32")


def test_seeded_implicit_draw_order(golden):
    """G12: the reference run with NO patched RNG call sites under torch.manual_seed.  Drawing t_rand [N,S], then
    eps_alpha [K,1], then eps_rgb [K,3] from the same seeded CPU generator (RUN:524 -> MOD:234 -> MOD:246) and
    feeding them to the oracle reproduces its outputs."""
    g = golden("g12_seeded_draws")
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    rays = T(g["rays"])
    N, K = rays.shape[1], cfg.K_samples
    torch.manual_seed(int(g["torch_seed"]))
    t_rand = torch.rand([N, 128])
    ea = torch.empty([K, 1]).normal_()
    er = torch.empty([K, 3]).normal_()
    r = O.render(p, int(g["H"]), int(g["W"]), float(g["focal"]), cfg, ea, er, True, rays=(rays[0], rays[1]), t_rand=t_rand)
    close(r["rgb_map"], g["rgb_map"], atol=5e-6, rtol=5e-5)
    close(r["depth_map"], g["depth_map"], atol=5e-6, rtol=5e-5)
    close(r["raw"][:4], g["raw_first4"], atol=5e-6, rtol=5e-5)
    close(r["loss_entropy"], g["loss_entropy"], atol=5e-6, rtol=5e-5)


# ------------------------------------------------------------------ fixtures added in round 2
def test_odd_netdepth_has_no_skip_concat(golden):
    """G13: args.skips = [netdepth / 2] is a float (RUN:327); at netdepth 5 no layer index equals 2.5, so the reference
    builds W x W trunk layers throughout and never concatenates (MOD:39,171).  The oracle mirrors that."""
    g = golden("g13_odd_depth")
    cfg = O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=int(g["K"]), netdepth=int(g["netdepth"]))
    shapes = O.param_shapes(cfg)
    for i in range(cfg.netdepth):
        assert list(g[f"shape.pts_linears.{i}.weight"]) == list(shapes[f"pts_linears.{i}.weight"])
        assert shapes[f"pts_linears.{i}.weight"][1] == (63 if i == 0 else 64)
    p = O.make_params(cfg, int(g["seed"]))
    ea, er = T(g["eps_alpha"]), T(g["eps_rgb"])
    raw_t, ent = O.nerf_flows_forward(p, T(g["x90"]), ea, er, cfg, is_test=False)
    close(raw_t, g["raw_train"], what="raw_train")
    close(ent, g["loss_entropy"], atol=2e-6, rtol=1e-5, what="loss_entropy")
    ea0, er0 = ea.clone(), er.clone()
    ea0[-1] = 0
    er0[-1] = 0
    raw_e, _ = O.nerf_flows_forward(p, T(g["x90"]), ea0, er0, cfg, is_test=True)
    close(raw_e, g["raw_eval"], what="raw_eval")


def test_render_config1_k1(golden):
    """G14 = BASELINE config 1: K = 1, N_rand = 256, default width, forward only (the K = 1 train loss is NaN, R4)."""
    g = golden("g14_render_c1_k1")
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    rays = T(g["rays"])
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    ea, er = T(g["eps_alpha"]), T(g["eps_rgb"])
    r = O.render(p, H, W, focal, cfg, ea, er, True, rays=(rays[0], rays[1]), t_rand=T(g["t_rand"]))
    assert list(r["rgb_map"].shape) == [256, 3, 1]
    close(r["rgb_map"], g["rgb_map"], atol=5e-6, rtol=5e-5, what="rgb_map")
    close(r["depth_map"], g["depth_map"], atol=5e-6, rtol=5e-5, what="depth_map")
    close(r["disp_map"], g["disp_map"], atol=5e-5, rtol=5e-5, what="disp_map")
    close(r["raw"][:4], g["raw_first4"], atol=5e-6, rtol=5e-5, what="raw")
    close(r["loss_entropy"], g["loss_entropy"], atol=5e-6, rtol=5e-5, what="entropy")
    assert list(g["loss_entropy_shape"]) == [256 * 128, 1, 1]
    e = O.render(p, H, W, focal, cfg, torch.zeros_like(ea), torch.zeros_like(er), False, rays=(rays[0], rays[1]), t_rand=None)
    close(e["rgb_map"], g["rgb_map_eval"], atol=5e-6, rtol=5e-5, what="rgb_map_eval")   # K = 1: the only latent is the zeroed last one
    close(e["depth_map"], g["depth_map_eval"], atol=5e-6, rtol=5e-5, what="depth_eval")


def test_three_optimizer_steps(golden):
    """G15: three iterations of the reference's own loop (forward, loss, backward, Adam, learning-rate write-back,
    RUN:1013-1077) with lrate_decay = 1, so that the decay is visible: the oracle's train_step + adam_step + the
    schedule of RUN:1073-1077 land on the same parameters."""
    g = golden("g15_three_steps")
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    rays = T(g["rays"])
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    packed = O.pack_rays(H, W, focal, rays[0], rays[1], True, 0., 1.)
    params, state = {k: v.clone() for k, v in p.items()}, {}
    lr = float(g["lrate"])
    for step in range(int(g["n_steps"])):
        scal, grads, _ = O.train_step(params, packed, T(g["target"]), cfg, T(g[f"eps_alpha{step}"]), T(g[f"eps_rgb{step}"]),
                                      T(g[f"t_rand{step}"]), float(g["beta1"]))
        close(scal["loss"], g[f"loss{step}"], atol=2e-5, rtol=2e-5, what=f"loss at step {step}")
        params = O.adam_step(params, grads, state, step + 1, lr)
        lr = O.lr_schedule(float(g["lrate"]), int(g["lrate_decay"]), step)     # global_step BEFORE its increment (RUN:1075,1198)
        close(lr, g[f"lr_after{step}"], atol=0, rtol=1e-12, what="lr")
    for k in p:
        if ("adam3." + k) in g:
            d = np.abs(params[k].numpy().astype(np.float64) - g["adam3." + k])
            # sign flips of ~0 gradients at step 1 move an entry by up to 2 lr; everything else agrees to fp32 noise
            assert d.max() <= 3 * 2 * 5e-4 + 1e-6, k
            assert (d > 3e-7).mean() <= 0.02, f"{k}: {(d > 3e-7).mean():.3%} entries differ"


def test_adam_on_the_references_own_gradients(golden):
    """G15: the oracle's adam_step fed with the gradients the reference's optimiser saw lands on its parameters after
    each of the three steps (pure elementwise arithmetic: tight)."""
    g = golden("g15_three_steps")
    keys = [str(k) for k in g["adam_keys"]]
    params = {k: T(g["p0." + k]).clone() for k in keys}
    state, lr = {}, float(g["lrate"])
    for step in range(3):
        grads = {k: T(g[f"g{step}." + k]) for k in keys}
        params = O.adam_step(params, grads, state, step + 1, lr)
        for k in keys:
            close(params[k], g[f"p{step + 1}." + k], atol=2e-9, rtol=1e-6, what=f"step {step} {k}")
        lr = float(g[f"lr_after{step}"])


@pytest.mark.parametrize("fixture", ["g16_train_k16", "g18_train_k100"])
def test_train_step_k16(golden, fixture):
    """G16 / G18: loss and every gradient at K = 16 latent samples (BASELINE config 4's count) and at K = 100."""
    g = golden(fixture)
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    rays = T(g["rays"])
    packed = O.pack_rays(int(g["H"]), int(g["W"]), float(g["focal"]), rays[0], rays[1], True, 0., 1.)
    scal, grads, ret = O.train_step(p, packed, T(g["target"]), cfg, T(g["eps_alpha"]), T(g["eps_rgb"]), T(g["t_rand"]), float(g["beta1"]))
    close(ret["rgb_map"], g["rgb_map"], atol=5e-6, rtol=5e-5, what="rgb_map")
    close(scal["loss"], g["loss"], rtol=2e-5, what="loss")
    n = 0
    for k in p:
        if ("grad." + k) in g:
            ref = g["grad." + k]
            scale = max(1e-7, float(np.abs(ref).max()))
            close(grads[k], ref, atol=2e-4 * scale, rtol=1e-3, what="grad " + k)
            n += 1
    assert n >= 30


def test_train_step_wide_config(golden):
    """G17: netwidth 192, netdepth 6, n_flows 3, h_alpha 96, h_rgb 96 - the oracle against the real reference outside the shipped configs"""
    g = golden("g17_train_wide_config")
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    assert list(O.param_shapes(cfg).keys()) == [k for k in g["state_dict_keys"] if "idx" not in k and "mask" not in k]
    rays = T(g["rays"])
    packed = O.pack_rays(int(g["H"]), int(g["W"]), float(g["focal"]), rays[0], rays[1], True, 0., 1.)
    scal, grads, ret = O.train_step(p, packed, T(g["target"]), cfg, T(g["eps_alpha"]), T(g["eps_rgb"]), T(g["t_rand"]), float(g["beta1"]))
    close(ret["rgb_map"], g["rgb_map"], atol=5e-6, rtol=5e-5, what="rgb_map")
    close(ret["raw"][:2], g["raw_first2"], atol=5e-6, rtol=5e-5, what="raw")
    close(scal["loss"], g["loss"], rtol=2e-5, what="loss")
    close(scal["loss_entropy"], g["loss_entropy"], atol=2e-6, rtol=2e-5, what="entropy")
    n = 0
    for k in p:
        if ("grad." + k) in g:
            ref = g["grad." + k]
            close(grads[k], ref, atol=2e-4 * max(1e-7, float(np.abs(ref).max())), rtol=1e-3, what="grad " + k)
            n += 1
        elif ("gradrows." + k) in g:
            ref = g["gradrows." + k]
            close(grads[k][:2], ref, atol=2e-4 * max(1e-7, float(np.abs(ref).max())), rtol=1e-3, what="gradrows " + k)
            close(grads[k].double().norm(), g["gradnorm." + k], atol=0, rtol=1e-4, what="gradnorm " + k)
            n += 1
    assert n >= 26
    ea, er = T(g["eps_alpha"]).clone(), T(g["eps_rgb"]).clone()
    ea[-1] = 0
    er[-1] = 0
    e = O.render_rays(p, packed, cfg, ea, er, False)
    close(e["rgb_map"], g["rgb_map_eval"], atol=5e-6, rtol=5e-5, what="rgb_map_eval")


def test_train_step_non_default_multires(golden):
    """G20: --multires 6 --multires_views 2 (RUN:641-644, HLP:54-69): 39 + 15 input channels through the real reference - model forward on
    a [P, 54] input, render, loss, every gradient, eval render"""
    g = golden("g20_train_multires_6_2")
    cfg = cfg_from(g)
    assert (cfg.multires, cfg.multires_views, cfg.input_ch, cfg.input_ch_views) == (6, 2, int(g["input_ch"]), int(g["input_ch_views"])) == (6, 2, 39, 15)
    shapes = O.param_shapes(cfg)
    for k in ("pts_linears.0.weight", "pts_linears.5.weight", "views_linears.0.weight"):
        assert list(shapes[k]) == list(g["shape." + k]), k
    p = O.make_params(cfg, int(g["seed"]))
    raw_x, ent_x = O.nerf_flows_forward(p, T(g["x54"]), T(g["eps_alpha"]), T(g["eps_rgb"]), cfg, is_test=False)
    close(raw_x, g["raw_x"], atol=5e-6, rtol=5e-5, what="raw of the [P, 54] input")
    close(ent_x, g["loss_entropy_x"], atol=2e-6, rtol=2e-5, what="entropy of the [P, 54] input")
    rays = T(g["rays"])
    packed = O.pack_rays(int(g["H"]), int(g["W"]), float(g["focal"]), rays[0], rays[1], True, 0., 1.)
    scal, grads, ret = O.train_step(p, packed, T(g["target"]), cfg, T(g["eps_alpha"]), T(g["eps_rgb"]), T(g["t_rand"]), float(g["beta1"]))
    close(ret["rgb_map"], g["rgb_map"], atol=5e-6, rtol=5e-5, what="rgb_map")
    close(ret["raw"][:2], g["raw_first2"], atol=5e-6, rtol=5e-5, what="raw")
    close(scal["loss"], g["loss"], rtol=2e-5, what="loss")
    close(scal["loss_entropy"], g["loss_entropy"], atol=2e-6, rtol=2e-5, what="entropy")
    n = 0
    for k in p:
        if ("grad." + k) in g:
            ref = g["grad." + k]
            close(grads[k], ref, atol=2e-4 * max(1e-7, float(np.abs(ref).max())), rtol=1e-3, what="grad " + k)
            n += 1
    assert n >= 30
    ea, er = T(g["eps_alpha"]).clone(), T(g["eps_rgb"]).clone()
    ea[-1] = 0
    er[-1] = 0
    e = O.render_rays(p, packed, cfg, ea, er, False)
    close(e["rgb_map"], g["rgb_map_eval"], atol=5e-6, rtol=5e-5, what="rgb_map_eval")


def test_train_step_authors_recipe_w512(golden):
    """G21: the authors' recipe (train_NF.sh:1-19: netwidth 512, h_alpha 64, h_rgb 64, K 32; africa_ds.txt: no NDC) through the real
    reference: outputs, loss, and of every gradient 64 sampled entries + Frobenius norm + sum; one Adam step on the same entries"""
    g = golden("g21_train_authors_recipe_w512")
    cfg = cfg_from(g)
    assert (cfg.netwidth, cfg.h_alpha_size, cfg.h_rgb_size, cfg.K_samples) == (512, 64, 64, 32)
    p = O.make_params(cfg, int(g["seed"]))
    rays = T(g["rays"])
    packed = O.pack_rays(int(g["H"]), int(g["W"]), float(g["focal"]), rays[0], rays[1], False, float(g["near"]), float(g["far"]))
    scal, grads, ret = O.train_step(p, packed, T(g["target"]), cfg, T(g["eps_alpha"]), T(g["eps_rgb"]), T(g["t_rand"]), float(g["beta1"]))
    close(ret["rgb_map"], g["rgb_map"], atol=5e-6, rtol=5e-5, what="rgb_map")
    close(ret["depth_map"], g["depth_map"], atol=5e-6, rtol=5e-5, what="depth_map")
    close(ret["raw"][:1], g["raw_first1"], atol=5e-6, rtol=5e-5, what="raw")
    close(scal["loss"], g["loss"], rtol=2e-5, what="loss")
    close(scal["loss_entropy"], g["loss_entropy"], atol=2e-6, rtol=2e-5, what="entropy")
    n = 0
    state = {}
    for k in p:
        if ("gradsample." + k) not in g:
            assert grads[k] is None or not grads[k].any(), k
            continue
        idx = T(g["gradidx." + k]).long()
        scale = max(1e-7, float(g["gradabsmax." + k]))
        close(grads[k].reshape(-1)[idx], g["gradsample." + k], atol=2e-4 * scale, rtol=1e-3, what="gradsample " + k)
        close(grads[k].double().norm(), g["gradnorm." + k], atol=0, rtol=2e-4, what="gradnorm " + k)
        close(grads[k].double().sum(), g["gradsum." + k], atol=2e-4 * scale * np.sqrt(grads[k].numel()), rtol=1e-3, what="gradsum " + k)
        n += 1
    assert n >= 30
    live = {k: v for k, v in grads.items() if v is not None}
    new = O.adam_step({k: p[k] for k in live}, live, state, 1, 5e-4)
    for k in live:
        idx = T(g["gradidx." + k]).long()
        d = (new[k].reshape(-1)[idx] - T(g["adam1sample." + k])).abs()
        assert float(d.max()) <= 2 * 5e-4 + 1e-6, k                    # Adam's first step is lr * sign(g): a ~0 gradient may flip
        assert float((d > 2e-5).float().mean()) <= 0.05, k
    ea, er = T(g["eps_alpha"]).clone(), T(g["eps_rgb"]).clone()
    ea[-1] = 0
    er[-1] = 0
    e = O.render_rays(p, packed, cfg, ea, er, False)
    close(e["rgb_map"], g["rgb_map_eval"], atol=5e-6, rtol=5e-5, what="rgb_map_eval")


def _manifest():
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


def _digest(v):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] + ":" + str(v.dtype) + str(list(v.shape))


def test_fixtures_match_the_manifest(golden):
    """Every committed .npz holds exactly the arrays (names, dtypes, shapes, bytes) the committed generator wrote:
    MANIFEST.json is produced by tests/golden/make_golden.py in the same run as the fixtures."""
    import glob
    import os
    from conftest import GOLDEN
    man = _manifest()
    names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert names == sorted(man), (set(names) ^ set(man))
    for name in names:
        g = golden(name)
        assert sorted(g) == sorted(man[name]), (name, set(g) ^ set(man[name]))
        for k, v in g.items():
            assert _digest(v) == man[name][k], (name, k)


def test_psnr_curve_of_the_reference_first_40_steps(golden):
    """G19: the REFERENCE's own training run (its loop lines RUN:1013-1077, 120 steps on a tiny procedural scene).  The oracle,
    driven with the same batches / jitter / latents (re-derived from the fixture's seed), follows its loss and train-batch PSNR
    (RUN:1027-1029) for the first 40 steps and its held-out K-mean PSNR at steps 0 and 40.  (All 120 steps: the -m gpu test.)"""
    import g19_common as GC
    g = golden("g19_psnr_curve")
    H, W, focal, near, far = int(g["H"]), int(g["W"]), float(g["focal"]), float(g["near"]), float(g["far"])
    K, n = int(g["K"]), int(g["n_rand"])
    cfg = cfg_from(g)
    p = O.make_params(cfg, int(g["seed"]))
    poses, images = T(g["poses"]), T(g["images"])
    i_train, i_test = GC.views(g)
    ro_all, rd_all, tg_all = [], [], []
    for v in i_train:
        ro, rd = O.get_rays(H, W, focal, poses[v])
        ro_all.append(ro.reshape(-1, 3)); rd_all.append(rd.reshape(-1, 3)); tg_all.append(images[v].reshape(-1, 3))
    ro_all, rd_all, tg_all = torch.cat(ro_all), torch.cat(rd_all), torch.cat(tg_all)
    ea_e, er_e = T(g["eps_alpha_eval"]).clone(), T(g["eps_rgb_eval"]).clone()
    ea_e[-1] = 0                                                            # MOD:199,205
    er_e[-1] = 0

    def held_out():
        out = []
        for v in i_test:
            with torch.no_grad():
                o = O.render(p, H, W, focal, cfg, ea_e, er_e, False, c2w=poses[v], ndc=False, near=near, far=far)
            out.append(GC.psnr_of(o["rgb_map"].mean(-1), images[v]))
        return out

    close(held_out(), g["psnr_test"][0], atol=1e-4, rtol=0, what="held-out PSNR at step 0")
    rng = np.random.default_rng(int(g["step_seed"]))
    state, sel_sum = {}, 0
    for s in range(40):
        sel, t_rand, ea, er = GC.draws(rng, ro_all.shape[0], n, K)
        sel_sum += int(sel.sum())
        packed = O.pack_rays(H, W, focal, ro_all[sel], rd_all[sel], False, near, far)
        scal, grads, _ = O.train_step(p, packed, tg_all[sel], cfg, ea, er, t_rand, float(g["beta1"]))
        p = O.adam_step(p, grads, state, s + 1, GC.lr_of_step(float(g["lrate"]), int(g["lrate_decay"]), s))
        close(scal["loss"], g["loss"][s], atol=2e-5, rtol=1e-5, what=f"loss at step {s}")
        close(scal["psnr"], g["psnr_train"][s], atol=1e-3, rtol=0, what=f"train-batch PSNR at step {s}")
    close(held_out(), g["psnr_test"][1], atol=1e-3, rtol=0, what="held-out PSNR at step 40")


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/model"), reason="the reference only exists in the build container")
def test_committed_generator_reproduces_the_fixtures(tmp_path):
    """Build container only: re-run the committed generator against the real reference and compare every array of every
    fixture with the committed files bit for bit (fixtures can never drift from the recipe that is said to make them)."""
    import os
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "--out", str(tmp_path)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    man = _manifest()
    import json
    with open(tmp_path / "MANIFEST.json") as f:
        assert json.load(f) == man
    for name in man:
        new = dict(np.load(tmp_path / (name + ".npz"), allow_pickle=False))
        old = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        assert sorted(new) == sorted(old), name
        for k in new:
            assert new[k].dtype == old[k].dtype and new[k].shape == old[k].shape and new[k].tobytes() == old[k].tobytes(), (name, k)


def _g22_tar():
    import os
    from conftest import GOLDEN
    return os.path.join(GOLDEN, "g22_reference_ckpt.tar")


def test_checkpoint_written_by_the_reference(golden):
    """G22 / SURVEY 8f-3: tests/golden/g22_reference_ckpt.tar is what the REFERENCE's own save lines (RUN:1085-1100) wrote for its real
    nn.DataParallel(NeRF_Flows) after two iterations of its loop (a pickled dict of tensors: data).  The committed file is the one the
    generator makes (sha256 recorded in the fixture, which the manifest covers); its dict has the reference's three entries, every
    network key carries the `module.` prefix, and the key LIST - parameters in state_dict order plus the flow stacks' registered buffers -
    is the one this build writes and reads (cfnerf_amd.api.param_layout + flow_buffers: device-free host logic).  The oracle renders
    the reference's eval output from those weights."""
    import hashlib
    import cfnerf_amd
    from cfnerf_amd import _lib as L
    from cfnerf_amd import api
    g = golden("g22_reference_ckpt")
    with open(_g22_tar(), "rb") as f:
        blob = f.read()
    assert len(blob) == int(g["tar_bytes"]) and hashlib.sha256(blob).hexdigest() == str(g["tar_sha256"])
    ck = torch.load(_g22_tar(), map_location="cpu", weights_only=True)
    assert list(ck) == ["global_step", "network_fn_state_dict", "optimizer_state_dict"] and ck["global_step"] == int(g["global_step"]) == 2
    sd = ck["network_fn_state_dict"]
    assert list(sd) == [str(k) for k in g["state_dict_keys"]] and all(k.startswith("module.") for k in sd)
    assert len(ck["optimizer_state_dict"]["state"]) == int(g["optimizer_state_entries"]) > 20           # Adam moments of every live tensor
    cfg = cfg_from(g)
    lcfg = L.Cfg(cfg.netdepth, cfg.netwidth, cfg.multires, cfg.multires_views, cfg.h_alpha_size, cfg.h_rgb_size, cfg.n_flows)
    layout, n_params = cfnerf_amd.param_layout(lcfg)
    ours = set(layout) | set(api.flow_buffers(cfg.n_flows))
    assert {k[len("module."):] for k in sd} == ours, {k[len("module."):] for k in sd} ^ ours
    shapes = O.param_shapes(cfg)
    for k, shp in shapes.items():
        v = sd["module." + k]
        assert tuple(v.shape) == tuple(shp) and v.dtype == torch.float32, k
        assert layout[k][1] == v.numel()
    for k, v in sd.items():
        f = v.reshape(-1)
        assert f.numel() == int(g["numel." + k])
        close(f[:4].double(), g["head." + k], atol=0, rtol=0, what="head " + k)
        close(f.double().sum(), g["sum." + k], atol=0, rtol=0, what="sum " + k)
    for k, v in api.flow_buffers(cfg.n_flows).items():                       # the buffers this build writes ARE the reference's
        assert torch.equal(v.to(sd["module." + k].dtype), sd["module." + k]), k
    p = {k: sd["module." + k].clone() for k in shapes}
    ea, er = T(g["sample_alpha"]).clone(), T(g["sample_rgb"]).clone()
    ea[-1] = 0
    er[-1] = 0
    rays = T(g["rays"])
    e = O.render(p, int(g["H"]), int(g["W"]), float(g["focal"]), cfg, ea, er, False, rays=(rays[0], rays[1]), t_rand=None)
    close(e["rgb_map"], g["rgb_map_eval"], atol=5e-6, rtol=5e-5, what="rgb_map_eval")
    close(e["depth_map"], g["depth_map_eval"], atol=5e-6, rtol=5e-5, what="depth_map_eval")


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/model"), reason="the reference only exists in the build container")
def test_the_references_loader_accepts_a_checkpoint_with_this_builds_key_set(tmp_path):
    """The reverse direction, build container only: a dict with exactly the keys this build's save_checkpoint writes (param_layout +
    flow_buffers under the `module.` prefix; values: the oracle's generator) in the reference's checkpoint format goes through the REAL
    reference's create_nerf(ft_path=...) - its loader lines RUN:345-378 - which resumes at that global_step with those weights."""
    import os
    import subprocess
    import sys
    import cfnerf_amd
    from cfnerf_amd import _lib as L
    from cfnerf_amd import api
    from conftest import GOLDEN, ROOT
    cfg = O.OracleCfg(netwidth=64, netdepth=4, K_samples=3)
    lcfg = L.Cfg(cfg.netdepth, cfg.netwidth, cfg.multires, cfg.multires_views, cfg.h_alpha_size, cfg.h_rgb_size, cfg.n_flows)
    layout, _ = cfnerf_amd.param_layout(lcfg)
    p = O.make_params(cfg, 91)
    assert list(p) == list(layout)
    sd = {"module." + k: v for k, v in p.items()}
    sd.update({"module." + k: v for k, v in api.flow_buffers(cfg.n_flows).items()})
    path = str(tmp_path / "{:06d}_{:02d}.tar".format(777, 1))
    torch.save({"global_step": 777, "network_fn_state_dict": sd, "optimizer_state_dict": {}}, path)      # = api.save_checkpoint's dict
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {GOLDEN!r})\n"
        "import make_golden as MG\n"
        "from oracle import cfnerf_oracle as O\n"
        "R = MG.import_reference()\n"
        "cfg = O.OracleCfg(netwidth=64, netdepth=4, K_samples=3)\n"
        f"args = MG.ref_args(cfg, {str(tmp_path)!r}, K_samples=3, ft_path={path!r}, no_reload=False)\n"
        "kw_train, kw_test, start, grad_vars, optimizer = R.create_nerf(args)\n"
        "assert start == 777, start\n"
        "sd = kw_train['network_fn'].state_dict()\n"
        f"ck = torch.load({path!r})['network_fn_state_dict']\n"
        "assert set(sd) == set(ck), set(sd) ^ set(ck)\n"
        "assert all(torch.equal(sd[k], ck[k].to(sd[k].dtype)) for k in sd)\n"
        "print('LOADED', len(sd))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "LOADED" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
