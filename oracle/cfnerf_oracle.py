"""CPU oracle for the CF-NeRF ray-batch hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional PyTorch-CPU restatement of the reference algorithm
(poetrywanderer/CF-NeRF).  It is the *checker* for the HIP path: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``cf-nerf_amd/`` imports it, and the product path
never falls back to it.

Parity status: PINNED for the reference's path - every function down to
``train_step`` is checked in ``tests/test_oracle_golden.py`` against golden
vectors produced by importing the real reference in the build container
(``tests/golden/make_golden.py``, fixtures ``tests/golden/*.npz``).
PARITY UNPINNED for the EXTENSION section at the end of the file
(``sample_pdf`` / ``render_rays_hierarchical``): the reference has no
hierarchical pass, so those two functions restate its upstream's published
algorithm and have no golden vector.

Abbreviations in the citations (all paths relative to the reference root):
  RUN = run_nerf_uncertainty_NF.py   HLP = run_nerf_helpers.py
  MOD = model/models.py              FLW = model/flow/flows.py

Differences from the reference are deliberate and limited to *plumbing*:
  * random tensors (``t_rand``, ``eps_alpha``, ``eps_rgb``) are explicit inputs
    instead of draws from torch's global generator (RUN:524, MOD:234,246);
  * parameters are a flat ``dict[str, Tensor]`` keyed like
    ``NeRF_Flows.state_dict()`` (without the DataParallel ``module.`` prefix);
  * the dead ``noise`` tensor of RUN:432-440 is not generated;
  * python-level chunk loops (RUN:47-64, RUN:88-100) are dropped: they do not
    change results (RUN:112-113) except for the per-chunk epsilon draw, which the
    explicit-epsilon interface replaces.
The arithmetic (operation order inside each formula, softplus threshold, the
1e1 last-interval, the +1e-10 terms, the n/(n-1) quirk) is kept as shipped.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclass
class OracleCfg:
    """The flags that reach the hot path (RUN:556-719)."""
    netdepth: int = 8            # RUN "--netdepth"
    netwidth: int = 256          # RUN "--netwidth"
    multires: int = 10           # -> input_ch 63 (HLP:54-69)
    multires_views: int = 4      # -> input_ch_views 27
    h_alpha_size: int = 32
    h_rgb_size: int = 64
    n_flows: int = 4
    K_samples: int = 4

    @property
    def input_ch(self) -> int:
        return 3 + 6 * self.multires

    @property
    def input_ch_views(self) -> int:
        return 3 + 6 * self.multires_views

    @property
    def skip(self) -> int:
        # RUN:327  args.skips = [args.netdepth / 2] - a float: for an odd netdepth `i in self.skips` (MOD:39,171) never
        # matches and the trunk has no skip concat (-2 equals no layer index)
        return self.netdepth // 2 if self.netdepth % 2 == 0 else -2


def param_shapes(cfg: OracleCfg) -> Dict[str, Tuple[int, ...]]:
    """state_dict layout of NeRF_Flows (MOD:38-67, MOD:339-350)."""
    W, D = cfg.netwidth, cfg.netdepth
    ic, icv = cfg.input_ch, cfg.input_ch_views
    F_ = cfg.n_flows
    s: Dict[str, Tuple[int, ...]] = {}
    s["alpha_mean"] = (1,)
    s["alpha_std"] = (1,)
    s["rgb_mean"] = (3,)
    s["rgb_std"] = (3,)
    for i in range(D):
        if i == 0:
            k = ic
        elif (i - 1) == cfg.skip:      # MOD:39: layer i (i>=1) is pts_linears[i], built for index i-1 in range(D-1)
            k = W + ic
        else:
            k = W
        s[f"pts_linears.{i}.weight"] = (W, k)
        s[f"pts_linears.{i}.bias"] = (W,)
    s["views_linears.0.weight"] = (W // 2, icv + W)
    s["views_linears.0.bias"] = (W // 2,)
    s["feature_linear.weight"] = (W, W)
    s["feature_linear.bias"] = (W,)
    s["alpha_linear.weight"] = (1, W)           # dead (MOD:59, never used in encode)
    s["alpha_linear.bias"] = (1,)
    s["alpha_std_linear.weight"] = (1, W)       # dead (MOD:60)
    s["alpha_std_linear.bias"] = (1,)
    s["h_alpha_linear.weight"] = (cfg.h_alpha_size, W)
    s["h_alpha_linear.bias"] = (cfg.h_alpha_size,)
    s["h_rgb_linear.weight"] = (cfg.h_rgb_size, W // 2)
    s["h_rgb_linear.bias"] = (cfg.h_rgb_size,)
    for name, z, hs in (("flows_rgb", 3, cfg.h_rgb_size), ("flows_alpha", 1, cfg.h_alpha_size)):
        s[f"{name}.amor_d.weight"] = (F_ * z * z, hs)
        s[f"{name}.amor_d.bias"] = (F_ * z * z,)
        s[f"{name}.amor_diag1.0.weight"] = (F_ * z, hs)
        s[f"{name}.amor_diag1.0.bias"] = (F_ * z,)
        s[f"{name}.amor_diag2.0.weight"] = (F_ * z, hs)
        s[f"{name}.amor_diag2.0.bias"] = (F_ * z,)
        s[f"{name}.amor_b.weight"] = (F_ * z, hs)
        s[f"{name}.amor_b.bias"] = (F_ * z,)
    return s


def make_params(cfg: OracleCfg, seed: int = 0, dtype=torch.float32,
                flow_gain: float = 1.0) -> Dict[str, Tensor]:
    """Deterministic weight generator (numpy PCG64, seed -> state_dict).

    Same family as nn.Linear's default init (uniform in +-1/sqrt(fan_in), the
    bound Kaiming-uniform(a=sqrt 5) reduces to).  The base-Gaussian parameters
    are perturbed away from their (0, 1) init (MOD:44-48) so that mean/std
    actually matter in parity tests.  Fixtures store only seeds, never weights.
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name in ("alpha_mean", "rgb_mean"):
            v = rng.uniform(-0.3, 0.3, size=shape)
        elif name in ("alpha_std", "rgb_std"):
            v = rng.uniform(0.7, 1.3, size=shape)
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:
                # bias: bound uses the fan_in of the matching weight
                wshape = param_shapes(cfg)[name.replace(".bias", ".weight")]
                fan_in = wshape[1]
            bound = 1.0 / math.sqrt(fan_in)
            if name.startswith("flows_"):
                bound *= flow_gain
            v = rng.uniform(-bound, bound, size=shape)
        out[name] = torch.tensor(v, dtype=dtype)
    return out


# --------------------------------------------------------------------------
# positional encoding  (HLP:21-69)
# --------------------------------------------------------------------------
def embed(x: Tensor, multires: int) -> Tensor:
    """gamma(v) = [v, sin(2^0 v), cos(2^0 v), ..., sin(2^(L-1) v), cos(2^(L-1) v)].

    HLP:38 ``freq_bands = 2.**linspace(0, L-1, L)`` (exact powers of two),
    HLP:42-45 order sin then cos per frequency, HLP:51 concatenation.
    """
    outs = [x]
    freqs = 2.0 ** torch.linspace(0.0, multires - 1, steps=multires)
    for f in freqs:
        f = f.to(x.dtype)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


# --------------------------------------------------------------------------
# MLP trunk + heads  (MOD:165-186)
# --------------------------------------------------------------------------
class relu_override:
    """TEST HOOK (not part of the restated algorithm).  ``record``: dict that receives every pre-activation
    (``trunk<i>``, ``views``); ``masks``: dict of 0/1 tensors used INSTEAD of ``pre > 0`` - the gradient tests impose
    the masks the HIP forward took, so that two fp32 implementations that round a ~0 pre-activation to different
    sides are compared on the same piecewise-linear function and every other difference must be at fp32 noise."""
    active = None

    def __init__(self, record=None, masks=None):
        self.record, self.masks = record, masks

    def __enter__(self):
        relu_override.active = self
        return self

    def __exit__(self, *exc):
        relu_override.active = None


def _relu(pre: Tensor, name: str) -> Tensor:
    hook = relu_override.active
    if hook is None:
        return F.relu(pre)
    if hook.record is not None:
        hook.record[name] = pre.detach()
    if hook.masks is not None:
        return pre * hook.masks[name].to(pre.dtype)
    return F.relu(pre)


def mlp_encode(p: Dict[str, Tensor], x: Tensor, cfg: OracleCfg,
               return_acts: bool = False):
    ic, icv = cfg.input_ch, cfg.input_ch_views
    input_pts, input_views = torch.split(x, [ic, icv], dim=-1)            # MOD:166
    h = input_pts
    acts = []
    for i in range(cfg.netdepth):                                          # MOD:168
        h = F.linear(h, p[f"pts_linears.{i}.weight"], p[f"pts_linears.{i}.bias"])
        h = _relu(h, f"trunk{i}")                                          # MOD:170  F.relu
        acts.append(h)
        if i == cfg.skip:                                                  # MOD:171-172 (input first)
            h = torch.cat([input_pts, h], -1)
    h_alpha = F.linear(h, p["h_alpha_linear.weight"], p["h_alpha_linear.bias"])      # MOD:175
    feature = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])      # MOD:176
    h2 = torch.cat([feature, input_views], -1)                                        # MOD:177
    h2 = _relu(F.linear(h2, p["views_linears.0.weight"], p["views_linears.0.bias"]), "views")  # MOD:180-181  F.relu
    h_rgb = F.linear(h2, p["h_rgb_linear.weight"], p["h_rgb_linear.bias"])            # MOD:182
    if return_acts:
        return h_alpha, h_rgb, dict(trunk=acts, feature=feature, views=h2)
    return h_alpha, h_rgb


# --------------------------------------------------------------------------
# triangular Sylvester flow  (MOD:294-416, FLW:168-276)
# --------------------------------------------------------------------------
def flow_encode(p: Dict[str, Tensor], name: str, h: Tensor, z: int, n_flows: int):
    """MOD:358-385.  Returns r1, r2 [B,z,z,F] and b [B,1,z,F]."""
    B = h.shape[0]
    full_d = F.linear(h, p[f"{name}.amor_d.weight"], p[f"{name}.amor_d.bias"])            # MOD:366
    diag1 = torch.tanh(F.linear(h, p[f"{name}.amor_diag1.0.weight"], p[f"{name}.amor_diag1.0.bias"]))  # MOD:367
    diag2 = torch.tanh(F.linear(h, p[f"{name}.amor_diag2.0.weight"], p[f"{name}.amor_diag2.0.bias"]))  # MOD:368
    full_d = full_d.reshape(B, z, z, n_flows)                                              # MOD:370
    diag1 = diag1.reshape(B, z, n_flows)
    diag2 = diag2.reshape(B, z, n_flows)
    triu = torch.triu(torch.ones(z, z, dtype=h.dtype), diagonal=1)[None, :, :, None]      # MOD:327-328
    r1 = full_d * triu                                                                     # MOD:374
    r2 = full_d.transpose(2, 1) * triu                                                     # MOD:375
    idx = torch.arange(z)
    r1 = r1.clone()
    r2 = r2.clone()
    r1[:, idx, idx, :] = diag1                                                             # MOD:377
    r2[:, idx, idx, :] = diag2                                                             # MOD:378
    b = F.linear(h, p[f"{name}.amor_b.weight"], p[f"{name}.amor_b.bias"])                 # MOD:380
    b = b.reshape(B, 1, z, n_flows)                                                        # MOD:383
    return r1, r2, b


def sylvester_step(zk: Tensor, r1: Tensor, r2: Tensor, b: Tensor, flip: bool, with_logdet: bool):
    """One TriangularSylvester._forward (FLW:189-268).  zk [B,z]; r1,r2 [B,z,z]; b [B,1,z]."""
    zk_ = zk.unsqueeze(1)                                                  # FLW:206/226
    z_per = torch.flip(zk_, dims=[2]) if flip else zk_                     # FLW:209/234 (flip_idx = reversed arange, MOD:323)
    r2qzb = torch.bmm(z_per, r2.transpose(2, 1)) + b                       # FLW:213/238
    t = torch.tanh(r2qzb)
    z = torch.bmm(t, r1.transpose(2, 1))                                   # FLW:214/239
    if flip:
        z = torch.flip(z, dims=[2])                                        # FLW:218/243
    z = (z + zk_).squeeze(1)                                               # FLW:220-221
    if not with_logdet:
        return z, None
    idx = torch.arange(zk.shape[-1])
    diag_j = r1[:, idx, idx] * r2[:, idx, idx]                             # FLW:229-230,251
    diag_j = (1.0 - t ** 2).squeeze(1) * diag_j                            # FLW:252 (der_tanh FLW:187)
    diag_j = diag_j + 1.0                                                  # FLW:253
    diag_j = diag_j.abs() + 1e-08                                          # FLW:255
    return z, torch.log(diag_j).sum(-1)                                    # FLW:259-262


def sylvester_flow(p: Dict[str, Tensor], name: str, z0: Tensor, h: Tensor, n_flows: int, is_test: bool):
    """TriangularSylvesterNeRF.forward (MOD:387-416): 4 steps, odd steps flipped."""
    zdim = z0.shape[-1]
    r1, r2, b = flow_encode(p, name, h, zdim, n_flows)
    z = z0
    logdet = torch.zeros(z0.shape[0], dtype=z0.dtype)
    for k in range(n_flows):
        z, ld = sylvester_step(z, r1[..., k], r2[..., k], b[..., k], flip=(k % 2 == 1),
                               with_logdet=not is_test)                    # MOD:404-410
        if ld is not None:
            logdet = logdet + ld                                           # MOD:413
    return z, logdet


# --------------------------------------------------------------------------
# NeRF_Flows.forward  (MOD:188-291)
# --------------------------------------------------------------------------
# TEST HOOK (not part of the restated algorithm): a dict that receives alpha0 [P K,1] / rgb0 [P K,3] of the next forward WITH their
# autograd history, so a gradient test can ask for d loss / d alpha0 per (point, latent): the gradient of alpha_mean is the plain
# sum of those terms, and its fp32 rounding error scales with the sum of their magnitudes, not with the (cancelling) sum itself.
latent_tap = None


def nerf_flows_forward(p: Dict[str, Tensor], x: Tensor, eps_alpha: Tensor, eps_rgb: Tensor,
                       cfg: OracleCfg, is_test: bool):
    """x [P,90]; eps_alpha [K,1]; eps_rgb [K,3] -> (raw [P,K,4], entropy scalar | None).

    Eval callers pass the module's fixed buffers with the LAST sample already
    zeroed (MOD:199,205); train callers pass fresh N(0,1) draws (MOD:234,246).
    """
    K = eps_alpha.shape[0]
    h_alpha, h_rgb = mlp_encode(p, x, cfg)                                  # MOD:190
    BN = h_alpha.shape[0]
    alpha_mean_k = p["alpha_mean"][None, None, :].expand(BN, K, 1)          # MOD:196/229
    alpha_std_k = p["alpha_std"][None, None, :].expand(BN, K, 1)
    alpha0 = (eps_alpha[None].expand(BN, K, 1) * alpha_std_k + alpha_mean_k).reshape(-1, 1)   # MOD:200/239
    rgb_mean_k = p["rgb_mean"][None, None, :].expand(BN, K, 3)
    rgb_std_k = p["rgb_std"][None, None, :].expand(BN, K, 3)
    rgb0 = (eps_rgb[None].expand(BN, K, 3) * rgb_std_k + rgb_mean_k).reshape(-1, 3)           # MOD:206/251
    if latent_tap is not None:          # TEST HOOK: the base samples as graph nodes (per-element gradient contributions of the base Gaussians)
        latent_tap["alpha0"], latent_tap["rgb0"] = alpha0, rgb0

    ha = h_alpha[:, None, :].expand(BN, K, cfg.h_alpha_size).reshape(-1, cfg.h_alpha_size)    # MOD:210-211
    hr = h_rgb[:, None, :].expand(BN, K, cfg.h_rgb_size).reshape(-1, cfg.h_rgb_size)          # MOD:215-216
    z_alpha, ld_alpha = sylvester_flow(p, "flows_alpha", alpha0, ha, cfg.n_flows, is_test)    # MOD:212/257
    z_rgb, ld_rgb = sylvester_flow(p, "flows_rgb", rgb0, hr, cfg.n_flows, is_test)            # MOD:217/273
    z_k_alpha = z_alpha.reshape(BN, K, 1)
    z_k_rgb = z_rgb.reshape(BN, K, 3)
    raw = torch.cat([z_k_rgb, z_k_alpha], -1)                                                 # MOD:221/289
    if is_test:
        return raw, None                                                                      # MOD:223 (zeros aux)

    ld_alpha = ld_alpha.reshape(BN, K) + (z_k_alpha.sum(-1) - F.softplus(z_k_alpha).sum(-1))  # MOD:259,263
    a0 = alpha0.reshape(BN, K, 1)
    base_a = -0.5 * (alpha_std_k.log() * 2 + (a0 - alpha_mean_k) * (a0 - alpha_mean_k)
                     * (alpha_std_k ** 2).reciprocal())                                       # MOD:268
    ld_rgb = ld_rgb.reshape(BN, K) + (z_k_rgb.sum(-1) - 2 * F.softplus(z_k_rgb).sum(-1))      # MOD:275,278
    r0 = rgb0.reshape(BN, K, 3)
    base_r = -0.5 * (rgb_std_k.log() * 2 + (r0 - rgb_mean_k) * (r0 - rgb_mean_k)
                     * (rgb_std_k ** 2).reciprocal())                                         # MOD:283
    loss_entropy = base_a.mean() - ld_alpha.mean() + base_r.mean() - ld_rgb.mean()            # MOD:286
    return raw, loss_entropy


# --------------------------------------------------------------------------
# network query (RUN:67-85)
# --------------------------------------------------------------------------
def run_network(p, pts: Tensor, viewdirs: Tensor, eps_alpha, eps_rgb, cfg: OracleCfg, is_test: bool):
    """pts [N,S,3], viewdirs [N,3] -> raw [N,S,K,4], entropy."""
    flat = pts.reshape(-1, 3)                                               # RUN:70
    e = embed(flat, cfg.multires)                                           # RUN:71
    dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)               # RUN:75-78
    e = torch.cat([e, embed(dirs, cfg.multires_views)], -1)                 # RUN:79-80
    raw, ent = nerf_flows_forward(p, e, eps_alpha, eps_rgb, cfg, is_test)   # RUN:82
    return raw.reshape(list(pts.shape[:-1]) + list(raw.shape[-2:])), ent    # RUN:83


# --------------------------------------------------------------------------
# volumetric composite (RUN:411-454)
# --------------------------------------------------------------------------
def raw2outputs(raw: Tensor, z_vals: Tensor, rays_d: Tensor, white_bkgd: bool = False):
    """raw [N,S,K,4], z_vals [N,S], rays_d [N,3] -> rgb_map [N,3,K], disp [N,K], weights [N,S,K], depth [N,K]."""
    dt = raw.dtype
    dists = z_vals[..., 1:] - z_vals[..., :-1]                              # RUN:426
    dists = torch.cat([dists, torch.tensor([1e1], dtype=dt).expand(dists[..., :1].shape)], -1)   # RUN:427
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)                # RUN:429
    rgb = torch.sigmoid(raw[..., :3])                                       # RUN:431
    alpha = 1.0 - torch.exp(-F.softplus(raw[..., 3]) * dists[..., None])    # RUN:424,442
    ones = torch.ones((alpha.shape[0], 1, alpha.shape[-1]), dtype=dt)
    weights = alpha * torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-10], -2), -2)[:, :-1, :]   # RUN:443
    rgb_map = torch.sum(weights[..., None] * rgb, -3).transpose(-1, -2)     # RUN:444-445
    depth_map = torch.sum(weights * z_vals[..., None], -2)                  # RUN:447
    acc_map = torch.sum(weights, -2)                                        # RUN:449
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map) + 1e-10,
                               depth_map / (acc_map + 1e-10) + 1e-10)       # RUN:448
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[:, None, :])                     # RUN:451-452
    return rgb_map, disp_map, weights, depth_map


# --------------------------------------------------------------------------
# sampling along rays + render_rays (RUN:457-553)
# --------------------------------------------------------------------------
def t_vals_table(dtype=torch.float32) -> Tensor:
    """RUN:510 - the hard-coded 96+32 = 128 sample table."""
    return torch.cat([torch.linspace(0.0, 0.5, steps=97)[:-1], torch.linspace(0.5, 1.0, steps=32)], 0).to(dtype)


def sample_z(near: Tensor, far: Tensor, t_vals: Tensor, lindisp: bool, t_rand: Optional[Tensor]):
    """RUN:511-532.  near, far [N,1]; t_vals [S]; t_rand [N,S] or None (perturb == 0)."""
    if not lindisp:
        z_vals = near * (1.0 - t_vals) + far * t_vals                      # RUN:512
    else:
        z_vals = 1.0 / (1.0 / near * (1.0 - t_vals) + 1.0 / far * t_vals)  # RUN:514
    z_vals = z_vals.expand([near.shape[0], t_vals.shape[0]])               # RUN:516
    if t_rand is not None:
        mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])                  # RUN:520
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * t_rand                          # RUN:532
    return z_vals


def render_rays(p, ray_batch: Tensor, cfg: OracleCfg, eps_alpha, eps_rgb, is_train: bool,
                t_rand: Optional[Tensor] = None, lindisp: bool = False, white_bkgd: bool = False,
                t_vals: Optional[Tensor] = None, z_vals: Optional[Tensor] = None):
    """ray_batch [N,11] = o3,d3,near,far,viewdir3 (RUN:504-507).  ``z_vals`` [N,S] (EXTENSION only: explicit depths of
    the fine pass) replaces the sampling lines RUN:510-532."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:]
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    if t_vals is None:
        t_vals = t_vals_table(ray_batch.dtype)
    if z_vals is None:
        z_vals = sample_z(near, far, t_vals, lindisp, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]      # RUN:534
    raw, ent = run_network(p, pts, viewdirs, eps_alpha, eps_rgb, cfg, is_test=not is_train)   # RUN:537-538
    rgb_map, disp_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, white_bkgd)     # RUN:540
    ret = dict(rgb_map=rgb_map, disp_map=disp_map, depth_map=depth_map, weights=weights, z_vals=z_vals, rays_d=rays_d)     # (z_vals, rays_d: test conveniences)
    if is_train:
        ret.update(raw=raw, loss_entropy=ent, pts=pts)                      # RUN:544-547
    return ret


# --------------------------------------------------------------------------
# ray helpers (HLP:288-297, HLP:360-377) and render (RUN:103-170)
# --------------------------------------------------------------------------
def get_rays(H: int, W: int, focal: float, c2w: Tensor):
    dt = c2w.dtype
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")   # HLP:289
    i = i.t().to(dt)
    j = j.t().to(dt)
    dirs = torch.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -torch.ones_like(i)], -1)       # HLP:292
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)                                          # HLP:294
    rays_o = c2w[:3, -1].expand(rays_d.shape)                                                         # HLP:296
    return rays_o, rays_d


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o: Tensor, rays_d: Tensor):
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]                           # HLP:362
    rays_o = rays_o + t[..., None] * rays_d                                 # HLP:363
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]         # HLP:366
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])   # HLP:370
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)     # HLP:374-375


def pack_rays(H, W, focal, rays_o: Tensor, rays_d: Tensor, ndc: bool, near: float, far: float, c2w_staticcam=None):
    """RUN:136-158: viewdirs from the world direction (before NDC), then NDC, then the [N,11] pack."""
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)            # RUN:143 (taken before the staticcam swap)
    viewdirs = viewdirs.reshape(-1, 3)
    if c2w_staticcam is not None:
        rays_o, rays_d = get_rays(H, W, focal, c2w_staticcam)               # RUN:139-141
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)          # RUN:149
    rays_o = rays_o.reshape(-1, 3)
    rays_d = rays_d.reshape(-1, 3)
    nr = near * torch.ones_like(rays_d[..., :1])                            # RUN:155
    fr = far * torch.ones_like(rays_d[..., :1])
    return torch.cat([rays_o, rays_d, nr, fr, viewdirs], -1)                # RUN:156-158


def render(p, H, W, focal, cfg: OracleCfg, eps_alpha, eps_rgb, is_train: bool, rays=None, c2w=None,
           ndc=True, near=0., far=1., t_rand=None, lindisp=False, white_bkgd=False, c2w_staticcam=None):
    """RUN:103-170 restated; returns the render_rays dict reshaped to the ray batch shape."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, focal, c2w)                         # RUN:131
    else:
        rays_o, rays_d = rays                                               # RUN:134
    sh = rays_d.shape
    packed = pack_rays(H, W, focal, rays_o, rays_d, ndc, near, far, c2w_staticcam)
    ret = render_rays(p, packed, cfg, eps_alpha, eps_rgb, is_train, t_rand, lindisp, white_bkgd)
    out = {}
    for k, v in ret.items():
        if k == "loss_entropy" or v is None:
            out[k] = v
        else:
            out[k] = v.reshape(list(sh[:-1]) + list(v.shape[1:]))           # RUN:162-165
    return out


# --------------------------------------------------------------------------
# train-step loss (RUN:1026-1050) and optimiser (RUN:339, 1065-1077)
# --------------------------------------------------------------------------
def train_loss(rgbs: Tensor, target_s: Tensor, loss_entropy: Tensor, K: int, beta1: float):
    """rgbs [N,3,K], target_s [N,3].  Returns dict(loss, loss_nll, mse, psnr)."""
    rgb_mean = torch.mean(rgbs, -1)                                         # RUN:1027
    mse = torch.mean((rgb_mean - target_s) ** 2)                            # RUN:1028, HLP:15
    psnr = -10. * torch.log(mse) / math.log(10.)                            # HLP:16
    eps = 1e-05
    n = K
    rgb_std = torch.std(rgbs, -1) * n / (n - 1)                             # RUN:1034 (n/(n-1) on top of unbiased std)
    H_sqrt = rgb_std.detach() * torch.pow(torch.tensor(0.8 / n), torch.tensor(-1 / 7)).to(rgbs.dtype) + eps   # RUN:1036
    H_sqrt = H_sqrt[..., None]
    r1 = torch.exp(-((rgbs - target_s[..., None]) ** 2) / (2 * H_sqrt * H_sqrt))          # RUN:1038
    r2 = torch.pow(torch.tensor(2 * math.pi), -1.5).to(rgbs.dtype) / H_sqrt               # RUN:1039
    r_mean = (r1 * r2).mean(-1) + eps                                       # RUN:1040-1041
    loss_nll = -torch.log(r_mean).mean()                                    # RUN:1042
    loss = loss_nll + beta1 * loss_entropy if beta1 else loss_nll           # RUN:1047-1050
    return dict(loss=loss, loss_nll=loss_nll, mse=mse, psnr=psnr)


def adam_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict[str, Dict[str, Tensor]],
              step: int, lr: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """torch.optim.Adam defaults (RUN:339): no weight decay, no amsgrad.  ``step`` is 1-based."""
    for k, g in grads.items():
        if g is None:
            continue
        st = state.setdefault(k, dict(m=torch.zeros_like(params[k]), v=torch.zeros_like(params[k])))
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k] = params[k] - (lr / bc1) * st["m"] / denom
    return params


def lr_schedule(lrate: float, lrate_decay: int, global_step: int) -> float:
    """RUN:1073-1077."""
    return lrate * (0.1 ** (global_step / (lrate_decay * 1000)))


def train_step(p: Dict[str, Tensor], packed_rays: Tensor, target_s: Tensor, cfg: OracleCfg,
               eps_alpha: Tensor, eps_rgb: Tensor, t_rand: Optional[Tensor], beta1: float,
               lindisp=False, white_bkgd=False, t_vals: Optional[Tensor] = None):
    """Forward + loss + backward on the oracle.  Returns (scalars, grads dict).  `t_vals`: a sample table other than the
    reference's hard-coded 128 entries (the HIP path accepts one through `t_vals=`; the loss lines are the reference's)."""
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ret = render_rays(q, packed_rays, cfg, eps_alpha, eps_rgb, True, t_rand, lindisp, white_bkgd, t_vals=t_vals)
    L = train_loss(ret["rgb_map"], target_s, ret["loss_entropy"], cfg.K_samples, beta1)
    L["loss"].backward()
    grads = {k: v.grad for k, v in q.items()}
    scal = {k: float(v.detach()) for k, v in L.items()}
    scal["loss_entropy"] = float(ret["loss_entropy"].detach())
    return scal, grads, {k: v.detach() for k, v in ret.items() if v is not None}


# --------------------------------------------------------------------------
# EXTENSION oracle - PARITY UNPINNED by poetrywanderer/CF-NeRF: the reference has no hierarchical pass
# (N_importance / network_fine are dead parameters, RUN:467-468; `searchsorted` survives only in a comment,
# HLP:9-11).  What is restated here is the published algorithm of its upstream, yenchenlin/nerf-pytorch
# (run_nerf_helpers.py `sample_pdf`, run_nerf.py render_rays lines "if N_importance > 0"), adapted to K latent
# samples by driving the resampling with the K-mean of the coarse weights.  No golden vector exists for it.
# --------------------------------------------------------------------------
def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tensor:
    """bins [N,M], weights [N,M-1], u [N,Ni] in [0,1] -> samples [N,Ni] (inverse-CDF sampling)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    matched_shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(matched_shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(matched_shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


def render_rays_hierarchical(p, ray_batch: Tensor, cfg: OracleCfg, eps_alpha, eps_rgb, is_train: bool, t_vals_coarse: Tensor,
                             u: Tensor, t_rand: Optional[Tensor] = None, lindisp=False, white_bkgd=False):
    """Coarse pass on t_vals_coarse, resample u.shape[1] depths from the K-mean coarse weights, fine pass on the
    merged sorted depths through the SAME network (network_fine is None in the reference)."""
    coarse = render_rays(p, ray_batch, cfg, eps_alpha, eps_rgb, is_train, t_rand, lindisp, white_bkgd, t_vals=t_vals_coarse)
    z = coarse["z_vals"]
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
    w = coarse["weights"].mean(-1)                                   # [N,S]
    z_samples = sample_pdf(z_mid, w[..., 1:-1], u).detach()
    z_all, _ = torch.sort(torch.cat([z, z_samples], -1), -1)
    fine = render_rays(p, ray_batch, cfg, eps_alpha, eps_rgb, is_train, None, lindisp, white_bkgd, z_vals=z_all)
    rgb_map, disp_map, weights, depth_map, ent = fine["rgb_map"], fine["disp_map"], fine["weights"], fine["depth_map"], fine.get("loss_entropy")
    return dict(rgb_map=rgb_map, disp_map=disp_map, depth_map=depth_map, weights=weights, z_vals=z_all, z_samples=z_samples,
                rgb0=coarse["rgb_map"], disp0=coarse["disp_map"], depth0=coarse["depth_map"], loss_entropy=ent)
