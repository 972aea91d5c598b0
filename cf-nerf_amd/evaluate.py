"""Full-image evaluation on the HIP path (reference: render_path_train RUN:247-314, the
uncertainty maps of RUN:1117-1131, sparsification_plot HLP:382-438).

* ``render_path_train`` mirrors the reference's working single-pose branch and returns numpy ``rgbs [1,H,W,3,K]``,
  ``disps [1,H,W,K]``.
* ``render_uncertainty`` is the MI355X-first form of what the training loop derives from those maps: the K-mean
  prediction, the ``np.std * n/(n-1)`` uncertainty, mean disparity/depth are reduced INSIDE the fused kernel
  (32 B per pixel leave the chip instead of 20*K B), optionally on a row slice of the image so an image is
  tiled across ranks with no exchange.
"""
from __future__ import annotations


import numpy as np
import torch

from . import _lib as L
from .api import _pose_arg, _unwrap, render, t_vals_table


def render_path_train(render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """RUN:247-314.  Only the single-pose branch works in the reference (the multi-pose branch reads an undefined
    ``var`` and unpacks 3 of render()'s 4 results, RUN:279-282 / SURVEY R11); it is rejected here."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor          # RUN:251-255
    poses = torch.as_tensor(render_poses)
    if poses.ndim == 3:
        raise NotImplementedError("the reference's multi-pose branch of render_path_train cannot run (undefined `var`, "
                                  "RUN:279-282); call it once per pose")
    c2w = poses[:3, :4]                                                                      # RUN:304
    rgb, disp, depth, extras = render(int(H), int(W), float(focal), chunk=chunk, c2w=c2w, **render_kwargs)
    return np.stack([rgb.cpu().numpy()], 0), np.stack([disp.cpu().numpy()], 0)              # RUN:307-314


def row_shard(H: int, rank: int, world: int):
    """Rows [r0, r1) of an H-row image for this rank (contiguous, sizes differ by at most one)."""
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def gather_rows(local: dict, H: int, world: int, rank: int, group=None, dst: int = 0):
    """The optional exchange of the row-tiled evaluation: every rank rendered rows ``row_shard(H, rank, world)`` of one
    image with ``render_uncertainty``; rank ``dst`` gets the full-image maps (``rgb_mean [H,W,3]``, ``rgb_unc``, ``disp_mean``,
    ``depth_mean``, and ``sq_err`` when present), the others ``None``.  32 B per pixel travel (20 MB for 800 x 800), once per image;
    the render itself needs no exchange.  Shards differ by at most one row: they are padded to the largest for ``all_gather``.
    A gloo group (CPU tests, ranks sharing one GPU) is served through host copies."""
    import torch.distributed as dist
    keys = [k for k in ("rgb_mean", "rgb_unc", "disp_mean", "depth_mean", "sq_err") if k in local]
    hmax = max(row_shard(H, r, world)[1] - row_shard(H, r, world)[0] for r in range(world))
    via_host = dist.get_backend(group) == "gloo"
    out = {}
    for k in keys:
        t = local[k].contiguous()
        dev = t.device
        if via_host:
            t = t.cpu()
        pad = torch.zeros((hmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        if rank == dst:
            rows = [parts[r][:row_shard(H, r, world)[1] - row_shard(H, r, world)[0]] for r in range(world)]
            out[k] = torch.cat(rows, 0).to(dev)
    if rank != dst:
        return None
    if "sq_err" in out:
        out["mse"] = out["sq_err"].mean()
    return out


@torch.no_grad()
def render_uncertainty(H, W, focal, c2w, network_fn, near=0., far=1., ndc=True, lindisp=False, white_bkgd=False,
                       rows=None, want_maps=False, t_vals=None, gt=None, **_ignored):
    """Eval render of rows ``rows=(r0, r1)`` (default: all) of the image seen from ``c2w`` with the reductions over the
    K latent samples fused into the kernel.  Returns a dict with ``rgb_mean [h,W,3]``, ``rgb_unc [h,W,3]``
    (= ``np.std(rgbs,-1) * n/(n-1)``, RUN:1129-1130), ``disp_mean [h,W]`` (RUN:1124), ``depth_mean [h,W]`` and, if
    ``want_maps``, the per-K ``rgb_map [h,W,3,K]``, ``disp_map``, ``depth_map`` as render() returns them.  With ``gt``
    (ground-truth colours of those rows, ``[h,W,3]``) the per-pixel squared error of the K-mean prediction is produced by the
    same launch (``sq_err [h,W,3]``; ``mse`` = its mean = ``img2mse(rgb_mean, gt)``, RUN:1028)."""
    net = _unwrap(network_fn)
    dev = net.device
    r0, r1 = rows if rows is not None else (0, H)
    n = (r1 - r0) * W
    K = net.K_samples
    if t_vals is None:
        t_vals = t_vals_table(dev)
    S = t_vals.shape[0]
    lib = L.lib()
    packed = torch.empty(n, 11, device=dev)
    arr, _keep = _pose_arg(c2w)
    L.check(lib.cfnerf_rays_setup(int(H), int(W), float(focal), arr, None, None, n, r0 * W, int(bool(ndc)), float(near), float(far),
                                  L.ptr(packed), L.stream()), "cfnerf_rays_setup")
    net._sync()
    eps = net.eval_eps()
    flags = (L.F_LINDISP if lindisp else 0) | (L.F_WHITE_BKGD if white_bkgd else 0)
    kst = torch.empty(n, 8, device=dev)
    rgb = disp = depth = None
    if want_maps:
        rgb, disp, depth = torch.empty(n, 3, K, device=dev), torch.empty(n, K, device=dev), torch.empty(n, K, device=dev)
    sq = None
    if want_maps:
        L.check(lib.cfnerf_render_fwd(net.handle, L.ptr(packed), L.ptr(t_vals), None, None, L.ptr(eps), n, S, K, flags, L.ptr(rgb), L.ptr(disp),
                                      L.ptr(depth), None, None, None, L.ptr(kst), None, L.stream()), "cfnerf_render_fwd")
        if gt is not None:
            sq = (kst[:, 0:3] - gt.to(dev, torch.float32).reshape(n, 3)) ** 2
    else:
        g = None
        if gt is not None:
            g = gt.to(dev, torch.float32).reshape(n, 3).contiguous()
            sq = torch.empty(n, 3, device=dev)
        L.check(lib.cfnerf_render_eval(net.handle, L.ptr(packed), L.ptr(t_vals), L.ptr(eps), n, S, K, flags, L.ptr(g), L.ptr(kst), L.ptr(sq),
                                       L.stream()), "cfnerf_render_eval")
    h = r1 - r0
    out = dict(rgb_mean=kst[:, 0:3].reshape(h, W, 3), rgb_unc=kst[:, 3:6].reshape(h, W, 3), disp_mean=kst[:, 6].reshape(h, W),
               depth_mean=kst[:, 7].reshape(h, W))
    if want_maps:
        out.update(rgb_map=rgb.reshape(h, W, 3, K), disp_map=disp.reshape(h, W, K), depth_map=depth.reshape(h, W, K))
    if sq is not None:
        out.update(sq_err=sq.reshape(h, W, 3), mse=sq.mean())
    return out


def sparsification_plot(var_vec, err_vec, uncert_type='c', err_type='rmse'):
    """HLP:382-438: error of the pixels that remain after removing the top r% by (a) error itself - the oracle
    curve - and (b) predicted uncertainty; r = 0, 1, ..., 99 %.  Returns ``(ause_err, ause_err_by_var)`` as numpy."""
    ratio_removed = np.linspace(0, 1, 100, endpoint=False)
    n = len(err_vec)
    agg = (lambda e: torch.sqrt(e.mean())) if err_type == 'rmse' else (lambda e: e.mean())
    err_sorted, _ = torch.sort(err_vec)
    oracle = np.array([agg(err_sorted[0:int((1 - r) * n)]).cpu().numpy() for r in ratio_removed])
    std = torch.sqrt(var_vec)
    _, idx = torch.sort(std, descending=(uncert_type == 'c'))
    by_var_sorted = err_vec[idx]
    by_var = np.array([agg(by_var_sorted[0:int((1 - r) * n)]).cpu().numpy() for r in ratio_removed])
    return oracle, by_var


def ause(var_vec, err_vec, err_type='rmse'):
    """Area under the sparsification error (mean gap between the by-uncertainty curve and the oracle curve).
    NOTE the reference's ``uncert_type='c'`` sorts DESCENDING and then keeps the head, i.e. it keeps the MOST
    uncertain pixels (HLP:414-416,424); AUSE as usually defined removes them, which is ``uncert_type='v'``."""
    o, v = sparsification_plot(var_vec, err_vec, uncert_type='v', err_type=err_type)
    return float(np.mean(v - o))
