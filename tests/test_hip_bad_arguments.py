"""-m gpu: the raw C ABI refuses bad arguments with a status and a message ("shape / flag violations are rejected, not
silently ignored") - NULL pointers, non-positive sizes, K beyond the limit, maps given partly, a stash that does not exist, a
workspace that is too small - never with a crash, and a refused call leaves the handle usable."""
import ctypes as C

import numpy as np
import pytest
import torch

from cfnerf_amd import _lib as L
from oracle import cfnerf_oracle as O
from util_hip import build_model, fern_rays

pytestmark = pytest.mark.gpu
DEV = "cuda"


def refused(rc):
    msg = L.lib().cfnerf_last_error().decode(errors="replace")
    assert rc < 0 and len(msg) > 0, (rc, msg)
    return msg


def test_bad_arguments_are_refused_with_a_status_and_the_handle_stays_usable():
    lib = L.lib()
    cfg = O.OracleCfg(netwidth=64, K_samples=3)
    _, kw_train, kw_test, model, _, _ = build_model(cfg, 3)
    net = model.module
    net._sync()                                                           # (the module re-packs lazily; the raw calls below bypass it)
    h, s = net.handle, L.stream()
    N, S, K = 8, 128, 3
    rng = np.random.default_rng(0)
    rays, (H, W, focal) = fern_rays(rng, N)
    packed = torch.empty(N, 11, device=DEV)
    L.check(lib.cfnerf_rays_setup(H, W, focal, None, L.ptr(rays[0].contiguous().to(DEV)), L.ptr(rays[1].contiguous().to(DEV)), N, 0, 1, 0., 1.,
                                  L.ptr(packed), s), "rays_setup")
    from cfnerf_amd.api import t_vals_table
    tv = t_vals_table(DEV)
    eps = torch.randn(K, 4, device=DEV)
    rgb, disp, depth = torch.empty(N, 3, K, device=DEV), torch.empty(N, K, device=DEV), torch.empty(N, K, device=DEV)
    ent = torch.empty(1, device=DEV)
    P = L.ptr

    def fwd(model=h, rays_=packed, tv_=tv, eps_=eps, n=N, s_=S, k=K, flags=0, rgb_=rgb, disp_=disp, depth_=depth, ent_=None, kstats=None):
        return lib.cfnerf_render_fwd(model, P(rays_), P(tv_), None, None, P(eps_), n, s_, k, flags, P(rgb_), P(disp_), P(depth_), None, None, None,
                                     P(kstats), P(ent_), s)

    rc = fwd()
    assert rc == 0, lib.cfnerf_last_error()                               # the valid call
    torch.cuda.synchronize()
    good = rgb.clone()
    refused(fwd(model=None))
    refused(fwd(rays_=None))
    refused(fwd(eps_=None))
    refused(fwd(tv_=None))                                                # neither t_vals nor explicit depths
    refused(fwd(n=-1))
    refused(fwd(s_=0))
    assert "K_samples" in refused(fwd(k=0))
    assert "K_samples" in refused(fwd(k=100000))
    refused(fwd(disp_=None))                                              # the three maps go together
    refused(fwd(rgb_=None, disp_=None, depth_=None))                      # ... or kstats must be asked for
    assert "entropy_out" in refused(fwd(flags=L.F_TRAIN))                 # TRAIN without entropy_out
    refused(fwd(rgb_=None, disp_=None, depth_=None, kstats=torch.empty(N, 8, device=DEV), k=1))     # kstats needs K >= 2
    assert fwd(n=0) == 0                                                  # an empty batch is fine
    # backward without / against the wrong stash
    grad = torch.empty(net.n_params, device=DEV)
    d_rgb = torch.zeros(N, 3, K, device=DEV)
    gen = lib.cfnerf_model_stash_generation(h)
    refused(lib.cfnerf_render_bwd(h, gen + 5, P(d_rgb), None, None, P(grad), s))
    assert fwd(flags=L.F_STASH, ent_=ent) == 0
    gen = lib.cfnerf_model_stash_generation(h)
    refused(lib.cfnerf_render_bwd(h, gen, None, None, None, P(grad), s))          # d_rgb_map is required
    refused(lib.cfnerf_render_bwd(h, gen, P(d_rgb), None, None, None, s))          # grad_flat is required
    refused(lib.cfnerf_network_bwd(h, gen, None, None, P(grad), s))                # that stash is a ray-mode one
    assert lib.cfnerf_render_bwd(h, gen, P(d_rgb), None, None, P(grad), s) == 0
    # a caller-owned workspace that is too small / misaligned
    need = lib.cfnerf_workspace_bytes(C.byref(net.ccfg) if hasattr(net, "ccfg") else C.byref(L.Cfg(8, 64, 10, 4, 32, 64, 4)), N, S, K)
    assert need > 0
    small = torch.empty(4096, dtype=torch.uint8, device=DEV)
    assert lib.cfnerf_model_set_workspace(h, C.c_void_p(small.data_ptr()), 4096) == 0
    msg = refused(fwd(flags=L.F_STASH, ent_=ent))
    assert "workspace" in msg.lower()
    refused(lib.cfnerf_model_set_workspace(h, C.c_void_p(small.data_ptr() + 4), 2048))          # not 256-byte aligned
    assert lib.cfnerf_model_set_workspace(h, None, 0) == 0                 # back to the model-owned block
    # stateless entries
    refused(lib.cfnerf_composite_fwd(None, None, None, N, S, K, 0, P(rgb), P(disp), P(depth), None, s))
    refused(lib.cfnerf_loss_fwd_bwd(P(rgb), None, None, N, K, C.c_float(0.), N, P(d_rgb), None, s))
    refused(lib.cfnerf_embed(None, 10, 10, None, s))
    refused(lib.cfnerf_adam_step(h, None, None, None, None, 1, C.c_float(1e-3), C.c_float(1.), s))
    refused(lib.cfnerf_model_set_precision(h, 7))
    refused(lib.cfnerf_model_set_flow_math(h, 9))
    # ... and after all of that the handle still computes the same thing
    assert fwd() == 0
    torch.cuda.synchronize()
    assert torch.equal(rgb, good)
