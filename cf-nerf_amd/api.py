"""Host-side mirror of the reference's hot-path interface, backed by libcfnerf_hip.so.

Same names, argument order, defaults and return structure as the reference
(RUN = run_nerf_uncertainty_NF.py, HLP = run_nerf_helpers.py, MOD = model/models.py):

    render            RUN:103-170      render_rays     RUN:457-553
    run_network       RUN:67-85        raw2outputs     RUN:411-454
    batchify(_rays)   RUN:47-64,88-100 create_nerf     RUN:317-409
    NeRF_Flows        MOD:13-291       get_embedder    HLP:54-69
    get_rays/ndc_rays HLP:288-297,360-377   img2mse/mse2psnr HLP:15-16

All arithmetic of the path runs in hand-written HIP kernels; PyTorch only owns device memory,
streams, autograd plumbing and the optimiser.  There is no CPU / eager fallback: every function
raises if the library is missing or the tensors are not on a GPU.

Extra (new) keyword arguments make the reference's hidden randomness explicit:
``t_rand [N,S]``, ``eps_alpha [K,1]``, ``eps_rgb [K,3]``.  When left ``None`` they are drawn with
torch in the reference's order (RUN:524 -> MOD:234 -> MOD:246).
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L

# --------------------------------------------------------------------------------------------
# misc helpers (HLP:15-16)
img2mse = lambda x, y: torch.mean((x - y) ** 2)
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the GPU: the CF-NeRF hot path is HIP-only (no CPU fallback)")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# --------------------------------------------------------------------------------------------
# positional encoding (HLP:21-69).  Used only by the unfused run_network path and by callers that
# want the embedding itself; the fused kernels encode on-chip.
class Embedder:
    def __init__(self, multires: int, input_dims: int = 3):
        self.multires = multires
        self.input_dims = input_dims
        self.out_dim = input_dims * (1 + 2 * multires)
        self.freq_bands = 2. ** torch.linspace(0., multires - 1, steps=multires)     # HLP:38

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:
        _need_gpu(inputs, "Embedder input")                                           # HIP kernel only (cfnerf_embed)
        if inputs.shape[-1] != self.input_dims or self.input_dims != 3:
            raise ValueError("Embedder encodes 3-vectors (HLP:57-64: input_dims = 3)")
        x = _f32c(inputs.reshape(-1, 3))
        out = torch.empty(x.shape[0], self.out_dim, device=x.device)
        L.check(L.lib().cfnerf_embed(L.ptr(x), x.shape[0], self.multires, L.ptr(out), L.stream()), "cfnerf_embed")
        return out.reshape(list(inputs.shape[:-1]) + [self.out_dim])


def get_embedder(multires, i=0):
    if i == -1:
        return nn.Identity(), 3                                                       # HLP:55-56
    eo = Embedder(multires)
    fn = lambda x, eo=eo: eo.embed(x)
    fn.multires = multires
    return fn, eo.out_dim


# --------------------------------------------------------------------------------------------
# ray helpers (HLP:288-297, 360-377) as standalone calls: the same kernels render() uses (cfnerf_rays_setup with a pose,
# cfnerf_ndc_rays).  GPU only, like everything else here.
def _pose_arg(c2w):
    """A [3,4] pose as the `const float* c2w_host` argument.  A pose that already lives on the host is passed as it is
    (no copy when it is contiguous fp32); a device pose is fetched once (the call needs it on the host: it travels in
    the kernel arguments).  Returns (ctypes pointer, keep-alive tensor)."""
    t = torch.as_tensor(c2w)
    if t.is_cuda:
        t = t.detach().cpu()
    t = t.detach()[:3, :4].to(torch.float32).contiguous()
    return C.cast(t.data_ptr(), C.POINTER(C.c_float)), t


def _device_of(*ts, default=None):
    for t in ts:
        if torch.is_tensor(t) and t.is_cuda:
            return t.device
    return torch.device("cuda", torch.cuda.current_device()) if default is None else default


def get_rays(H, W, focal, c2w):
    """HLP:288-297 on the ray set-up kernel: ``rays_o, rays_d [H,W,3]`` on the GPU (of ``c2w`` if it lives on one, else the
    current device)."""
    H, W = int(H), int(W)
    dev = _device_of(c2w)
    arr, keep = _pose_arg(c2w)
    packed = torch.empty(H * W, 11, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().cfnerf_rays_setup(H, W, float(focal), arr, None, None, H * W, 0, 0, 0.0, 1.0, L.ptr(packed), L.stream()),
                "cfnerf_rays_setup")
    return packed[:, 0:3].reshape(H, W, 3), packed[:, 3:6].reshape(H, W, 3)


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """HLP:360-377 on ``cfnerf_ndc_rays``."""
    _need_gpu(rays_d, "rays_d")
    ro, rd = _f32c(rays_o.reshape(-1, 3)), _f32c(rays_d.reshape(-1, 3))
    o, d = torch.empty_like(ro), torch.empty_like(rd)
    L.check(L.lib().cfnerf_ndc_rays(int(H), int(W), float(focal), float(near), L.ptr(ro), L.ptr(rd), ro.shape[0], L.ptr(o), L.ptr(d),
                                    L.stream()), "cfnerf_ndc_rays")
    return o.reshape(rays_o.shape), d.reshape(rays_d.shape)


_T_VALS_CACHE = {}


def t_vals_table(device=None) -> torch.Tensor:
    """The hard-coded 96 + 32 = 128 sample table of RUN:510 (computed on the CPU like the reference; the device copy
    is made once per device - treat it as read-only)."""
    key = str(torch.device(device)) if device is not None else "cpu"
    t = _T_VALS_CACHE.get(key)
    if t is None:
        t = torch.cat([torch.linspace(0., 0.5, steps=97)[:-1], torch.linspace(0.5, 1., steps=32)], 0)
        if device is not None:
            t = t.to(device)
        _T_VALS_CACHE[key] = t
    return t


# --------------------------------------------------------------------------------------------
# parameter layout (mirror of cfnerf_param_key / cfnerf_param_offset)
def _cfg_struct(netdepth, netwidth, multires, multires_views, h_alpha_size, h_rgb_size, n_flows) -> L.Cfg:
    return L.Cfg(int(netdepth), int(netwidth), int(multires), int(multires_views), int(h_alpha_size),
                 int(h_rgb_size), int(n_flows))


def param_layout(cfg: L.Cfg):
    """OrderedDict key -> (offset, shape) in the flat parameter buffer, as the C ABI defines it."""
    lib = L.lib()
    total = lib.cfnerf_param_count(C.byref(cfg))
    if total < 0:
        raise RuntimeError("unsupported configuration: " + lib.cfnerf_last_error().decode())
    out = OrderedDict()
    i = 0
    W, D = cfg.netwidth, cfg.netdepth
    while True:
        k = lib.cfnerf_param_key(C.byref(cfg), i)
        if k is None:
            break
        key = k.decode()
        n = C.c_int64(0)
        off = lib.cfnerf_param_offset(C.byref(cfg), k, C.byref(n))
        out[key] = (int(off), int(n.value))
        i += 1
    return out, int(total)


def _shape_of(key: str, numel: int, cfg: L.Cfg):
    if key.endswith(".weight"):
        W, ic, icv = cfg.netwidth, 3 + 6 * cfg.multires, 3 + 6 * cfg.multires_views
        rows = {"views_linears.0.weight": W // 2, "feature_linear.weight": W, "alpha_linear.weight": 1,
                "alpha_std_linear.weight": 1, "h_alpha_linear.weight": cfg.h_alpha_size,
                "h_rgb_linear.weight": cfg.h_rgb_size}.get(key)
        if rows is None:
            if key.startswith("pts_linears."):
                rows = W
            elif key.startswith("flows_rgb."):
                rows = cfg.n_flows * (9 if ".amor_d." in key else 3)
            else:
                rows = cfg.n_flows
        return (rows, numel // rows)
    return (numel,)


# --------------------------------------------------------------------------------------------
@torch.no_grad()
def flow_buffers(n_flows, device="cpu"):
    """The registered buffers of the reference's two flow stacks (MOD:323-333, FLW:180-181), which its ``state_dict()`` carries next to
    the parameters: key -> tensor.  Device-free host logic (a checkpoint this build writes holds exactly the reference's key set:
    ``param_layout`` keys + these)."""
    out = OrderedDict()
    for name, z in (("flows_rgb", 3), ("flows_alpha", 1)):
        out[f"{name}.flip_idx"] = torch.arange(z - 1, -1, -1, device=device).long()             # MOD:323
        out[f"{name}.triu_mask"] = torch.triu(torch.ones(z, z, device=device), diagonal=1)[None, :, :, None]
        out[f"{name}.diag_idx"] = torch.arange(0, z, device=device).long()
        for k in range(n_flows):
            out[f"{name}.flow_{k}.diag_idx"] = torch.arange(0, z, device=device).long()         # FLW:180-181
    return out


def reference_init(shapes, netdepth, K_samples=None):
    """Initial values exactly as ``NeRF_Flows.__init__`` produces them under the same ``torch.manual_seed``
    (MOD:38-67, 339-350).  Throw-away CPU ``nn.Linear``s are created in the reference's construction order, so torch's
    own default init (U(+-1/sqrt(fan_in)) for weight and bias) consumes the CPU generator identically, and the fixed
    latents are drawn at the same point of the stream (MOD:50-55: after ``views_linears``, before ``feature_linear``).

    ``shapes``: key -> shape in state_dict order.  Returns ``(values, (sample_alpha, sample_rgb) | None)``; with
    ``K_samples=None`` the latent draws are skipped (re-initialisation of an existing module)."""
    vals = OrderedDict()

    def lin(prefix):
        out_f, in_f = shapes[prefix + ".weight"]
        l = nn.Linear(in_f, out_f)
        vals[prefix + ".weight"], vals[prefix + ".bias"] = l.weight.detach(), l.bias.detach()

    for i in range(netdepth):
        lin(f"pts_linears.{i}")                                            # MOD:38-39
    lin("views_linears.0")                                                 # MOD:42
    vals["alpha_mean"], vals["alpha_std"] = torch.zeros(1), torch.ones(1)  # MOD:44-45
    vals["rgb_mean"], vals["rgb_std"] = torch.zeros(3), torch.ones(3)      # MOD:47-48
    latents = None
    if K_samples is not None:
        # MOD:50-51: two `intepolation_*` latents come first.  They only feed NeRF_Flows.interpolation, which - like
        # .sample - cannot run in the reference (both call the flows without the required is_test argument), so the
        # draws are made for the RNG stream and dropped.
        torch.empty([2, 1]).normal_()
        torch.empty([2, 3]).normal_()
        latents = (torch.empty([K_samples, 1]).normal_(), torch.empty([K_samples, 3]).normal_())     # MOD:54-55
    for name in ("feature_linear", "alpha_linear", "alpha_std_linear", "h_alpha_linear", "h_rgb_linear"):
        lin(name)                                                          # MOD:58-62
    for flow in ("flows_rgb", "flows_alpha"):                              # MOD:66-67
        for name in ("amor_d", "amor_diag1.0", "amor_diag2.0", "amor_b"):
            lin(f"{flow}.{name}")                                          # MOD:339-350
    missing = set(shapes) - set(vals)
    if missing:
        raise RuntimeError(f"reference_init: no rule for {sorted(missing)}")
    return vals, latents


class NeRF_Flows(nn.Module):
    """Drop-in for the reference's ``NeRF_Flows`` (MOD:13-291).

    All parameters live in ONE flat fp32 tensor ``self.flat`` (the layout of ``state_dict()``),
    which is what the kernels, the gradient all-reduce and the fused Adam work on.
    ``state_dict()`` / ``load_state_dict()`` speak the reference's key names (with the buffers
    ``flip_idx`` / ``triu_mask`` / ``diag_idx``), so released checkpoints load unchanged.
    """

    def __init__(self, args):
        super().__init__()
        self.D = args.netdepth
        self.W = args.netwidth
        self.input_ch = args.input_ch
        self.input_ch_views = args.input_ch_views
        self.K_samples = args.K_samples
        self.skips = args.skips
        self.use_viewdirs = args.use_viewdirs
        self.h_alpha_size = args.h_alpha_size
        self.h_rgb_size = args.h_rgb_size
        args.z_size = 3                                            # MOD:31
        self.z_size = 3
        self.n_flows = args.n_flows
        self.type_flows = getattr(args, "type_flows", "triangular")
        self.n_hidden = getattr(args, "n_hidden", 128)
        dev = torch.device(getattr(args, "device", "cuda"))
        if dev.type != "cuda":
            raise RuntimeError("NeRF_Flows needs a GPU device: the CF-NeRF hot path is HIP-only")
        self.device = dev
        if not self.use_viewdirs:
            # the reference crashes later with an AttributeError (MOD:63-64,183-186); reject up front
            raise ValueError("use_viewdirs=False is not supported (the reference's NeRF_Flows cannot run it either)")
        if list(self.skips) != [self.D / 2]:
            # the float division matters: for an odd netdepth nothing matches `i in skips` and there is no skip concat
            raise ValueError(f"skips must be [netdepth / 2] (RUN:327), got {self.skips}")
        if (self.input_ch - 3) % 6 or (self.input_ch_views - 3) % 6:
            raise ValueError("input_ch / input_ch_views must come from get_embedder (3 + 6*multires)")
        self.cfg = _cfg_struct(self.D, self.W, (self.input_ch - 3) // 6, (self.input_ch_views - 3) // 6,
                               self.h_alpha_size, self.h_rgb_size, self.n_flows)
        self.layout, n_params = param_layout(self.cfg)
        self.n_params = n_params
        self.flat = nn.Parameter(torch.zeros(n_params, dtype=torch.float32, device=dev))
        self.sample_size = args.K_samples
        self._replay_reference_init(draw_latents=True)
        h = C.c_void_p()
        L.check(L.lib().cfnerf_model_create(C.byref(self.cfg), C.byref(h)), "cfnerf_model_create")
        self._h = h
        self._packed_version = None
        self._next_eps = None          # explicit latents handed from render_rays to an unfused network_query_fn
        self._ws = None                # train-step workspace: a torch-owned block lent to the library

    # ---- parameters ------------------------------------------------------------------------
    def view(self, key: str) -> torch.Tensor:
        off, n = self.layout[key]
        return self.flat.data[off:off + n].view(_shape_of(key, n, self.cfg))

    @torch.no_grad()
    def _replay_reference_init(self, draw_latents: bool):
        shapes = OrderedDict((k, tuple(self.view(k).shape)) for k in self.layout)
        vals, latents = reference_init(shapes, self.D, self.sample_size if draw_latents else None)
        for k, v in vals.items():
            self.view(k).copy_(v)
        if draw_latents:
            self.sample_alpha, self.sample_rgb = latents    # eval latents: plain attributes, NOT in state_dict
        self._dirty = True
        self.params_serial += 1             # (writes through .data views do not move flat._version)

    def reset_parameters(self):
        """Fresh nn.Linear default init (U(+-1/sqrt(fan_in)) for weight and bias), base Gaussians mean 0 / std 1; the
        fixed eval latents are kept."""
        self._replay_reference_init(draw_latents=False)

    def _buffers_ref(self):
        return flow_buffers(self.n_flows, self.device)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for key in self.layout:
            v = self.view(key)
            destination[prefix + key] = v if keep_vars else v.detach().clone()
        for k, v in self._buffers_ref().items():
            destination[prefix + k] = v

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        bufs = self._buffers_ref()
        with torch.no_grad():
            for key in self.layout:
                k = prefix + key
                if k in state_dict:
                    src = state_dict[k]
                    dst = self.view(key)
                    if tuple(src.shape) != tuple(dst.shape):
                        error_msgs.append(f"size mismatch for {k}: {tuple(src.shape)} vs {tuple(dst.shape)}")
                        continue
                    dst.copy_(src.to(dst.device, torch.float32))
                elif strict:
                    missing_keys.append(k)
        self._dirty = True                  # re-pack before the next launch
        self.params_serial += 1             # (writes through .data views do not move flat._version): a pending backward must notice
        if strict:
            for k in state_dict:
                if k.startswith(prefix) and k[len(prefix):] not in self.layout and k[len(prefix):] not in bufs:
                    unexpected_keys.append(k)

    def _sync(self):
        """Re-pack the MFMA-ordered weight copies if ``flat`` changed (optimizer.step, load_state_dict)."""
        if (getattr(self, "_dirty", True) or self.flat.data_ptr() != getattr(self, "_packed_ptr", None)
                or self.flat._version != self._packed_version):
            L.check(L.lib().cfnerf_model_set_params(self._h, L.ptr(self.flat.data), L.stream()), "cfnerf_model_set_params")
            self.mark_packed()

    def mark_packed(self):
        """Tell the module that the library already re-packed (cfnerf_adam_step does)."""
        self._packed_version = self.flat._version
        self._packed_ptr = self.flat.data_ptr()
        self._dirty = False
        self.pack_serial += 1               # identifies the packed weights the library holds: a backward must meet its forward's

    pack_serial = 0

    def mark_dirty(self):
        """Call after writing into ``flat.data`` / ``view(key)`` directly."""
        self._dirty = True
        self.params_serial += 1

    # bumped by every parameter update that does not go through a torch in-place op on ``flat`` (the fused Adam kernel of
    # train.Trainer writes through the raw pointer; mark_dirty): with ``flat._version`` it identifies "the weights of a forward"
    params_serial = 0

    @property
    def handle(self):
        return self._h

    def ensure_workspace(self, N: int, S: int, K: int):
        """Lend the library a torch-owned block big enough for a STASH forward + backward of an (N,S,K) batch
        (``cfnerf_workspace_bytes`` / ``cfnerf_model_set_workspace``): the library itself never allocates on the train
        path.  The block only grows; a pending backward of an earlier forward is invalidated when it does."""
        lib = L.lib()
        need = lib.cfnerf_workspace_bytes(C.byref(self.cfg), N, S, K)
        if need < 0:
            raise RuntimeError("cfnerf_workspace_bytes: " + lib.cfnerf_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            if self._ws is not None:
                # growth is rare: wait for every kernel that may still use the old block (whatever stream it ran on) before the
                # block goes back to torch's caching allocator
                torch.cuda.synchronize(self.device)
            L.check(lib.cfnerf_model_set_workspace(self._h, None, 0), "cfnerf_model_set_workspace")   # drop the old block first
            self._ws = None
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            L.check(lib.cfnerf_model_set_workspace(self._h, C.c_void_p(ws.data_ptr()), ws.numel()), "cfnerf_model_set_workspace")
            self._ws = ws

    def release_workspace(self):
        """Give the block back (e.g. before a large evaluation render)."""
        if self._ws is not None:
            torch.cuda.synchronize(self.device)
            L.check(L.lib().cfnerf_model_set_workspace(self._h, None, 0), "cfnerf_model_set_workspace")
            self._ws = None

    def set_precision(self, mode: str):
        """'fp32' (default: exact-fp32 MFMA) or 'bf16x3' (opt-in split-bf16 MFMA in the forward's dense layers)."""
        code = {"fp32": 0, "bf16x3": 1}[mode]
        L.check(L.lib().cfnerf_model_set_precision(self._h, code), "cfnerf_model_set_precision")
        self.precision = mode

    def set_flow_math(self, mode: str):
        """Arithmetic of the flows + composite inside the fused kernels: 'libm' (correctly rounded routines), 'fast' (hardware
        exp2 / log2 / rcp forms, same parity bounds) or 'auto' (default: libm below 16 latent samples, fast from 16 on)."""
        L.check(L.lib().cfnerf_model_set_flow_math(self._h, {"auto": 0, "libm": 1, "fast": 2}[mode]), "cfnerf_model_set_flow_math")
        self.flow_math = mode

    def eval_eps(self):
        """[K,4] eval latents: the fixed buffers with the LAST sample zeroed (MOD:199,205)."""
        # the device copy is rebuilt only when the (plain-attribute) latents were replaced or edited in place
        key = (id(self.sample_rgb), self.sample_rgb._version, id(self.sample_alpha), self.sample_alpha._version)
        if getattr(self, "_eval_eps_key", None) != key:
            e = torch.cat([self.sample_rgb, self.sample_alpha], -1).to(torch.float32).clone()
            e[-1] = 0
            self._eval_eps_dev, self._eval_eps_key = e.to(self.device), key
        return self._eval_eps_dev

    def draw_eps(self):
        """Fresh train latents in the reference's order: eps_alpha then eps_rgb (MOD:234,246), drawn from torch's CPU
        generator like the reference.  The copy to the device goes through pinned memory and does not block the host
        (a pageable copy would stall the launch queue every step)."""
        ea = torch.empty([self.K_samples, 1]).normal_()
        er = torch.empty([self.K_samples, 3]).normal_()
        return torch.cat([er, ea], -1).pin_memory().to(self.device, non_blocking=True)

    # ---- forward (MOD:188-291) ---------------------------------------------------------------
    def forward(self, x, is_val=False, is_test=False, eps_alpha=None, eps_rgb=None):
        _need_gpu(x, "x")
        if x.shape[-1] != self.input_ch + self.input_ch_views:
            raise ValueError(f"x must have {self.input_ch + self.input_ch_views} channels, got {x.shape[-1]}")
        self._sync()
        xf = _f32c(x.reshape(-1, x.shape[-1]))
        P, K = xf.shape[0], self.K_samples
        if eps_alpha is not None or eps_rgb is not None:
            eps = torch.cat([eps_rgb, eps_alpha], -1).to(self.device, torch.float32).contiguous()
        elif self._next_eps is not None:                            # latents chosen by the enclosing render_rays call
            eps = self._next_eps
        else:
            eps = self.eval_eps() if is_test else self.draw_eps()
        if torch.is_grad_enabled() and self.flat.requires_grad and not is_test and P > 0:
            # the reference's forward is an autograd graph (MOD:188-291): so is this one - cfnerf_network_fwd with the
            # activation stash, differentiated by cfnerf_network_bwd (gradients reach the parameters; x is a constant)
            raw, ent = _NetworkFn.apply(self.flat, self, xf, eps)
            return raw, ent.reshape(1, 1, 1).expand(P, K, 1)         # MOD:291
        raw = torch.empty(P, K, 4, device=self.device, dtype=torch.float32)
        ent = torch.zeros(1, device=self.device, dtype=torch.float32)
        flags = 0 if is_test else L.F_TRAIN
        L.check(L.lib().cfnerf_network_fwd(self._h, L.ptr(xf), L.ptr(eps), P, K, flags, L.ptr(raw), L.ptr(ent), L.stream()),
                "cfnerf_network_fwd")
        if is_test:
            return raw, torch.zeros_like(raw)                        # MOD:223
        return raw, ent.reshape(1, 1, 1).expand(P, K, 1)             # MOD:291

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                L.lib().cfnerf_model_destroy(h)      # synchronises the device before the lent workspace tensor is freed
            except Exception:
                pass
            try:
                object.__setattr__(self, "_h", None)     # (nn.Module.__setattr__ may already be torn down at interpreter exit)
            except Exception:
                pass


def _params_token(model):
    """Identifies "the weights of a forward": torch in-place ops on ``flat`` (optimizer.step) move ``flat._version``; everything that
    writes through the raw pointer or a ``.data`` view (the fused Adam kernel, load_state_dict, mark_dirty) bumps ``params_serial``."""
    return (model.flat._version, model.params_serial)


def _refuse_changed_params(model, token, pack_serial):
    """A backward differentiates the stashed activations of ITS forward against what the library holds NOW: the packed weights AND -
    for the base Gaussians (alpha_mean / alpha_std / rgb_mean / rgb_std) - the flat parameter buffer itself, which the flow-adjoint
    kernels read live.  If either changed since the forward (optimizer.step(), Trainer.step() through the same handle,
    load_state_dict(), mark_dirty(); or a re-pack by a later launch) the result would be silently inconsistent whether or not the
    activation stash is still the forward's (round-4 advisor) - torch autograd raises in this situation, so do we."""
    if _params_token(model) != token or model.pack_serial != pack_serial:
        raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: the "
                           "parameters of NeRF_Flows changed between this forward and its backward (call backward() before "
                           "optimizer.step() / Trainer.step() / load_state_dict())")


class _NetworkFn(torch.autograd.Function):
    """NeRF_Flows.forward as an autograd node: cfnerf_network_fwd with the activation stash + cfnerf_network_bwd.  The model has
    ONE stash.  If a later grad-enabled forward replaced it before this node's backward runs (several chunks of one batch through
    a caller's own batchify loop), the node re-runs its forward from the inputs it kept (x, latents - the parameters cannot
    have changed in between) and then differentiates: chunked callers train, at the price of one extra forward per chunk."""

    @staticmethod
    def _forward_stash(model, xf, eps, raw, ent):
        P, K = xf.shape[0], eps.shape[0]
        model.ensure_workspace(1, P, K)
        lib = L.lib()
        L.check(lib.cfnerf_network_fwd(model.handle, L.ptr(xf), L.ptr(eps), P, K, L.F_TRAIN | L.F_STASH, L.ptr(raw), L.ptr(ent), L.stream()),
                "cfnerf_network_fwd")
        return lib.cfnerf_model_stash_generation(model.handle)

    @staticmethod
    def forward(ctx, flat, model, xf, eps):
        P, K = xf.shape[0], eps.shape[0]
        raw = torch.empty(P, K, 4, device=xf.device)
        ent = torch.zeros(1, device=xf.device)
        ctx.generation = _NetworkFn._forward_stash(model, xf, eps, raw, ent)
        ctx.model, ctx.xf, ctx.eps, ctx.n_params = model, xf, eps, flat.numel()
        ctx.params_at = _params_token(model)                       # the weights this graph was taken at
        ctx.pack_serial = model.pack_serial                        # ... and the packed copy of them the library held
        return raw, ent.reshape(())

    @staticmethod
    def backward(ctx, d_raw, d_ent):
        model, lib = ctx.model, L.lib()
        if _params_token(model) != ctx.params_at or getattr(model, "_dirty", False):
            _refuse_changed_params(model, None, None)              # (always raises)
        if lib.cfnerf_model_stash_generation(model.handle) == ctx.generation:
            _refuse_changed_params(model, ctx.params_at, ctx.pack_serial)      # the stash is this forward's: so must the packed weights be
        else:
            # the stash is gone (a later grad-enabled forward replaced it): re-run the forward - the SAME forward, the parameters being the
            # ones it was taken at (checked above; a re-pack of unchanged parameters in between is harmless here)
            model._sync()
            P, K = ctx.xf.shape[0], ctx.eps.shape[0]
            ctx.generation = _NetworkFn._forward_stash(model, ctx.xf, ctx.eps, torch.empty(P, K, 4, device=ctx.xf.device),
                                                       torch.zeros(1, device=ctx.xf.device))
        grad = torch.empty(ctx.n_params, device=model.flat.device)
        dr = _f32c(d_raw) if d_raw is not None else None
        de = _f32c(d_ent.reshape(1)) if d_ent is not None else None
        if dr is None and de is None:
            return torch.zeros(ctx.n_params, device=model.flat.device), None, None, None
        L.check(lib.cfnerf_network_bwd(model.handle, ctx.generation, L.ptr(dr), L.ptr(de), L.ptr(grad), L.stream()), "cfnerf_network_bwd")
        return grad, None, None, None


class _CompositeFn(torch.autograd.Function):
    """raw2outputs (RUN:411-454) as an autograd node: cfnerf_composite_fwd + cfnerf_composite_bwd (stateless; differentiable with
    respect to raw through every output: rgb_map, disp_map, weights, depth_map)."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, white_bkgd):
        N, S, K = raw.shape[0], raw.shape[1], raw.shape[2]
        dev = raw.device
        rgb_map = torch.empty(N, 3, K, device=dev)
        disp_map = torch.empty(N, K, device=dev)
        depth_map = torch.empty(N, K, device=dev)
        weights = torch.empty(N, S, K, device=dev)
        L.check(L.lib().cfnerf_composite_fwd(L.ptr(raw), L.ptr(z_vals), L.ptr(rays_d), N, S, K, int(bool(white_bkgd)), L.ptr(rgb_map),
                                             L.ptr(disp_map), L.ptr(depth_map), L.ptr(weights), L.stream()), "cfnerf_composite_fwd")
        ctx.save_for_backward(raw, z_vals, rays_d)
        ctx.white_bkgd = int(bool(white_bkgd))
        return rgb_map, disp_map, weights, depth_map

    @staticmethod
    def backward(ctx, d_rgb, d_disp, d_weights, d_depth):
        raw, z_vals, rays_d = ctx.saved_tensors
        N, S, K = raw.shape[0], raw.shape[1], raw.shape[2]
        if N == 0 or (d_rgb is None and d_disp is None and d_weights is None and d_depth is None):
            return torch.zeros_like(raw), None, None, None
        d_raw = torch.empty_like(raw)
        g_rgb = _f32c(d_rgb) if d_rgb is not None else torch.zeros(N, 3, K, device=raw.device)
        c = lambda t: _f32c(t) if t is not None else None
        L.check(L.lib().cfnerf_composite_bwd(L.ptr(raw), L.ptr(z_vals), L.ptr(rays_d), N, S, K, ctx.white_bkgd, L.ptr(g_rgb), L.ptr(c(d_disp)),
                                             L.ptr(c(d_depth)), L.ptr(c(d_weights)), L.ptr(d_raw), L.stream()), "cfnerf_composite_bwd")
        return d_raw, None, None, None


class _DataParallelShim(nn.Module):
    """Stands where the reference puts ``nn.DataParallel(model)`` (RUN:330): same ``.module`` attribute and
    the same ``module.``-prefixed ``state_dict`` keys, but no replication - multi-GPU is one process per
    GPU with ray sharding (see cf-nerf_amd/train.py)."""

    def __init__(self, module: NeRF_Flows):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def _unwrap(network_fn) -> NeRF_Flows:
    m = network_fn.module if isinstance(network_fn, _DataParallelShim) else network_fn
    if not isinstance(m, NeRF_Flows):
        raise TypeError("network_fn must be the NeRF_Flows returned by create_nerf")
    return m


# --------------------------------------------------------------------------------------------
def batchify(fn, chunk):
    """RUN:47-64.  The kernels tile internally, so chunking only bounds the size of one launch."""
    if chunk is None:
        return fn

    def ret(inputs, is_val, is_test):
        m = getattr(fn, "module", fn)
        if isinstance(m, NeRF_Flows) and torch.is_grad_enabled() and m.flat.requires_grad and not is_test:
            # one launch = the model's one stash; chunks would each replace it and the backward would re-run their forwards
            return fn(inputs, is_val, is_test)
        A, B = [], []
        for i in range(0, inputs.shape[0], chunk):
            a, b = fn(inputs[i:i + chunk], is_val, is_test)
            A.append(a)
            B.append(b)
        return torch.cat(A, 0), torch.cat(B, 0)
    return ret


def run_network(inputs, viewdirs, fn, is_val, is_test, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """RUN:67-85 (the unfused query: embed with torch, then the network kernel)."""
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs.shape) if viewdirs.ndim == inputs.ndim - 1 else viewdirs
        input_dirs_flat = torch.reshape(input_dirs, [-1, input_dirs.shape[-1]])
        embedded = torch.cat([embedded, embeddirs_fn(input_dirs_flat)], -1)
    outputs_flat, loss_entropy = batchify(fn, netchunk)(embedded, is_val, is_test)
    outputs = torch.reshape(outputs_flat, list(inputs.shape[:-1]) + list(outputs_flat.shape[-2:]))
    return outputs, loss_entropy


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """RUN:411-454 on the standalone composite kernel.  ``raw_noise_std`` is accepted and has no effect,
    exactly like the reference (the noise is generated but never added, RUN:432-442)."""
    _need_gpu(raw, "raw")
    N, S, K = raw.shape[0], raw.shape[1], raw.shape[2]
    if torch.is_grad_enabled() and raw.requires_grad:                # differentiable like the reference's (with respect to raw)
        rc = raw.to(torch.float32).contiguous()
        return _CompositeFn.apply(rc, _f32c(z_vals), _f32c(rays_d), bool(white_bkgd))
    raw_c, z_c, d_c = _f32c(raw), _f32c(z_vals), _f32c(rays_d)
    dev = raw.device
    rgb_map = torch.empty(N, 3, K, device=dev)
    disp_map = torch.empty(N, K, device=dev)
    depth_map = torch.empty(N, K, device=dev)
    weights = torch.empty(N, S, K, device=dev)
    L.check(L.lib().cfnerf_composite_fwd(L.ptr(raw_c), L.ptr(z_c), L.ptr(d_c), N, S, K, int(bool(white_bkgd)),
                                         L.ptr(rgb_map), L.ptr(disp_map), L.ptr(depth_map), L.ptr(weights), L.stream()),
            "cfnerf_composite_fwd")
    return rgb_map, disp_map, weights, depth_map


# --------------------------------------------------------------------------------------------
class _RenderFn(torch.autograd.Function):
    """Fused render (forward with activation stash) + cfnerf_render_bwd.  The model has ONE stash: the node remembers
    the generation of its forward and the backward is refused (loudly) if a later grad-enabled forward replaced it."""

    @staticmethod
    def forward(ctx, flat, model, rays, t_vals, t_rand, eps, flags, want_pts, z_vals):
        N, K = rays.shape[0], eps.shape[0]
        S = z_vals.shape[1] if z_vals is not None else t_vals.shape[0]
        dev = rays.device
        rgb_map = torch.empty(N, 3, K, device=dev)
        disp = torch.empty(N, K, device=dev)
        depth = torch.empty(N, K, device=dev)
        raw = torch.empty(N, S, K, 4, device=dev)
        pts = torch.empty(N, S, 3, device=dev) if want_pts else None
        ent = torch.zeros(1, device=dev)
        lib = L.lib()
        model.ensure_workspace(N, S, K)
        L.check(lib.cfnerf_render_fwd(model.handle, L.ptr(rays), L.ptr(t_vals), L.ptr(t_rand), L.ptr(z_vals), L.ptr(eps), N, S, K,
                                      flags | L.F_STASH, L.ptr(rgb_map), L.ptr(disp), L.ptr(depth), L.ptr(raw), None,
                                      L.ptr(pts), None, L.ptr(ent), L.stream()), "cfnerf_render_fwd")
        ctx.model = model
        ctx.n_params = flat.numel()
        ctx.generation = lib.cfnerf_model_stash_generation(model.handle)
        ctx.params_at, ctx.pack_serial = _params_token(model), model.pack_serial
        ctx.shape = (N, 3, K)
        ctx.mark_non_differentiable(disp, raw)
        if pts is None:
            pts = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(pts)
        return rgb_map, disp, depth, ent.reshape(()), raw, pts

    @staticmethod
    def backward(ctx, d_rgb, d_disp, d_depth, d_ent, d_raw, d_pts):
        model = ctx.model
        _refuse_changed_params(model, ctx.params_at, ctx.pack_serial)      # (a replaced STASH is refused by the library itself: generation id)
        dev = model.flat.device
        grad = torch.empty(ctx.n_params, device=dev)
        d_rgb = _f32c(d_rgb) if d_rgb is not None else torch.zeros(ctx.shape, device=dev)
        dd = _f32c(d_depth) if d_depth is not None else None
        de = _f32c(d_ent.reshape(1)) if d_ent is not None else None
        L.check(L.lib().cfnerf_render_bwd(model.handle, ctx.generation, L.ptr(d_rgb), L.ptr(dd), L.ptr(de), L.ptr(grad), L.stream()),
                "cfnerf_render_bwd")
        return grad, None, None, None, None, None, None, None, None


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, is_train, uniformsample, retraw=False,
                lindisp=False, K_samples=0, perturb=0., N_importance=0, network_fine=None, white_bkgd=False,
                raw_noise_std=0., verbose=False, pytest=False, t_rand=None, eps_alpha=None, eps_rgb=None,
                t_vals=None, retweights=False, hierarchical_extension=False, u_fine=None):
    """Volumetric rendering of a ray batch (RUN:457-553) in ONE fused launch.

    Returns ``{'rgb_map' [N,3,K], 'disp_map' [N,K], 'depth_map' [N,K]}`` plus ``raw``, ``loss_entropy``
    and ``pts`` when ``is_train`` (RUN:542-547).  ``retraw``, ``uniformsample``, ``K_samples``, ``verbose``
    are accepted and never read, like the reference; ``N_importance > 0`` / ``network_fine`` - which the
    reference silently ignores (there is no fine pass) - are rejected.
    """
    _need_gpu(ray_batch, "ray_batch")
    if hierarchical_extension and N_importance and N_importance > 0:
        return _render_rays_hierarchical(ray_batch, network_fn, N_samples, N_importance, is_train, lindisp, perturb, white_bkgd,
                                         t_rand, eps_alpha, eps_rgb, t_vals, u_fine)
    if N_importance and N_importance > 0 or network_fine is not None:
        raise NotImplementedError("the reference has no hierarchical/fine pass (N_importance is dead there); refusing to ignore it "
                                  "(pass hierarchical_extension=True for the coarse+fine EXTENSION of this build)")
    if ray_batch.shape[-1] != 11:
        raise ValueError("ray_batch must be [N,11] = o3,d3,near,far,viewdir3 (use_viewdirs=True)")
    model = _unwrap(network_fn)
    dev = ray_batch.device
    if t_vals is None:
        t_vals = t_vals_table(dev)
        if N_samples != t_vals.shape[0]:
            # the reference's z_vals.expand([N_rays, N_samples]) raises for any other value (RUN:510,516)
            raise ValueError(f"N_samples must be {t_vals.shape[0]} (hard-coded sample table, RUN:510); pass t_vals= to override")
    else:
        t_vals = t_vals.to(dev, torch.float32).contiguous()
    N, S, K = ray_batch.shape[0], t_vals.shape[0], model.K_samples
    rays = _f32c(ray_batch)
    # randomness, in the reference's order: t_rand (RUN:524), eps_alpha (MOD:234), eps_rgb (MOD:246)
    if perturb > 0.:
        if t_rand is None:
            t_rand = torch.rand([N, S]).to(dev) if not pytest else torch.tensor(__import__("numpy").random.rand(N, S), dtype=torch.float32).to(dev)
        t_rand = _f32c(t_rand.to(dev))
    else:
        t_rand = None
    if eps_alpha is not None or eps_rgb is not None:
        eps = torch.cat([eps_rgb, eps_alpha], -1).to(dev, torch.float32).contiguous()
    else:
        eps = model.draw_eps() if is_train else model.eval_eps()
    flags = (L.F_LINDISP if lindisp else 0) | (L.F_WHITE_BKGD if white_bkgd else 0) | (L.F_TRAIN if is_train else 0)
    model._sync()

    fused = network_query_fn is None or getattr(network_query_fn, "_cfnerf_fused", False)
    if not fused:
        return _render_rays_unfused(rays, model, network_fn, network_query_fn, t_vals, t_rand, eps, is_train, lindisp, white_bkgd)

    if is_train and N > 0 and torch.is_grad_enabled() and model.flat.requires_grad:
        rgb_map, disp, depth, ent, raw, pts = _RenderFn.apply(model.flat, model, rays, t_vals, t_rand, eps, flags, True, None)
        return {'rgb_map': rgb_map, 'disp_map': disp, 'depth_map': depth, 'raw': raw,
                'loss_entropy': ent.reshape(1, 1, 1).expand(N * S, K, 1), 'pts': pts}

    rgb_map = torch.empty(N, 3, K, device=dev)
    disp = torch.empty(N, K, device=dev)
    depth = torch.empty(N, K, device=dev)
    raw = torch.empty(N, S, K, 4, device=dev) if is_train else None
    pts = torch.empty(N, S, 3, device=dev) if is_train else None
    wts = torch.empty(N, S, K, device=dev) if retweights else None
    ent = torch.zeros(1, device=dev)
    L.check(L.lib().cfnerf_render_fwd(model.handle, L.ptr(rays), L.ptr(t_vals), L.ptr(t_rand), None, L.ptr(eps), N, S, K, flags,
                                      L.ptr(rgb_map), L.ptr(disp), L.ptr(depth), L.ptr(raw), L.ptr(wts), L.ptr(pts), None, L.ptr(ent),
                                      L.stream()), "cfnerf_render_fwd")
    ret = {'rgb_map': rgb_map, 'disp_map': disp, 'depth_map': depth}
    if is_train:
        ret['raw'] = raw
        ret['loss_entropy'] = ent.reshape(1, 1, 1).expand(N * S, K, 1)
        ret['pts'] = pts
    if retweights:
        ret['weights'] = wts
    return ret


def _render_rays_hierarchical(ray_batch, network_fn, N_samples, N_importance, is_train, lindisp, perturb, white_bkgd, t_rand,
                              eps_alpha, eps_rgb, t_vals, u_fine):
    """EXTENSION, not in the reference: classic coarse+fine sampling through the single CF-NeRF network.
    Coarse pass on ``t_vals`` (default ``linspace(0,1,N_samples)``), ``cfnerf_sample_pdf`` on the K-mean coarse weights,
    fine pass on the merged depths.  Under autograd the FINE pass is differentiated (explicit-depth STASH launch +
    ``cfnerf_render_bwd``); the resampled depths are constants, as nerf-pytorch detaches ``z_samples``, and the coarse
    pass only provides the sampling distribution (one network, one stash: its outputs ``rgb0`` / ``disp0`` / ``depth0``
    are returned detached - ``train.Trainer.step_hierarchical`` adds a coarse loss term with a second stashed pass).
    Parity is pinned against the build's own CPU restatement of nerf-pytorch's ``sample_pdf`` only - the reference has
    nothing to compare with."""
    model = _unwrap(network_fn)
    dev = ray_batch.device
    rays = _f32c(ray_batch)
    N, K = rays.shape[0], model.K_samples
    tv = torch.linspace(0., 1., steps=N_samples).to(dev) if t_vals is None else t_vals.to(dev, torch.float32).contiguous()
    S = tv.shape[0]
    tr = None
    if perturb > 0.:
        tr = _f32c((torch.rand([N, S]) if t_rand is None else t_rand).to(dev))
    if eps_alpha is not None or eps_rgb is not None:
        eps = torch.cat([eps_rgb, eps_alpha], -1).to(dev, torch.float32).contiguous()
    else:
        eps = model.draw_eps() if is_train else model.eval_eps()
    if u_fine is None:      # det=(perturb == 0.) in nerf-pytorch
        u_fine = torch.linspace(0., 1., steps=N_importance).expand(N, N_importance) if not perturb > 0. else torch.rand(N, N_importance)
    u = _f32c(u_fine.to(dev))
    flags = (L.F_LINDISP if lindisp else 0) | (L.F_WHITE_BKGD if white_bkgd else 0) | (L.F_TRAIN if is_train else 0)
    model._sync()
    lib = L.lib()

    def launch(S_, z_in, want_w):
        rgb, disp, depth = torch.empty(N, 3, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, K, device=dev)
        wts = torch.empty(N, S_, K, device=dev) if want_w else None
        ent = torch.zeros(1, device=dev)
        L.check(lib.cfnerf_render_fwd(model.handle, L.ptr(rays), L.ptr(tv), L.ptr(tr) if z_in is None else None, L.ptr(z_in), L.ptr(eps),
                                      N, S_, K, flags, L.ptr(rgb), L.ptr(disp), L.ptr(depth), None, L.ptr(wts), None, None, L.ptr(ent),
                                      L.stream()), "cfnerf_render_fwd")
        return rgb, disp, depth, wts, ent
    with torch.no_grad():
        rgb0, disp0, depth0, w0, _ = launch(S, None, True)
        z_all = torch.empty(N, S + N_importance, device=dev)
        L.check(lib.cfnerf_sample_pdf(L.ptr(rays), L.ptr(tv), L.ptr(tr), flags, L.ptr(w0), L.ptr(u), N, S, K, N_importance, L.ptr(z_all),
                                      L.stream()), "cfnerf_sample_pdf")
    if is_train and N > 0 and torch.is_grad_enabled() and model.flat.requires_grad:
        rgb, disp, depth, ent, _raw, _pts = _RenderFn.apply(model.flat, model, rays, tv, None, eps, flags, False, z_all)
        ent = ent.reshape(1)
    else:
        rgb, disp, depth, _, ent = launch(S + N_importance, z_all, False)
    ret = {'rgb_map': rgb, 'disp_map': disp, 'depth_map': depth, 'rgb0': rgb0, 'disp0': disp0, 'depth0': depth0, 'z_vals': z_all}
    if is_train:
        ret['loss_entropy'] = ent.reshape(1, 1, 1).expand(N * (S + N_importance), K, 1)
    return ret


def _render_rays_unfused(rays, model, network_fn, network_query_fn, t_vals, t_rand, eps, is_train, lindisp, white_bkgd):
    """A caller-supplied network_query_fn: sample with torch, query through it, composite kernel."""
    rays_d, viewdirs = rays[:, 3:6], rays[:, 8:11]
    N, S = rays.shape[0], t_vals.shape[0]
    z_vals = torch.empty(N, S, device=rays.device)
    pts0 = torch.empty(N, S, 3, device=rays.device)
    L.check(L.lib().cfnerf_sample_points(L.ptr(rays), L.ptr(t_vals), L.ptr(t_rand), L.F_LINDISP if lindisp else 0, N, S, L.ptr(z_vals),
                                         L.ptr(pts0), L.stream()), "cfnerf_sample_points")            # RUN:510-534
    model._next_eps = eps                    # NeRF_Flows.forward consumes it: a custom query fn honours the explicit latents too
    try:
        raw, loss_entropy = network_query_fn(pts0, viewdirs, network_fn, is_val=False, is_test=not is_train)
    finally:
        model._next_eps = None               # never leaks into a later direct call
    rgb_map, disp_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, 0., white_bkgd)
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'depth_map': depth_map}
    if is_train:
        ret.update(raw=raw, loss_entropy=loss_entropy, pts=pts0)
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """RUN:88-100."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (torch.cat(v, 0) if len(v) > 1 else v[0]) for k, v in all_ret.items()}


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """Render rays (RUN:103-170).  Returns ``[rgb_map [..,3,K], disp_map [..,K], depth_map [..,K], extras]``.

    ``chunk`` does not affect results (RUN:112-113); here it does not even split the launch: the fused
    kernel tiles rays internally and keeps nothing per-point in HBM, so one launch renders the batch.
    """
    if not use_viewdirs:
        raise ValueError("use_viewdirs=False is not supported (the reference's model cannot run it, SURVEY R8)")
    lib = L.lib()
    if c2w is not None:
        dev = kwargs["network_fn"].module.device if hasattr(kwargs.get("network_fn"), "module") else _unwrap(kwargs["network_fn"]).device
        sh = (H, W, 3)
        N = H * W
        packed = torch.empty(N, 11, device=dev)
        arr, _keep = _pose_arg(c2w)              # a host pose goes into the kernel arguments as it is: no device round trip
        if c2w_staticcam is None:
            L.check(lib.cfnerf_rays_setup(H, W, float(focal), arr, None, None, N, 0, int(bool(ndc)), float(near), float(far),
                                          L.ptr(packed), L.stream()), "cfnerf_rays_setup")
        else:   # RUN:139-141: view directions from c2w, geometry from the static camera
            L.check(lib.cfnerf_rays_setup(H, W, float(focal), arr, None, None, N, 0, 0, float(near), float(far), L.ptr(packed),
                                          L.stream()), "cfnerf_rays_setup")
            vd = packed[:, 8:11].clone()
            arr2, _keep2 = _pose_arg(c2w_staticcam)
            L.check(lib.cfnerf_rays_setup(H, W, float(focal), arr2, None, None, N, 0, int(bool(ndc)), float(near), float(far),
                                          L.ptr(packed), L.stream()), "cfnerf_rays_setup")
            packed[:, 8:11] = vd
    else:
        rays_o, rays_d = rays
        _need_gpu(rays_d, "rays")
        sh = tuple(rays_d.shape)
        ro, rd = _f32c(rays_o.reshape(-1, 3)), _f32c(rays_d.reshape(-1, 3))
        N = rd.shape[0]
        packed = torch.empty(N, 11, device=rd.device)
        nf, ff = (near if not torch.is_tensor(near) else 0.), (far if not torch.is_tensor(far) else 1.)
        L.check(lib.cfnerf_rays_setup(H, W, float(focal), None, L.ptr(ro), L.ptr(rd), N, 0, int(bool(ndc)), float(nf), float(ff),
                                      L.ptr(packed), L.stream()), "cfnerf_rays_setup")
        if torch.is_tensor(near):
            packed[:, 6] = near.reshape(-1).to(packed)
        if torch.is_tensor(far):
            packed[:, 7] = far.reshape(-1).to(packed)

    all_ret = render_rays(packed, **kwargs)                                   # one launch (see docstring)
    for k in all_ret:
        if k != 'loss_entropy' and k != 'loss_entropy_uniformsample':         # RUN:163
            all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'depth_map']
    return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]


# --------------------------------------------------------------------------------------------
def default_args(**over):
    """argparse.Namespace with the reference's defaults for every flag create_nerf() reads (config_parser, RUN:556-719;
    ``use_viewdirs`` on, as every shipped config sets it).  Keyword arguments override."""
    import argparse
    a = argparse.Namespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=0, netdepth=8, netwidth=256, K_samples=4,
        h_alpha_size=32, h_rgb_size=64, z_size=4, n_flows=4, type_flows="triangular", n_hidden=128, netchunk_per_gpu=1024 * 64,
        n_gpus=1, lrate=5e-4, lrate_decay=250, ft_path=None, basedir="./logs/", dataname="leaves", expname="cfnerf", no_reload=True,
        index_step=-1, is_train=True, uniformsample=False, perturb=1.0, N_samples=128, white_bkgd=False, raw_noise_std=0.0,
        dataset_type="llff", no_ndc=False, lindisp=False, beta1=0.0, device=torch.device("cuda"))
    for k, v in over.items():
        setattr(a, k, v)
    return a


def save_checkpoint(path, global_step, network_fn, optimizer=None, trainer=None):
    """The reference's checkpoint dict (RUN:1085-1100): ``global_step``, ``network_fn_state_dict`` with the
    DataParallel ``module.`` key prefix, ``optimizer_state_dict``.  Like the reference's loader (RUN:360, commented
    out) create_nerf never restores the optimiser state; for a fused ``Trainer`` the flat Adam moments are saved."""
    sd = network_fn.state_dict()
    if not any(k.startswith("module.") for k in sd):
        sd = OrderedDict(("module." + k, v) for k, v in sd.items())
    if trainer is not None:
        opt = {"exp_avg": trainer.exp_avg.detach().cpu(), "exp_avg_sq": trainer.exp_avg_sq.detach().cpu(), "t": trainer.t}
    else:
        opt = optimizer.state_dict() if optimizer is not None else {}
    torch.save({'global_step': global_step, 'network_fn_state_dict': {k: v.detach().cpu() for k, v in sd.items()},
                'optimizer_state_dict': opt}, path)
    return path


def _checkpoint_to_reload(args):
    """The checkpoint create_nerf() resumes from, by the reference's rules (RUN:345-357): an explicit ``ft_path`` wins;
    otherwise the log directory ``basedir/dataname/type_flows/expname`` is searched for ``*tar*`` files and either the
    newest one (``index_step == -1``) or ``<index_step>_01.tar`` is taken.  ``no_reload`` disables all of it."""
    if getattr(args, "no_reload", False):
        return None
    ft = getattr(args, "ft_path", None)
    logdir = os.path.join(args.basedir, args.dataname, args.type_flows, args.expname)
    found = [ft] if ft is not None and ft != 'None' else (
        [os.path.join(logdir, f) for f in sorted(os.listdir(logdir)) if 'tar' in f] if os.path.isdir(logdir) else [])
    if not found:
        return None
    return found[-1] if args.index_step == -1 else os.path.join(logdir, '{:06d}_{:02d}.tar'.format(args.index_step, 1))


def _maybe_reload(args, model) -> int:
    """Load network weights from a reference-format checkpoint (RUN:358-378): keys the model does not have are dropped,
    the optimiser state is NOT restored (that line is commented out in the reference).  Returns the global step."""
    path = _checkpoint_to_reload(args)
    if path is None:
        print('No reloading')
        return 0
    print('Reloading from', path)
    ckpt = torch.load(path, map_location="cpu")
    own = model.state_dict()
    own.update({k: v for k, v in ckpt['network_fn_state_dict'].items() if k in own})
    model.load_state_dict(own)
    return ckpt['global_step']


def create_nerf(args):
    """Instantiate the CF-NeRF model (RUN:317-409).  Returns
    ``(render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer)``."""
    args.embed_fn, args.input_ch = get_embedder(args.multires, args.i_embed)
    args.input_ch_views = 0
    args.embeddirs_fn = None
    if args.use_viewdirs:
        args.embeddirs_fn, args.input_ch_views = get_embedder(args.multires_views, args.i_embed)
    args.output_ch = 5 if args.N_importance > 0 else 4
    args.skips = [args.netdepth / 2]                                          # RUN:327
    if not hasattr(args, "device") or args.device is None:
        args.device = torch.device("cuda")
    model = _DataParallelShim(NeRF_Flows(args))
    grad_vars = list(model.parameters())

    def network_query_fn(inputs, viewdirs, network_fn, is_val, is_test):
        return run_network(inputs, viewdirs, network_fn, is_val, is_test, embed_fn=args.embed_fn,
                           embeddirs_fn=args.embeddirs_fn, netchunk=args.netchunk_per_gpu * max(1, getattr(args, "n_gpus", 1)))
    network_query_fn._cfnerf_fused = True        # render_rays may replace query + composite by the fused launch

    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = _maybe_reload(args, model)

    render_kwargs_train = {
        'is_train': args.is_train, 'uniformsample': args.uniformsample, 'network_query_fn': network_query_fn,
        'perturb': args.perturb, 'N_importance': args.N_importance, 'N_samples': args.N_samples,
        'K_samples': args.K_samples, 'network_fn': model, 'use_viewdirs': args.use_viewdirs,
        'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:                            # RUN:397-400
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    render_kwargs_test['is_train'] = False
    render_kwargs_test['uniformsample'] = False
    render_kwargs_test['retraw'] = True
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer
