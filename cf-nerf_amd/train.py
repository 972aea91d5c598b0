"""Train step of the CF-NeRF hot path on the HIP kernels (reference: train() inner loop, RUN:1013-1077).

One process per GPU.  A step is: ray set-up -> fused forward (activations stashed) -> KDE-NLL loss
(+ beta1 * entropy) -> backward -> ONE all-reduce (sum) of the flat gradient over RCCL when
world_size > 1 -> fused Adam on the flat buffers -> re-pack.  Rays are sharded by the caller
(rank r renders its own N_rand rays); every rank holds the full weights and applies the same update.

Gradient normalisation across ranks: the reference's losses are means over rays / points
(RUN:1042,1045).  Each rank computes the gradient of  nll_local_sum / (3 * N_total) + (beta1 / world) *
entropy_local, so the SUM over ranks is the gradient of the global means for equal shards.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .api import NeRF_Flows, _f32c, _unwrap


def backward_available() -> bool:
    """True when libcfnerf_hip.so carries the real backward (an empty batch is accepted)."""
    rc = L.lib().cfnerf_loss_fwd_bwd(None, None, None, 0, 2, C.c_float(0.0), 0, None, None, None)
    return rc == 0


def shard_bounds(n_rays: int, rank: int, world: int):
    """Contiguous equal shards (the reference's DataParallel splits the same way along dim 0)."""
    if n_rays % world:
        raise ValueError(f"N_rand ({n_rays}) must be divisible by world_size ({world})")
    per = n_rays // world
    return rank * per, (rank + 1) * per


def lr_at(lrate: float, lrate_decay: int, start: int, t: int) -> float:
    """Learning rate used by the t-th optimiser step of this run (RUN:1073-1077: the decayed rate is
    written to the param groups AFTER optimizer.step(), computed from the current global_step)."""
    if t == 0:
        return lrate
    return lrate * (0.1 ** ((start + t - 1) / (lrate_decay * 1000)))


def allreduce_sum_(grad: torch.Tensor, world: int, group=None, force: bool = False):
    """ONE sum all-reduce of the flat buffer (RCCL over xGMI when the group's backend is "nccl").  A gloo group (CPU
    tests, two test processes sharing one GPU) is served through a host staging copy."""
    if world > 1 or force:
        import torch.distributed as dist
        if grad.is_cuda and dist.get_backend(group) == "gloo":
            host = grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            grad.copy_(host)
        else:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return grad


MAX_K = 128     # kMaxK of the kernels


class Trainer:
    """Fused train step on one rank.  Multi-GPU: construct with ``world_size`` (and ``group``) after
    ``torch.distributed.init_process_group``; every rank renders its own shard of the step's rays.

    Latent samples: the reference draws eps once per forward call (MOD:234,246); here every rank must use the SAME eps in
    a step.  That is enforced, not assumed: rank 0 draws the latents of step t+1 and they travel in a 4*K-float tail of
    the step-t gradient all-reduce (the other ranks contribute zeros there), so no extra collective and no reliance on
    identical seeding; step 0 uses one broadcast.  An explicit ``eps=`` argument overrides this (tests, benchmarks)."""

    def __init__(self, net, lrate=5e-4, lrate_decay=250, beta1=0.0, world_size=1, group=None, start=0, force_allreduce=False,
                 overlap_comm=False, time_comm=False, max_rays_per_launch=None):
        # max_rays_per_launch: a step's shard larger than this is walked in EQUAL slices (forward -> loss -> backward per slice, the
        # gradient accumulated by cfnerf_render_bwd_accumulate, ONE exchange and ONE Adam step at the end): the train-step workspace
        # is sized for a slice (3 MiB per ray at W = 256), not for the batch - the reference trains any N_rand (RUN:88-100,602)
        self.max_rays = None if not max_rays_per_launch else int(max_rays_per_launch)
        self.force_allreduce = bool(force_allreduce)
        self.overlap_comm = bool(overlap_comm)
        # time_comm: two events on the compute stream around every gradient exchange - what the exchange EXPOSES on that stream
        # (an exchange that ran entirely under compute would read ~0); read back with comm_stats()
        self._comm_ev = [] if time_comm else None
        self.net: NeRF_Flows = _unwrap(net)
        dev = self.net.flat.device
        self.lrate, self.lrate_decay, self.beta1 = float(lrate), int(lrate_decay), float(beta1)
        self.world, self.group, self.start = int(world_size), group, int(start)
        self.rank = 0
        import torch.distributed as dist
        if (self.world > 1 or self.force_allreduce) and dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)       # (world_size > 1 without a process group: shard semantics only, see forward_backward)
        n = self.net.n_params
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.gbuf = torch.zeros(n + 4 * MAX_K, device=dev)      # flat gradient | next step's latents (rank 0's, via the all-reduce)
        self.grad = self.gbuf[:n]
        self._eps_next = None
        self.d_ent = torch.tensor([self.beta1 / self.world], device=dev)
        # loss, loss_nll, mse, psnr of the local shard; loss and loss_nll are CONTRIBUTIONS (nll / (3 N_total) and
        # beta1 / world on the shard's entropy): their sum over ranks is the global value (RUN:1042-1050)
        self.scalars = torch.zeros(4, device=dev)
        self.entropy = torch.zeros(1, device=dev)
        self.t = 0
        self._buf_n = None

    # ---- gradient exchange ---------------------------------------------------------------------------------------
    def _exchange_plan(self):
        """Index tensors of the two buckets: `early` = flat ranges that are final before the backward's last launch
        (cfnerf_grad_early_ranges), `late` = the rest + the latents tail.  Built once, after the first backward."""
        if getattr(self, "_xplan", None) is None:
            lib, n = L.lib(), self.net.n_params
            offs, cnts = (C.c_int64 * 64)(), (C.c_int64 * 64)()
            k = lib.cfnerf_grad_early_ranges(self.net.handle, offs, cnts, 64)
            if k < 0:
                raise RuntimeError("cfnerf_grad_early_ranges: " + lib.cfnerf_last_error().decode())
            dev = self.gbuf.device
            mask = torch.zeros(self.gbuf.numel(), dtype=torch.bool, device=dev)
            for i in range(k):
                mask[offs[i]:offs[i] + cnts[i]] = True
            idx = torch.arange(self.gbuf.numel(), device=dev)
            self._xplan = (idx[mask], idx[~mask])
            self._comm = torch.cuda.Stream(device=dev)
        return self._xplan

    def _exchange(self):
        """Sum the gradient (and the latents tail) over the ranks.  Default: ONE all-reduce of the whole flat buffer
        (2.47 MB at W = 256).  `overlap_comm=True` (RCCL groups): two buckets - the early one (every bias, the base
        Gaussians and the weights fed by the big dW launch alone: ~3/4 of the bytes) is gathered and all-reduced on a side
        stream as soon as those tensors are final, i.e. UNDER the small-job launch that ends the backward (~0.23 ms);
        the late one follows on the main stream.  On ONE GPU, where the exchange itself costs nothing, the gathers,
        scatters and the second launch of the two-bucket form cost ~0.13 ms per step, more than a 2.5 MB all-reduce is
        expected to expose on 8 GPUs - hence opt-in until it can be measured on a multi-GPU node."""
        import torch.distributed as dist
        if not self.overlap_comm or not self.gbuf.is_cuda or dist.get_backend(self.group) != "nccl":
            allreduce_sum_(self.gbuf, self.world, self.group, self.force_allreduce)
            return
        early, late = self._exchange_plan()
        main = torch.cuda.current_stream(self.gbuf.device)
        L.check(L.lib().cfnerf_stream_wait_grad_early(self.net.handle, C.c_void_p(self._comm.cuda_stream)), "cfnerf_stream_wait_grad_early")
        with torch.cuda.stream(self._comm):
            e_buf = self.gbuf.index_select(0, early)
            w_early = dist.all_reduce(e_buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        l_buf = self.gbuf.index_select(0, late)
        dist.all_reduce(l_buf, op=dist.ReduceOp.SUM, group=self.group)
        self.gbuf.index_copy_(0, late, l_buf)
        w_early.wait()                              # the main stream waits for the side stream's all-reduce
        main.wait_stream(self._comm)
        self.gbuf.index_copy_(0, early, e_buf)
        e_buf.record_stream(main)

    def _timed_exchange(self, fn):
        if self._comm_ev is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self._comm_ev.append((e0, e1))
        if len(self._comm_ev) > 4096:
            del self._comm_ev[:2048]

    def comm_stats(self, last_n=None):
        """Exposed time of the gradient exchange over the last ``last_n`` steps (device-synchronising), its payload and what
        torch.distributed reports about the group - the self-diagnosis block of a multi-GPU bench line."""
        import torch.distributed as dist
        out = {"payload_bytes": int(self.gbuf.numel() * 4), "form": "two buckets (overlap_comm)" if self.overlap_comm else "one all-reduce",
               "backend": None, "world_size_reported": None}
        if dist.is_available() and dist.is_initialized():
            out["backend"] = dist.get_backend(self.group)
            out["world_size_reported"] = dist.get_world_size(self.group)
        ev = (self._comm_ev or [])[-(last_n or 0):]
        if ev:
            ev[-1][1].synchronize()
            ms = [a.elapsed_time(b) for a, b in ev]
            out.update(exposed_ms_mean=sum(ms) / len(ms), exposed_ms_max=max(ms), exposed_ms_min=min(ms), steps=len(ms))
        return out

    def _step_eps(self):
        """Latents of the coming step, identical on every rank."""
        net = self.net
        if self.world == 1 and not self.force_allreduce:
            return net.draw_eps()
        if self._eps_next is None:                              # first step: one broadcast from rank 0
            import torch.distributed as dist
            eps = net.draw_eps() if self.rank == 0 else torch.zeros(net.K_samples, 4, device=net.flat.device)
            if eps.is_cuda and dist.get_backend(self.group) == "gloo":
                host = eps.cpu()
                dist.broadcast(host, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                eps = host.to(eps.device)
            else:
                dist.broadcast(eps, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            return eps
        return self._eps_next

    def _queue_next_eps(self):
        """Before the all-reduce: rank 0 writes the NEXT step's latents behind the gradient, everyone else zeros."""
        K, n = self.net.K_samples, self.net.n_params
        tail = self.gbuf[n:n + 4 * K]
        if self.rank == 0:
            tail.copy_(self.net.draw_eps().reshape(-1))      # pinned, non-blocking: the host keeps running ahead
        else:
            tail.zero_()

    def _take_next_eps(self):
        K, n = self.net.K_samples, self.net.n_params
        self._eps_next = self.gbuf[n:n + 4 * K].reshape(K, 4).clone()

    def _buffers(self, N, K):
        if self._buf_n != (N, K):
            dev = self.net.flat.device
            self.packed = torch.empty(N, 11, device=dev)
            self.rgb_map = torch.empty(N, 3, K, device=dev)
            self.disp = torch.empty(N, K, device=dev)
            self.depth = torch.empty(N, K, device=dev)
            self.d_rgb = torch.empty(N, 3, K, device=dev)
            self._buf_n = (N, K)

    def forward_backward(self, H, W, focal, rays, target, t_rand=None, eps=None, near=0., far=1., ndc=True,
                         lindisp=False, white_bkgd=False, perturb=1., t_vals=None, **_ignored):
        """Forward + loss + backward of this rank's shard.  Leaves the (un-reduced) gradient in ``self.grad``."""
        net, lib = self.net, L.lib()
        dev = net.flat.device
        rays_o, rays_d = rays
        ro, rd = _f32c(rays_o.reshape(-1, 3)), _f32c(rays_d.reshape(-1, 3))
        N, K = rd.shape[0], net.K_samples
        if t_vals is None:
            if not hasattr(self, "_tv"):
                from .api import t_vals_table
                self._tv = t_vals_table(dev)
            t_vals = self._tv
        S = t_vals.shape[0]
        self._buffers(N, K)
        st = L.stream()
        L.check(lib.cfnerf_rays_setup(H, W, float(focal), None, L.ptr(ro), L.ptr(rd), N, 0, int(bool(ndc)), float(near), float(far),
                                      L.ptr(self.packed), st), "cfnerf_rays_setup")
        if perturb > 0. and t_rand is None:
            t_rand = torch.rand(N, S, device=dev)
        if perturb <= 0.:
            t_rand = None
        if eps is None:
            eps = net.draw_eps()
        eps = _f32c(eps)
        net._sync()
        flags = L.F_STASH | L.F_TRAIN | (L.F_LINDISP if lindisp else 0) | (L.F_WHITE_BKGD if white_bkgd else 0)
        n_sl = self.n_slices(N)
        t_rand = _f32c(t_rand) if t_rand is not None else None
        target = _f32c(target)
        if n_sl == 1:
            net.ensure_workspace(N, S, K)
            L.check(lib.cfnerf_render_fwd(net.handle, L.ptr(self.packed), L.ptr(t_vals), L.ptr(t_rand),
                                          None, L.ptr(eps), N, S, K, flags, L.ptr(self.rgb_map), L.ptr(self.disp), L.ptr(self.depth),
                                          None, None, None, None, L.ptr(self.entropy), st), "cfnerf_render_fwd")
            L.check(lib.cfnerf_loss_fwd_bwd(L.ptr(self.rgb_map), L.ptr(target), L.ptr(self.entropy), N, K,
                                            C.c_float(self.beta1 / self.world), N * self.world, L.ptr(self.d_rgb), L.ptr(self.scalars), st),
                    "cfnerf_loss_fwd_bwd")
            gen = lib.cfnerf_model_stash_generation(net.handle)
            L.check(lib.cfnerf_render_bwd(net.handle, gen, L.ptr(self.d_rgb), None, L.ptr(self.d_ent) if self.beta1 else None,
                                          L.ptr(self.grad), st), "cfnerf_render_bwd")
            return self.grad
        # ---- the shard in n_sl equal slices: every loss term is taken with n_total = the FULL batch and beta1 / (world n_sl) on the slice's
        # entropy (equal slices: the mean of the slice means is the batch mean), so the slice gradients and the scalar contributions ADD
        Ns = N // n_sl
        net.ensure_workspace(Ns, S, K)
        if getattr(self, "_sl_n", None) != n_sl:
            dev = net.flat.device
            self._d_ent_sl = torch.tensor([self.beta1 / (self.world * n_sl)], device=dev)
            self._sc_sl, self._ent_sl = torch.zeros(4, device=dev), torch.zeros(1, device=dev)
            self._sl_n = n_sl
        beta_sl = C.c_float(self.beta1 / (self.world * n_sl))
        sc_sum, ent_sum = torch.zeros(4, device=net.flat.device), torch.zeros(1, device=net.flat.device)
        for i in range(n_sl):
            a, b = i * Ns, (i + 1) * Ns
            L.check(lib.cfnerf_render_fwd(net.handle, L.ptr(self.packed[a:b]), L.ptr(t_vals), L.ptr(t_rand[a:b]) if t_rand is not None else None,
                                          None, L.ptr(eps), Ns, S, K, flags, L.ptr(self.rgb_map[a:b]), L.ptr(self.disp[a:b]), L.ptr(self.depth[a:b]),
                                          None, None, None, None, L.ptr(self._ent_sl), st), "cfnerf_render_fwd")
            L.check(lib.cfnerf_loss_fwd_bwd(L.ptr(self.rgb_map[a:b]), L.ptr(target[a:b]), L.ptr(self._ent_sl), Ns, K, beta_sl, N * self.world,
                                            L.ptr(self.d_rgb[a:b]), L.ptr(self._sc_sl), st), "cfnerf_loss_fwd_bwd")
            gen = lib.cfnerf_model_stash_generation(net.handle)
            bwd = lib.cfnerf_render_bwd if i == 0 else lib.cfnerf_render_bwd_accumulate
            L.check(bwd(net.handle, gen, L.ptr(self.d_rgb[a:b]), None, L.ptr(self._d_ent_sl) if self.beta1 else None, L.ptr(self.grad), st),
                    "cfnerf_render_bwd" if i == 0 else "cfnerf_render_bwd_accumulate")
            sc_sum += self._sc_sl
            ent_sum += self._ent_sl
        self.scalars[:3] = sc_sum[:3]
        self.scalars[3] = -10.0 * torch.log10(sc_sum[2])                # HLP:16 on the batch's mse
        self.entropy.copy_(ent_sum / n_sl)
        return self.grad

    def n_slices(self, N):
        """Slices a shard of N rays is walked in: the fewest EQUAL ones of at most max_rays_per_launch rays (1 without a limit)."""
        if not self.max_rays or N <= self.max_rays:
            return 1
        n = -(-N // self.max_rays)
        while N % n:
            n += 1
        return n

    def step(self, H, W, focal, rays, target, **kw):
        """One full train step.  Returns the device tensor [loss, loss_nll, mse, psnr] of the local shard."""
        kw = {k: v for k, v in kw.items() if k in ("t_rand", "eps", "near", "far", "ndc", "lindisp", "white_bkgd", "perturb", "t_vals")}
        dist_on = self.world > 1 or self.force_allreduce
        if kw.get("eps") is None and dist_on:
            kw["eps"] = self._step_eps()
        self.forward_backward(H, W, focal, rays, target, **kw)
        if dist_on:
            self._queue_next_eps()
            self._timed_exchange(self._exchange)
            self._take_next_eps()
        lr = lr_at(self.lrate, self.lrate_decay, self.start, self.t)
        self.t += 1
        net = self.net
        L.check(L.lib().cfnerf_adam_step(net.handle, L.ptr(net.flat.data), L.ptr(self.grad), L.ptr(self.exp_avg),
                                         L.ptr(self.exp_avg_sq), self.t, C.c_float(lr), C.c_float(1.0), L.stream()),
                "cfnerf_adam_step")
        net.mark_packed()
        net.params_serial += 1
        return self.scalars

    # ---- EXTENSION (not in the reference): coarse + fine sampling through the single network -----
    def forward_backward_hierarchical(self, H, W, focal, rays, target, N_samples=64, N_importance=128, coarse_loss=True, t_rand=None,
                                      u_fine=None, eps=None, near=0., far=1., ndc=True, lindisp=False, white_bkgd=False, perturb=1.,
                                      **_ignored):
        """Coarse pass on ``linspace(0,1,N_samples)`` -> ``cfnerf_sample_pdf`` (depths are constants, as nerf-pytorch
        detaches ``z_samples``) -> fine pass on the merged N_samples + N_importance depths, loss and backward.  With
        ``coarse_loss`` the coarse pass keeps a stash and its own loss term is differentiated too
        (nerf-pytorch adds img2mse(rgb0); here the same KDE-NLL as the fine term), so ``self.grad`` is the gradient of
        loss_fine + loss_coarse.  Returns it; ``self.scalars`` holds the fine pass's [loss, nll, mse, psnr]."""
        net, lib = self.net, L.lib()
        dev = net.flat.device
        rays_o, rays_d = rays
        ro, rd = _f32c(rays_o.reshape(-1, 3)), _f32c(rays_d.reshape(-1, 3))
        N, K, S, Ni = rd.shape[0], net.K_samples, int(N_samples), int(N_importance)
        self._buffers(N, K)
        st = L.stream()
        L.check(lib.cfnerf_rays_setup(H, W, float(focal), None, L.ptr(ro), L.ptr(rd), N, 0, int(bool(ndc)), float(near), float(far),
                                      L.ptr(self.packed), st), "cfnerf_rays_setup")
        tv = torch.linspace(0., 1., steps=S).to(dev)
        if perturb > 0.:
            t_rand = _f32c(torch.rand(N, S, device=dev) if t_rand is None else t_rand)
            u = _f32c(torch.rand(N, Ni, device=dev) if u_fine is None else u_fine)
        else:
            t_rand = None
            u = _f32c(torch.linspace(0., 1., steps=Ni).expand(N, Ni).to(dev) if u_fine is None else u_fine)
        eps = _f32c(net.draw_eps() if eps is None else eps)
        net._sync()
        net.ensure_workspace(N, S + Ni, K)
        base = (L.F_LINDISP if lindisp else 0) | (L.F_WHITE_BKGD if white_bkgd else 0) | L.F_TRAIN
        target = _f32c(target)
        beta_w, n_tot = C.c_float(self.beta1 / self.world), N * self.world
        d_ent = L.ptr(self.d_ent) if self.beta1 else None
        # 1. coarse pass -> per-sample weights -> resampled depths.  With a coarse loss term the same launch also stashes its
        #    activations, so the coarse term costs one backward, not a second forward.
        w0 = torch.empty(N, S, K, device=dev)
        L.check(lib.cfnerf_render_fwd(net.handle, L.ptr(self.packed), L.ptr(tv), L.ptr(t_rand), None, L.ptr(eps), N, S, K,
                                      base | (L.F_STASH if coarse_loss else 0), L.ptr(self.rgb_map), L.ptr(self.disp), L.ptr(self.depth),
                                      None, L.ptr(w0), None, None, L.ptr(self.entropy), st), "cfnerf_render_fwd")
        z_all = torch.empty(N, S + Ni, device=dev)
        L.check(lib.cfnerf_sample_pdf(L.ptr(self.packed), L.ptr(tv), L.ptr(t_rand), base, L.ptr(w0), L.ptr(u), N, S, K, Ni, L.ptr(z_all), st),
                "cfnerf_sample_pdf")
        grad_c = None
        if coarse_loss:     # 2. coarse loss term and its backward (before the fine pass replaces the stash)
            L.check(lib.cfnerf_loss_fwd_bwd(L.ptr(self.rgb_map), L.ptr(target), L.ptr(self.entropy), N, K, beta_w, n_tot, L.ptr(self.d_rgb),
                                            L.ptr(self.scalars), st), "cfnerf_loss_fwd_bwd")
            grad_c = torch.empty_like(self.grad)
            L.check(lib.cfnerf_render_bwd(net.handle, lib.cfnerf_model_stash_generation(net.handle), L.ptr(self.d_rgb), None, d_ent,
                                          L.ptr(grad_c), st), "cfnerf_render_bwd")
            self.scalars_coarse = self.scalars.clone()
        # 3. fine pass on the merged depths, loss, backward
        L.check(lib.cfnerf_render_fwd(net.handle, L.ptr(self.packed), L.ptr(tv), None, L.ptr(z_all), L.ptr(eps), N, S + Ni, K, base | L.F_STASH,
                                      L.ptr(self.rgb_map), L.ptr(self.disp), L.ptr(self.depth), None, None, None, None,
                                      L.ptr(self.entropy), st), "cfnerf_render_fwd")
        L.check(lib.cfnerf_loss_fwd_bwd(L.ptr(self.rgb_map), L.ptr(target), L.ptr(self.entropy), N, K, beta_w, n_tot, L.ptr(self.d_rgb),
                                        L.ptr(self.scalars), st), "cfnerf_loss_fwd_bwd")
        L.check(lib.cfnerf_render_bwd(net.handle, lib.cfnerf_model_stash_generation(net.handle), L.ptr(self.d_rgb), None, d_ent,
                                      L.ptr(self.grad), st), "cfnerf_render_bwd")
        if grad_c is not None:
            self.grad.add_(grad_c)
        self.z_vals = z_all
        return self.grad

    def step_hierarchical(self, H, W, focal, rays, target, **kw):
        """One full train step of the coarse + fine EXTENSION (see forward_backward_hierarchical)."""
        dist_on = self.world > 1 or self.force_allreduce
        if kw.get("eps") is None and dist_on:
            kw["eps"] = self._step_eps()
        self.forward_backward_hierarchical(H, W, focal, rays, target, **kw)
        if dist_on:
            self._queue_next_eps()
            self._timed_exchange(lambda: allreduce_sum_(self.gbuf, self.world, self.group, self.force_allreduce))
            self._take_next_eps()
        lr = lr_at(self.lrate, self.lrate_decay, self.start, self.t)
        self.t += 1
        net = self.net
        L.check(L.lib().cfnerf_adam_step(net.handle, L.ptr(net.flat.data), L.ptr(self.grad), L.ptr(self.exp_avg),
                                         L.ptr(self.exp_avg_sq), self.t, C.c_float(lr), C.c_float(1.0), L.stream()),
                "cfnerf_adam_step")
        net.mark_packed()
        net.params_serial += 1
        return self.scalars

    @property
    def global_step(self):
        return self.start + self.t
