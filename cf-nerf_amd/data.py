"""Device-side ray pool / batch feeder (reference: RUN:860-884, 915-951, get_rays_np HLP:350-357).

The reference builds ``rays_rgb [M,3,3]`` (origin, direction, colour per pixel of every training view) with numpy on
the host, shuffles it there and copies ~1 GB to the GPU.  Here the rays are generated on the GPU by the
``cfnerf_rays_setup`` kernel (the get_rays formula, no NDC - render() applies NDC per batch like the reference),
concatenated with the pixels, shuffled with a device permutation and sliced per step; an exhausted pool is
re-shuffled exactly like RUN:946-949.

Multi-GPU (one process per GPU; replaces the DataParallel split of RUN:330): ``RayPool(..., rank=r, world=G)``.  Every
rank holds the whole pool (as the reference holds ``rays_rgb_train`` on its one device) in the SAME order - ONE
permutation per epoch for all ranks - and a step consumes a window of ``G * N_rand`` rows of which rank r takes rows
``[i + r * N_rand, i + (r + 1) * N_rand)``: the union of the shards is the batch a one-process pool with
``N_rand_global = G * N_rand`` delivers, every rank re-shuffles at the same step, and nothing is exchanged per step.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .api import _pose_arg


class RayPool:
    def __init__(self, images, poses, H, W, focal, i_train, N_rand, generator=None, shuffle=True, rank=0, world=1, seed=None,
                 group=None, sync="auto"):
        """images [V,H,W,3] in [0,1]; poses [V,3,>=4]; i_train: indices of the training views; ``N_rand`` rays PER RANK.

        One rank (``world == 1``): the permutations come from ``generator`` (a device generator) or torch's global one,
        as in rounds 1-3.  ``world > 1``: all ranks must see ONE permutation per epoch -
          sync="seed"       every rank draws it from a CPU generator seeded with (``seed``, epoch): identical by
                            construction, no communication (``seed`` must be the same number on every rank; default 0);
          sync="broadcast"  rank 0 draws it (from ``generator`` / ``seed``) and broadcasts it over ``group`` once per epoch;
          sync="auto"       "broadcast" when torch.distributed is initialised, else "seed"."""
        self.H, self.W, self.focal = int(H), int(W), float(focal)
        self._init_feeder(N_rand, generator, rank, world, seed, group, sync)
        images = torch.as_tensor(images, dtype=torch.float32)
        poses = torch.as_tensor(poses, dtype=torch.float32).cpu()        # the poses travel in kernel arguments: one fetch for all views
        if not images.is_cuda:
            images = images.cuda()
        dev = images.device
        lib = L.lib()
        n = self.H * self.W
        chunks = []
        packed = torch.empty(n, 11, device=dev)
        for v in i_train:
            arr, _keep = _pose_arg(poses[int(v)])
            L.check(lib.cfnerf_rays_setup(self.H, self.W, self.focal, arr, None, None, n, 0, 0, 0.0, 1.0, L.ptr(packed), L.stream()),
                    "cfnerf_rays_setup")
            rays = torch.stack([packed[:, 0:3], packed[:, 3:6], images[int(v)].reshape(n, 3)], 1)      # [n, ro+rd+rgb, 3]
            chunks.append(rays)
        self.rays_rgb = torch.cat(chunks, 0)                                                           # [(V_train)*H*W, 3, 3]
        if shuffle:
            self._shuffle()

    @classmethod
    def from_rays_rgb(cls, rays_rgb, N_rand, generator=None, shuffle=True, rank=0, world=1, seed=None, group=None, sync="auto"):
        """The feeder over a ready ``rays_rgb [M,3,3]`` table (the reference's ``rays_rgb_train``, RUN:869-874) - any device."""
        self = cls.__new__(cls)
        self.H = self.W = self.focal = None
        self._init_feeder(N_rand, generator, rank, world, seed, group, sync)
        self.rays_rgb = rays_rgb
        if shuffle:
            self._shuffle()
        return self

    def _init_feeder(self, N_rand, generator, rank, world, seed, group, sync):
        self.N_rand = int(N_rand)
        self.rank, self.world, self.group = int(rank), int(world), group
        if not (0 <= self.rank < self.world):
            raise ValueError(f"rank {rank} outside world {world}")
        if sync not in ("auto", "seed", "broadcast"):
            raise ValueError(f"sync={sync!r}")
        if sync == "auto":
            import torch.distributed as dist
            sync = "broadcast" if (self.world > 1 and dist.is_available() and dist.is_initialized()) else "seed"
        self.sync = sync
        self.seed = None if seed is None else int(seed)
        self.generator = generator
        self.i_batch = 0
        self.epoch = 0

    # ---- ONE permutation per epoch, the same on every rank ---------------------------------------------------------
    @staticmethod
    def _mix(seed: int, epoch: int) -> int:
        """(seed, epoch) -> generator seed through a splitmix64-style finaliser: distinct pairs do not collide the way
        ``seed * 1000003 + epoch`` did ((0, 1000003) and (1, 0)), and neighbouring epochs get unrelated streams."""
        x = ((seed & 0xffffffffffffffff) * 0x9E3779B97F4A7C15 + (epoch + 1) * 0xBF58476D1CE4E5B9) & 0xffffffffffffffff
        x ^= x >> 30
        x = (x * 0xBF58476D1CE4E5B9) & 0xffffffffffffffff
        x ^= x >> 27
        x = (x * 0x94D049BB133111EB) & 0xffffffffffffffff
        x ^= x >> 31
        return x & 0x7fffffffffffffff

    def _seeded_permutation(self, M, dev):
        """Drawn ON the pool's device from a per-epoch seeded generator of that device: no 8 M-byte host permutation and upload at
        every epoch boundary (30 MB at fern size, on every rank, stalling the host), and the same stream on every rank because every
        rank runs the same generator algorithm on the same kind of device."""
        g = torch.Generator(device=dev).manual_seed(self._mix(self.seed or 0, self.epoch))
        return torch.randperm(M, device=dev, generator=g)

    def _check_same_permutation(self, idx):
        """sync="seed" draws the epoch's permutation on every rank with no exchange - and relies on every rank's generator being the same
        algorithm on the same kind of device (torch's device-side randperm picks its launch geometry from the device: another CU count or
        partition mode, or a host-resident pool on one rank, would give another stream and shards that overlap or miss rays silently).  Once
        per epoch the ranks therefore compare a position-weighted checksum of the indices (two scalars, MAX all-reduce; nothing when no
        process group exists) and a mismatch raises instead of training on."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        w = torch.arange(1, idx.numel() + 1, device=idx.device, dtype=torch.int64) % 65521
        c = (idx * w).sum()                                                  # < 2^63 up to ~10^7 rays
        t = torch.stack([c, -c])
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
            t = host
        elif not t.is_cuda and dist.get_backend(self.group) == "nccl":
            stage = t.to(torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(stage, op=dist.ReduceOp.MAX, group=self.group)
            t = stage
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        if int(t[0]) != -int(t[1]):
            raise RuntimeError(f"RayPool(sync='seed'): the ranks drew DIFFERENT permutations for epoch {self.epoch} from seed {self.seed} "
                               "(different devices or generator algorithms); use sync='broadcast'")

    def _permutation(self):
        M, dev = self.rays_rgb.shape[0], self.rays_rgb.device
        if self.world == 1 and self.seed is None:
            return torch.randperm(M, device=dev, generator=self.generator)
        if self.world == 1 or self.sync == "seed":
            idx = self._seeded_permutation(M, dev)
            if self.world > 1:
                self._check_same_permutation(idx)
            return idx
        import torch.distributed as dist
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        if self.rank == 0:
            idx = self._seeded_permutation(M, dev) if self.seed is not None else torch.randperm(M, device=dev, generator=self.generator)
        else:
            idx = torch.empty(M, dtype=torch.int64, device=dev)
        backend = dist.get_backend(self.group)
        if backend == "gloo" and idx.is_cuda:                               # CPU tests / two test ranks on one GPU
            host = idx.cpu()
            dist.broadcast(host, src=src, group=self.group)
            idx = host.to(dev)
        elif backend == "nccl" and not idx.is_cuda:                         # a host-resident pool under an RCCL group: RCCL moves device
            stage = idx.to(torch.device("cuda", torch.cuda.current_device()))      # buffers only, so the indices cross through this rank's GPU
            dist.broadcast(stage, src=src, group=self.group)
            idx = stage.to(dev)
        else:
            dist.broadcast(idx, src=src, group=self.group)
        return idx

    def _shuffle(self):
        self.rays_rgb = self.rays_rgb[self._permutation()]

    def __len__(self):
        return self.rays_rgb.shape[0]

    @property
    def global_batch(self):
        return self.N_rand * self.world

    def next_batch(self):
        """RUN:942-951: ``batch_rays [2,N,3]`` (origins, directions) and ``target_s [N,3]`` - this rank's shard of the
        step's window.  The last window of an epoch is short like the reference's last slice; with several ranks it is cut
        to a multiple of ``world`` so that the shards stay equal (the Trainer's gradient normalisation assumes that)."""
        M, G = self.rays_rgb.shape[0], self.N_rand * self.world
        if M < self.world:
            raise ValueError(f"a pool of {M} rays cannot feed {self.world} ranks")
        if M - self.i_batch < self.world:                     # fewer rays left than ranks: this epoch is over
            self._next_epoch()
        n_win = min(G, M - self.i_batch)
        per = self.N_rand if n_win == G else n_win // self.world
        lo = self.i_batch + self.rank * per
        batch = self.rays_rgb[lo:lo + per]
        batch = torch.transpose(batch, 0, 1)
        batch_rays, target_s = batch[:2], batch[2]
        self.i_batch += G
        if self.i_batch >= M:
            self._next_epoch()                                # "Shuffle data after an epoch!"
        return batch_rays, target_s

    def _next_epoch(self):
        self.epoch += 1
        self._shuffle()
        self.i_batch = 0
