"""Device-side ray pool / batch feeder (reference: RUN:860-884, 915-951, get_rays_np HLP:350-357).

The reference builds ``rays_rgb [M,3,3]`` (origin, direction, colour per pixel of every training view) with numpy on
the host, shuffles it there and copies ~1 GB to the GPU.  Here the rays are generated on the GPU by the
``cfnerf_rays_setup`` kernel (the get_rays formula, no NDC - render() applies NDC per batch like the reference),
concatenated with the pixels, shuffled with a device permutation and sliced per step; an exhausted pool is
re-shuffled exactly like RUN:946-949.
"""
from __future__ import annotations


import torch

from . import _lib as L
from .api import _pose_arg


class RayPool:
    def __init__(self, images, poses, H, W, focal, i_train, N_rand, generator=None, shuffle=True):
        """images [V,H,W,3] in [0,1]; poses [V,3,>=4]; i_train: indices of the training views."""
        images = torch.as_tensor(images, dtype=torch.float32)
        poses = torch.as_tensor(poses, dtype=torch.float32).cpu()        # the poses travel in kernel arguments: one fetch for all views
        if not images.is_cuda:
            images = images.cuda()
        dev = images.device
        self.H, self.W, self.focal, self.N_rand = int(H), int(W), float(focal), int(N_rand)
        self.generator = generator
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
        self.i_batch = 0
        self.epoch = 0

    def _shuffle(self):
        M = self.rays_rgb.shape[0]
        idx = torch.randperm(M, device=self.rays_rgb.device, generator=self.generator)
        self.rays_rgb = self.rays_rgb[idx]

    def __len__(self):
        return self.rays_rgb.shape[0]

    def next_batch(self):
        """RUN:942-951: ``batch_rays [2,N,3]`` (origins, directions) and ``target_s [N,3]``."""
        batch = self.rays_rgb[self.i_batch:self.i_batch + self.N_rand]
        batch = torch.transpose(batch, 0, 1)
        batch_rays, target_s = batch[:2], batch[2]
        self.i_batch += self.N_rand
        if self.i_batch >= self.rays_rgb.shape[0]:
            self._shuffle()                                   # "Shuffle data after an epoch!"
            self.i_batch = 0
            self.epoch += 1
        return batch_rays, target_s
