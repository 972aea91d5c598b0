"""-m gpu: the drop-in boundary used WITHOUT Python or torch in the process.  tests/abi_consumer.c is a plain C99 program over
include/cfnerf.h (built by cf-nerf_amd/build.py with gcc): hipMalloc'ed buffers, then the reference's train iteration through the C
ABI - rays set-up, fused forward with stash, loss, backward, Adam - and a comparison with what the REAL reference computed for the
same inputs (golden fixture G5/G7, written here into the program's flat case file).  The program runs as a child of a fresh
fork-server process, never as a fork + exec of this (GPU-initialised) pytest process."""
import os
import struct
import tempfile

import numpy as np
import pytest
import torch

from oracle import cfnerf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cf-nerf_amd", "build", "abi_consumer")


def write_case(path, g, lr=5e-4):
    """flat binary case file of abi_consumer.c; parameters in the flat layout come from the layout queries of the library
    (host-side, no device) and the oracle's deterministic generator - the weights the fixture was made with"""
    import ctypes as C
    from cfnerf_amd import _lib as L
    lib = L.lib()
    cfg = O.OracleCfg(netwidth=int(g["netwidth"]), K_samples=int(g["K"]))
    ccfg = L.Cfg(cfg.netdepth, cfg.netwidth, cfg.multires, cfg.multires_views, cfg.h_alpha_size, cfg.h_rgb_size, cfg.n_flows)
    n_params = int(lib.cfnerf_param_count(C.byref(ccfg)))
    p = O.make_params(cfg, int(g["seed"]))
    flat = np.zeros(n_params, np.float32)
    grad = np.full(n_params, np.nan, np.float32)
    adam = np.full(n_params, np.nan, np.float32)
    for key, v in p.items():
        numel = C.c_int64(0)
        off = int(lib.cfnerf_param_offset(C.byref(ccfg), key.encode(), C.byref(numel)))
        assert off >= 0 and numel.value == v.numel(), key
        flat[off:off + numel.value] = v.reshape(-1).numpy()
        if ("grad." + key) in g:
            grad[off:off + numel.value] = g["grad." + key].reshape(-1)
        if ("adam1." + key) in g:
            adam[off:off + numel.value] = g["adam1." + key].reshape(-1)
    N, S, K = int(g["rays"].shape[1]), 128, int(g["K"])
    flags = 1 | (2 if bool(g["lindisp"]) else 0) | (4 if bool(g["white_bkgd"]) else 0)        # CFNERF_F_TRAIN | LINDISP | WHITE_BKGD
    eps = np.concatenate([g["eps_rgb"], g["eps_alpha"]], -1).astype(np.float32)                 # [K,4] = rgb3, alpha1
    with open(path, "wb") as f:
        f.write(b"CFNB" + struct.pack("<i", 1))
        f.write(struct.pack("<7i", cfg.netdepth, cfg.netwidth, cfg.multires, cfg.multires_views, cfg.h_alpha_size, cfg.h_rgb_size, cfg.n_flows))
        f.write(struct.pack("<7i", N, S, K, int(g["H"]), int(g["W"]), int(bool(g["ndc"])), flags))
        f.write(struct.pack("<5f", float(g["focal"]), float(g["near"]), float(g["far"]), float(g["beta1"]), lr))
        f.write(struct.pack("<q", n_params))
        for a in (flat, g["rays"][0], g["rays"][1], O.t_vals_table(torch.float32).numpy(), g["t_rand"], eps, g["target"],
                  g["rgb_map"], g["disp_map"], g["depth_map"],
                  np.array([g["loss"], g["loss_nll"], g["mse"], g["psnr"]]), grad, adam):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    return int(np.isfinite(grad).sum()), int(np.isfinite(adam).sum())


@pytest.mark.parametrize("tag", ["w64_ndc", "w64_nondc_lindisp_wb", "w256_ndc"])
def test_plain_c_program_reproduces_the_references_train_iteration(golden, tag):
    from conftest import FORKSERVER_CTX as ctx
    import mp_workers
    assert os.path.exists(EXE), "cf-nerf_amd/build.py builds tests/abi_consumer.c next to the library"
    g = golden(f"g57_render_{tag}")
    with tempfile.TemporaryDirectory() as d:
        case, out = os.path.join(d, "case.bin"), os.path.join(d, "out.txt")
        n_grad, n_adam = write_case(case, g)
        assert n_grad > 1000 and n_adam > 1000
        p = ctx.Process(target=mp_workers.run_program, args=([EXE, case], out))
        p.start()
        p.join(300)
        text = open(out).read() if os.path.exists(out) else ""
        assert p.exitcode == 0 and "abi_consumer: OK" in text, f"exit code {p.exitcode}\n{text}"
