"""CPU: the identity behind the round-6 transmittance adjoint (csrc/cfnerf_device.h: comp_adjoint_D).  The reference's autograd evaluates
d loss / d alpha_s = g_s T_s - (sum_{j>s} g_j w_j) / x_s  (torch.cumprod's backward, RUN:443); the kernels carry the quantity that is left after
the cancellation,  d alpha_s = T_s D_s,  D_s = (g_s - g_{s+1}) + x_{s+1} D_{s+1} - 1e-10 g_{s+1}.  Same number in exact arithmetic: checked here in
fp64 against torch's autograd of the density path of one synthetic ray (several chunks of 64 samples, white background on and off), and
in fp32 in the kernel's own order (a 64-lane suffix scan of affine maps per chunk + a carry: tests/tools/density_bisect.py 'D32tree')."""
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("density_bisect", os.path.join(ROOT, "tests", "tools", "density_bisect.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("S,K,wb", [(130, 2, True), (257, 3, False), (64, 4, True), (16, 2, False)])
def test_cancelled_quantity_recurrence_is_the_cumprod_adjoint(S, K, wb):
    db = _tool()
    g = torch.Generator().manual_seed(S * 10 + K)
    th = torch.randn(S, 128, generator=g) * 0.5
    th[:, 96:104] = torch.tanh(th[:, 96:104])                         # the diagonals are stored tanh-ed (MOD:341-348)
    z = torch.sort(torch.rand(S, generator=g)).values
    raw = torch.randn(S, K, 4, generator=g)
    eps = torch.randn(K, generator=g)
    mean, std = 0.1, 1.2
    raw[:, :, 3] = db.flow_chain(th[:, 96:108], eps * std + mean)[0]     # the stashed raw IS the flow chain's output
    d = dict(theta=th, raw=raw, z=z, rays=torch.randn(11, generator=g))
    G = torch.randn(3, K, generator=g)
    E, Em, Es, fw = db.autograd_ref(d, G, eps, mean, std, wb, torch.float64)
    d["at"] = torch.stack([(1 - fw["alpha"]).float(), fw["T"].float()], -1)      # what the forward stashes: (e, T) in fp32
    ha = torch.randn(S, 32, generator=g)
    # the suffix form in fp64 at exact inputs IS autograd
    gs, ms, ss, _ = db.kernel_formula(d, G, eps, mean, std, wb, set())
    assert db.rel(gs, E) <= 1e-12 and abs(ms - Em) <= 1e-12 * abs(Em) and abs(ss - Es) <= 1e-12 * abs(Es)
    # the recurrence: fp64 arithmetic on the fp32 stash (limited by the stash's own rounding), then fp32 in the kernel's scan order
    for how, tol in (("D64", 2e-6), ("D32seq", 4e-6), ("D32tree", 4e-6)):
        r = db.summarize(how, *db.alt_formula(d, G, eps, mean, std, wb, how), E, Em, Es, ha)
        assert max(r["per_sample"], r["bias_sums"], r["head_weights"], r["alpha_mean"], r["alpha_std"]) <= tol, (how, r)
