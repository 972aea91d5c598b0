"""Diagnostic: the HIP path over fixture G19 (the reference's own 120-step training curve); prints the deviations the -m gpu
test bounds (tests/g19_common.py).   python tests/tools/g19_hip_curve.py   [G19_PREC=bf16x3]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import g19_common as GC

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g19_psnr_curve.npz")))
dl, dp, held = GC.hip_curve(g, os.environ.get("G19_PREC"))
for k, v in held.items():
    print("held-out PSNR - reference at step", k, v)
for a, b, _, _ in GC.CURVE_BOUNDS:
    print(f"steps {a}-{b}: max rel loss diff {dl[a:b].max():.3e}, max |train PSNR diff| {dp[a:b].max():.3e} dB")
