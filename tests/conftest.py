import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)

# Multi-process GPU tests need FRESH children: a process that has initialised the GPU must never fork-and-exec (on
# this pool that takes the machine down).  So the fork server is started HERE, at collection time, before any test
# of this process has touched the GPU; the ranks of tests/test_hip_multirank.py are forked from that clean server.
FORKSERVER_CTX = None
if os.path.exists("/dev/kfd"):
    import multiprocessing as _mp
    from multiprocessing import forkserver as _fs
    FORKSERVER_CTX = _mp.get_context("forkserver")
    _fs.ensure_running()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("CFNERF_GRAD_STATS_ONLY"):
        raise pytest.UsageError("CFNERF_GRAD_STATS_ONLY is set: rounds 2-3 used it to switch the gradient assertions off for surveys; it is no "
                                "longer honoured and must not reach a test run (CFNERF_GRAD_STATS=<path> alone logs every comparison AND judges it)")
    # never test a stale library: rebuild in-tree when a source is newer than libcfnerf_hip.so (no-op otherwise)
    import importlib.util
    import shutil
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        spec = importlib.util.spec_from_file_location("cfnerf_build", os.path.join(ROOT, "cf-nerf_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return load
