"""The TEST-HOOKS library (tests/cfnerf_debug.h, tests/csrc/cfnerf_testhooks.hip -> cf-nerf_amd/build/libcfnerf_testhooks.so): six
cfnerf_debug_* functions that read what the ABI hides (packed operand layout, weight-gradient plan, the activation stash of a handle
the product library created).  The product library exports none of them and never loads this file."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# CFNERF_TESTHOOKS_LIB: another build of the hooks (tools/asan_host.sh runs the host planners under the sanitizers through it)
PATH = os.environ.get("CFNERF_TESTHOOKS_LIB") or os.path.join(ROOT, "cf-nerf_amd", "build", "libcfnerf_testhooks.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            raise RuntimeError(f"{PATH} is missing: `python cf-nerf_amd/build.py` builds it next to the product library (needs hipcc)")
        _lib = C.CDLL(PATH)
    return _lib
