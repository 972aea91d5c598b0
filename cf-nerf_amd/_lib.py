"""ctypes binding of libcfnerf_hip.so (include/cfnerf.h).  There is NO fallback: if the library is
missing or a call fails, a RuntimeError is raised."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CFNERF_LIB: development aid for same-box A/B runs of two builds of the SAME library (never a fallback)
LIB_PATH = os.environ.get("CFNERF_LIB") or os.path.join(HERE, "libcfnerf_hip.so")

F_TRAIN, F_LINDISP, F_WHITE_BKGD, F_STASH = 1, 2, 4, 8


class Cfg(C.Structure):
    _fields_ = [("netdepth", C.c_int32), ("netwidth", C.c_int32), ("multires", C.c_int32),
                ("multires_views", C.c_int32), ("h_alpha_size", C.c_int32), ("h_rgb_size", C.c_int32),
                ("n_flows", C.c_int32)]


_P = C.c_void_p
_SIGS = {
    "cfnerf_version": (C.c_int, []),
    "cfnerf_last_error": (C.c_char_p, []),
    "cfnerf_param_count": (C.c_int64, [C.POINTER(Cfg)]),
    "cfnerf_param_offset": (C.c_int64, [C.POINTER(Cfg), C.c_char_p, C.POINTER(C.c_int64)]),
    "cfnerf_param_key": (C.c_char_p, [C.POINTER(Cfg), C.c_int]),
    "cfnerf_model_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(_P)]),
    "cfnerf_model_destroy": (C.c_int, [_P]),
    "cfnerf_model_set_params": (C.c_int, [_P, _P, _P]),
    "cfnerf_rays_setup": (C.c_int, [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), _P, _P, C.c_int64, C.c_int64, C.c_int,
                                    C.c_float, C.c_float, _P, _P]),
    "cfnerf_ndc_rays": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, _P, _P, C.c_int64, _P, _P, _P]),
    "cfnerf_embed": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P]),
    "cfnerf_sample_points": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P]),
    "cfnerf_render_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P,
                                    _P, _P, _P]),
    "cfnerf_render_eval": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "cfnerf_sample_pdf": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P]),
    "cfnerf_network_fwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P]),
    "cfnerf_composite_fwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "cfnerf_loss_fwd_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_float, C.c_int64, _P, _P, _P]),
    "cfnerf_workspace_bytes": (C.c_int64, [C.POINTER(Cfg), C.c_int64, C.c_int, C.c_int]),
    "cfnerf_model_set_workspace": (C.c_int, [_P, _P, C.c_size_t]),
    "cfnerf_model_stash_generation": (C.c_uint64, [_P]),
    "cfnerf_render_bwd": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P]),
    "cfnerf_render_bwd_accumulate": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P]),
    "cfnerf_network_bwd": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P]),
    "cfnerf_composite_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "cfnerf_grad_early_ranges": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "cfnerf_stream_wait_grad_early": (C.c_int, [_P, _P]),
    "cfnerf_adam_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, _P]),
    "cfnerf_model_set_precision": (C.c_int, [_P, C.c_int]),
    "cfnerf_model_set_flow_math": (C.c_int, [_P, C.c_int]),
    "cfnerf_model_workspace_bytes": (C.c_int64, [_P]),
    "cfnerf_timing_enable": (C.c_int, [_P, C.c_int]),
    "cfnerf_timing_last_ms": (C.c_float, [_P, C.c_int]),
    "cfnerf_timing_fwd_mean_ms": (C.c_float, [_P, C.c_int]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python cf-nerf_amd/build.py` (needs hipcc). "
                "The CF-NeRF hot path has no CPU or PyTorch fallback.")
        if os.environ.get("CFNERF_LIB"):
            import sys
            print(f"cf-nerf_amd: warning: CFNERF_LIB is set - loading {LIB_PATH} instead of the in-tree libcfnerf_hip.so "
                  "(development aid for same-box A/B runs of two builds)", file=sys.stderr, flush=True)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(l, name)  # AttributeError here = header / library out of sync
            except AttributeError:
                if os.environ.get("CFNERF_LIB"):
                    continue           # (an OLDER build under the A/B override: entry points added since are simply absent there)
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().cfnerf_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  The tensor must be contiguous fp32."""
    if t is None:
        return None
    assert t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 4, (t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
