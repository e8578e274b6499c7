"""ctypes binding of libgpmpc_hip.so (C ABI: include/gpmpc.h).

torch is imported first on purpose: its bundled HIP runtime (SONAME libamdhip64.so.7)
must be the one already mapped when the library is dlopen'ed, so that device pointers
and streams handed over from torch tensors belong to the same runtime instance.
"""
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPMPC_LIB_PATH: another build of the SAME library (diagnostic builds with in-kernel stamps, A/B runs); never a fallback
LIB_PATH = os.environ.get("GPMPC_LIB_PATH") or os.path.join(_HERE, "csrc", "libgpmpc_hip.so")

MAX_D = 8
MAX_DS = 8
WANT_GRAD = 1
COV_BUG_COMPAT = 2
USE_GRAPH = 4
FP32_ACCUM = 8
FP32_ALL = 16

_ERR = {-1: "bad argument", -2: "device allocation failed", -3: "HIP call / kernel launch failed",
        -4: "workspace too small", -5: "pack not built",
        -6: "current HIP device differs from the device the pack was created on"}


class LibraryMissing(RuntimeError):
    pass


class GpmpcError(RuntimeError):
    pass


class CostParamsC(ctypes.Structure):
    _fields_ = [("gamma", ctypes.c_double),
                ("Q", ctypes.c_double * (MAX_DS * MAX_DS)),
                ("R", ctypes.c_double * (MAX_D * MAX_D)),
                ("R_delta", ctypes.c_double * (MAX_D * MAX_D)),
                ("x_ref", ctypes.c_double * MAX_DS),
                ("u_ref", ctypes.c_double * MAX_D),
                ("last_u", ctypes.c_double * MAX_D),
                ("has_R_delta", ctypes.c_int),
                ("reserved", ctypes.c_int)]


_vp, _i, _d, _sz, _u = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t, ctypes.c_uint
_dp = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); every symbol include/gpmpc.h declares
SIGNATURES = {
    "gpmpc_device_count": (_i, []),
    "gpmpc_version": (ctypes.c_char_p, []),
    "gpmpc_last_error": (ctypes.c_char_p, []),
    "gpmpc_pack_create": (_i, [ctypes.POINTER(_vp), _i, _i, _i]),
    "gpmpc_pack_destroy": (_i, [_vp]),
    "gpmpc_pack_resize": (_i, [_vp, _i]),
    "gpmpc_pack_reload_tuning": (_i, [_vp]),
    "gpmpc_pack_graph_captures": (ctypes.c_longlong, [_vp]),
    "gpmpc_store_host": (_i, [_vp, _vp, _sz, _vp]),
    "gpmpc_build_ky": (_i, [_i, _i, _vp, _dp, _d, _d, _vp, _vp, _vp]),
    "gpmpc_pack_build": (_i, [_vp, _vp, _vp, _vp, _dp, _dp, _vp]),
    "gpmpc_pack_build_strided": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _dp, _dp, _vp]),
    "gpmpc_pack_build_beta": (_i, [_vp, _vp, _vp, _vp, _dp, _dp, _vp]),
    "gpmpc_pack_enable_fullcov": (_i, [_vp, _vp]),
    "gpmpc_pack_dims": (_i, [_vp] + [ctypes.POINTER(_i)] * 4),
    "gpmpc_pack_shared_lambda": (_i, [_vp]),
    "gpmpc_pack_export": (_i, [_vp, _vp, _vp, _vp]),
    "gpmpc_moment_match_workspace_bytes": (_sz, [_vp, _i]),
    "gpmpc_moment_match": (_i, [_vp, _i, _vp, _vp, _u] + [_vp] * 10 + [_vp, _sz, _vp]),
    "gpmpc_cost": (_i, [_i, _i, _i, _i, ctypes.POINTER(CostParamsC), _vp, _vp, _vp, _vp, _vp]),
    "gpmpc_cost_grad": (_i, [_i, _i, _i, _i, ctypes.POINTER(CostParamsC)] + [_vp] * 8),
    "gpmpc_rollout_jac_workspace_bytes": (_sz, [_vp, _i, _i]),
    "gpmpc_rollout_jac": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpmpc_rollout_vjp": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gpmpc_rollout_workspace_bytes": (_sz, [_vp, _i, _i, _u]),
    "gpmpc_plan_describe": (_i, [_vp, _i, _i, _u, ctypes.c_char_p, _sz]),
    "gpmpc_pack_autotune": (_i, [_vp, _i, _i, _u, ctypes.c_char_p, _sz]),
    "gpmpc_pack_autotune_clear": (_i, [_vp]),
    "gpmpc_rollout": (_i, [_vp, _i, _i, _vp, _vp, ctypes.POINTER(CostParamsC), _u, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpmpc_objective_gradient": (_i, [_vp, _i, _dp, _dp, ctypes.POINTER(CostParamsC), _u, _dp, _vp]),
    "gpmpc_rollout_fullcov_workspace_bytes": (_sz, [_vp, _i, _i, _u]),
    "gpmpc_rollout_fullcov_describe": (_i, [_vp, _i, _i, _u, ctypes.c_char_p, _sz]),
    "gpmpc_rollout_fullcov": (_i, [_vp, _i, _i, _vp, _vp, ctypes.POINTER(CostParamsC), _u, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpmpc_timing_enable": (_i, [_i]),
    "gpmpc_debug_run_list": (_i, [_i, _i, _i, ctypes.POINTER(_i), _i, ctypes.POINTER(_i)]),
    "gpmpc_debug_xcd_order": (_i, [_i, _i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "gpmpc_pair_kernel_time": (_i, [_dp, ctypes.POINTER(ctypes.c_longlong), _i]),
    "gpmpc_pair_kernel_time_class": (_i, [_i, _dp, ctypes.POINTER(ctypes.c_longlong)]),
    "gpmpc_matvec": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "gpmpc_kinv_append_workspace_bytes": (_sz, [_i]),
    "gpmpc_kinv_append": (_i, [_i, _vp, _vp, _d, _vp, _vp, _sz, _vp]),
    "gpmpc_gp_append_workspace_bytes": (_sz, [_i, _i]),
    "gpmpc_gp_append": (_i, [_i, _i, _vp, _vp, _dp, _d, _d, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "gpmpc_predict_workspace_bytes": (_sz, [_i, _i, _i]),
    "gpmpc_predict": (_i, [_i, _i, _vp, _dp, _d, _vp, _vp, _d, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpmpc_ml_grad_workspace_bytes": (_sz, [_i, _i]),
    "gpmpc_ml_grad": (_i, [_i, _i, _vp, _vp, _vp, _vp, _dp, _d, _d, _vp, _vp, _sz, _vp]),
}

_lib = None


def lib():
    """The loaded library; raises LibraryMissing (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gaussian_process_mpc_amd/csrc`. There is no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        # A/B runs against an OLDER build of the library (GPMPC_LIB_PATH + GPMPC_LIB_ALLOW_MISSING=1, tools/lib_ab.py) may lack
        # entry points added since; the product library must export every declared symbol
        tolerant = bool(os.environ.get("GPMPC_LIB_PATH")) and os.environ.get("GPMPC_LIB_ALLOW_MISSING") == "1"
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(h, name)      # AttributeError if a declared symbol is missing
            except AttributeError:
                if tolerant:
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        detail = lib().gpmpc_last_error().decode() if rc == -3 else ""
        raise GpmpcError(f"{what} failed: {_ERR.get(rc, rc)} {detail}".strip())


def require_gpu():
    """Device for all numerical work.  Raises if no GPU is visible (no CPU fallback)."""
    lib()
    if not torch.cuda.is_available():
        raise GpmpcError("no HIP device visible: the GP-MPC rollout runs on the MI355X only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr(device=None):
    """Current HIP stream of `device` (default: the current device) as a void*.  The device index is passed explicitly:
    torch.cuda.current_stream() WITHOUT one goes through torch.cuda.is_available() on every call, which costs ~0.1 ms per
    call in a fresh process (measured on the solver-callback path: it doubled the latency of a C1 callback pair)."""
    idx = device.index if isinstance(device, torch.device) and device.index is not None else torch.cuda.current_device()
    return ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)


def ptr(t):
    """Device pointer of a contiguous float64 CUDA tensor (or NULL for None)."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return ctypes.c_void_p(t.data_ptr())


def host_doubles(a):
    import numpy as np
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a, a.ctypes.data_as(_dp)
