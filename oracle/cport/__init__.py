"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of the plain-C / OpenMP CPU port (oracle/cport/gpmpc_cpu.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgpmpc_cpu.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "gpmpc_cpu.c")):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _LIB = ctypes.CDLL(so)
        _LIB.gpmpc_cpu_rollout.restype = ctypes.c_int
    return _LIB


def rollout(pb, Ky_inv, gamma, x0=None, U=None, nthreads=0):
    """pb: dict from synth_problem; Ky_inv: (ds, N, N).  Returns dict(means, vars, cost, grad) as numpy arrays."""
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))      # noqa: E731
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))          # noqa: E731
    X, Y, lam, sf = c(pb["X"]), c(pb["Y"]), c(pb["lambdas"]), c(pb["sigma_f"])
    x0 = c(pb["x0"] if x0 is None else x0).reshape(-1, pb["ds"])
    U = c(pb["U"] if U is None else U)
    U = U.reshape(-1, U.shape[-2], pb["da"])
    B, H = U.shape[0], U.shape[1]
    K, Q, R, xr, ur = c(Ky_inv), c(pb["Q"]), c(pb["R"]), c(pb["x_ref"]), c(pb["u_ref"])
    means = np.zeros((B, H + 1, pb["ds"])); vars_ = np.zeros((B, H + 1, pb["ds"]))
    cost = np.zeros(B); grad = np.zeros((B, H, pb["da"]))
    rc = lib().gpmpc_cpu_rollout(X.shape[0], pb["ds"], pb["da"], H, B, p(X), p(K), p(Y), p(lam), p(sf), p(x0), p(U),
                                 ctypes.c_double(gamma), p(Q), p(R), p(xr), p(ur), p(means), p(vars_), p(cost), p(grad),
                                 int(nthreads))
    if rc != 0:
        raise RuntimeError(f"gpmpc_cpu_rollout failed: {rc}")
    return {"means": means, "vars": vars_, "cost": cost, "grad": grad}
