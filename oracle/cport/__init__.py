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
        alt = os.environ.get("GPMPC_CPORT_LIB")           # the sanitizer build (make -C oracle/cport asan), tools/run_sanitizers.sh
        if alt:
            _LIB = ctypes.CDLL(alt)
            _LIB.gpmpc_cpu_rollout.restype = ctypes.c_int
            _LIB.gpmpc_cpu_rollout_fullcov.restype = ctypes.c_int
            return _LIB
        so = os.path.join(_HERE, "libgpmpc_cpu.so")
        srcs = [os.path.join(_HERE, f) for f in ("gpmpc_cpu.c", "gpmpc_cpu_fullcov.c", "gpmpc_cpu_ld.c")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _LIB = ctypes.CDLL(so)
        _LIB.gpmpc_cpu_rollout.restype = ctypes.c_int
        _LIB.gpmpc_cpu_rollout_fullcov.restype = ctypes.c_int
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


def rollout_extended(pb, Ky_inv, x0=None, U=None, nthreads=0):
    """Forward pass (means, vars) of the diagonal rollout with every operation in x87 extended precision
    (oracle/cport/gpmpc_cpu_ld.c): the yardstick for the noise-level accuracy sweep.  Small N only."""
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))      # noqa: E731
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))          # noqa: E731
    X, Y, lam, sf = c(pb["X"]), c(pb["Y"]), c(pb["lambdas"]), c(pb["sigma_f"])
    x0 = c(pb["x0"] if x0 is None else x0).reshape(-1, pb["ds"])
    U = c(pb["U"] if U is None else U)
    U = U.reshape(-1, U.shape[-2], pb["da"])
    B, H = U.shape[0], U.shape[1]
    K = c(Ky_inv)
    means = np.zeros((B, H + 1, pb["ds"])); vars_ = np.zeros((B, H + 1, pb["ds"]))
    lib().gpmpc_cpu_rollout_ld.restype = ctypes.c_int
    rc = lib().gpmpc_cpu_rollout_ld(X.shape[0], pb["ds"], pb["da"], H, B, p(X), p(K), p(Y), p(lam), p(sf), p(x0), p(U),
                                    p(means), p(vars_), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"gpmpc_cpu_rollout_ld failed: {rc}")
    return {"means": means, "vars": vars_}


def rollout_fullcov(pb, Ky_inv, gamma, x0=None, U=None, dirs=None, nthreads=0):
    """FULL-covariance rollout (oracle/cport/gpmpc_cpu_fullcov.c; BASELINE config 5).  dirs: optional (B, ndir, H, da)
    directions; ``ddir[b, d]`` is then the directional derivative dirs[b, d] . dcost_b/dU by the complex step.
    Returns dict(means (B,H+1,ds), covs (B,H+1,ds,ds), cost (B,), ddir (B,ndir))."""
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))      # noqa: E731
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))          # noqa: E731
    ds, da = pb["ds"], pb["da"]
    X, Y, lam, sf = c(pb["X"]), c(pb["Y"]), c(pb["lambdas"]), c(pb["sigma_f"])
    x0 = c(pb["x0"] if x0 is None else x0).reshape(-1, ds)
    U = c(pb["U"] if U is None else U)
    U = U.reshape(-1, U.shape[-2], da)
    B, H = U.shape[0], U.shape[1]
    K, Q, R, xr, ur = c(Ky_inv), c(pb["Q"]), c(pb["R"]), c(pb["x_ref"]), c(pb["u_ref"])
    dirs = np.zeros((B, 0, H, da)) if dirs is None else c(dirs).reshape(B, -1, H, da)
    ndir = dirs.shape[1]
    means = np.zeros((B, H + 1, ds)); covs = np.zeros((B, H + 1, ds, ds))
    cost = np.zeros(B); ddir = np.zeros((B, max(ndir, 1)))
    rc = lib().gpmpc_cpu_rollout_fullcov(X.shape[0], ds, da, H, B, p(X), p(K), p(Y), p(lam), p(sf), p(x0), p(U),
                                         ctypes.c_double(gamma), p(Q), p(R), p(xr), p(ur), p(means), p(covs), p(cost),
                                         ndir, p(dirs if ndir else np.zeros(1)), p(ddir), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"gpmpc_cpu_rollout_fullcov failed: {rc}")
    return {"means": means, "covs": covs, "cost": cost, "ddir": ddir[:, :ndir]}


def moment_match_fullcov(X, Ky_inv, Y, lambdas, sigma_f, u, S, nthreads=0):
    """One moment-matching step for N(u, S) with a FULL covariance S: (mean (ds,), cov (ds, ds)) of the ds GP outputs
    (variances on the diagonal, consistent-form cross-covariances off it).  Y: (N, ds)."""
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))      # noqa: E731
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))          # noqa: E731
    X, K, Y, lam, sf, u, S = c(X), c(Ky_inv), c(Y), c(lambdas), c(sigma_f), c(u), c(S)
    N, D = X.shape
    ds = Y.shape[1]
    lib().gpmpc_cpu_moment_match_fullcov.restype = ctypes.c_int
    mean = np.zeros(ds); cov = np.zeros((ds, ds))
    rc = lib().gpmpc_cpu_moment_match_fullcov(N, ds, D, p(X), p(K), p(Y), p(lam), p(sf), p(u), p(S), p(mean), p(cov), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"gpmpc_cpu_moment_match_fullcov failed: {rc}")
    return mean, cov
