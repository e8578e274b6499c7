"""Thin, batched host API over the C ABI: GP pack, cost parameters, rollout, moment matching.

Everything here is plumbing (device buffers, streams, argument marshalling); the
arithmetic lives in csrc/*.hip.  Tensors are float64 CUDA tensors; numpy inputs are
copied to the current device.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import CostParamsC, check, lib, ptr, require_gpu, stream_ptr, host_doubles


def _dev(a, device):
    if isinstance(a, torch.Tensor):
        return a.detach().to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=device)


class CostParams:
    """Parameters of RiskSensitiveMPC.cost_torch (reference src/mpc.py:156-200)."""

    def __init__(self, gamma, Q, R, R_delta=None, x_ref=None, u_ref=None, last_u=None):
        Q = np.atleast_2d(np.asarray(Q, dtype=np.float64))
        R = np.atleast_2d(np.asarray(R, dtype=np.float64))
        self.ds, self.da = Q.shape[0], R.shape[0]
        if Q.shape != (self.ds, self.ds) or R.shape != (self.da, self.da):
            raise ValueError("Q and R must be square")
        if self.ds > _lib.MAX_DS or self.da > _lib.MAX_D:
            raise ValueError("state / input dimension exceeds the library limits")
        c = CostParamsC()
        c.gamma = float(gamma)
        c.Q[:self.ds * self.ds] = Q.reshape(-1).tolist()
        c.R[:self.da * self.da] = R.reshape(-1).tolist()
        if R_delta is not None:
            Rd = np.atleast_2d(np.asarray(R_delta, dtype=np.float64))
            c.R_delta[:self.da * self.da] = Rd.reshape(-1).tolist()
            c.has_R_delta = 1
            lu = np.zeros(self.da) if last_u is None else np.asarray(last_u, dtype=np.float64).reshape(-1)[:self.da]
            c.last_u[:self.da] = lu.tolist()
        xr = np.zeros(self.ds) if x_ref is None else np.asarray(x_ref, dtype=np.float64).reshape(-1)
        ur = np.zeros(self.da) if u_ref is None else np.asarray(u_ref, dtype=np.float64).reshape(-1)
        c.x_ref[:self.ds] = xr.tolist()
        c.u_ref[:self.da] = ur.tolist()
        self.c = c


class GPPack:
    """Device-resident state of ``ds`` GPs sharing X (reference: what Dynamics +
    GaussianProcessRegression hold, src/dynamics.py:33-37, src/gpr.py:24-36) folded into
    the per-data-update constants of the rollout (beta, weight matrices)."""

    def __init__(self, X, Y, Ky_inv, lambdas, sigma_f, device=None, y_is_beta=False):
        """Y: (N, ds) targets, or the beta vectors themselves when ``y_is_beta`` (then Ky_inv may be
        None: only means and cross-covariances are meaningful)."""
        self.device = device if device is not None else require_gpu()
        self._h = None
        self.generation = 0          # bumped by every (re)build: caches keyed on the pack object include it
        self._ws = {}
        self._graph_bufs = {}
        self.fullcov = False
        self._fill(X, Y, Ky_inv, lambdas, sigma_f, y_is_beta)

    def _fill(self, X, Y, Ky_inv, lambdas, sigma_f, y_is_beta):
        self.X = _dev(X, self.device)
        Y = _dev(Y, self.device)
        self.Y = Y.reshape(self.X.shape[0], -1)
        self.N, self.D = self.X.shape
        self.ds = self.Y.shape[1]
        self.da = self.D - self.ds
        ld = gstride = 0
        if Ky_inv is not None:
            Kt = Ky_inv if isinstance(Ky_inv, torch.Tensor) else None
            if (Kt is not None and Kt.is_cuda and Kt.device == self.device and Kt.dtype == torch.float64 and not Kt.is_contiguous()
                    and Kt.stride(-1) == 1 and Kt.shape[-2:] == (self.N, self.N) and (Kt.dim() == 2 or (Kt.dim() == 3 and Kt.shape[0] == self.ds))):
                # a view of a capacity-padded buffer (closed loop) and / or ONE matrix shared by all GPs: read in place
                ld, gstride = Kt.stride(-2), (0 if Kt.dim() == 2 else Kt.stride(0))
                Ky_inv = Kt
            elif Kt is not None and Kt.dim() == 2 and self.ds > 1 and tuple(Kt.shape) == (self.N, self.N):
                Ky_inv = _dev(Kt, self.device)          # one contiguous matrix shared by all GPs
                ld, gstride = self.N, 0
            else:
                Ky_inv = _dev(Ky_inv, self.device).reshape(self.ds, self.N, self.N)
        self.lambdas = np.ascontiguousarray(np.asarray(lambdas, dtype=np.float64).reshape(self.ds, self.D))
        self.sigma_f = np.ascontiguousarray(np.asarray(sigma_f, dtype=np.float64).reshape(self.ds))
        if self._h is None:
            h = ctypes.c_void_p()
            with torch.cuda.device(self.device):       # the pack's HBM buffers belong to THIS device, whatever is current
                check(lib().gpmpc_pack_create(ctypes.byref(h), self.N, self.ds, self.da), "gpmpc_pack_create")
            self._h = h
        _, lp = host_doubles(self.lambdas)
        _, sp = host_doubles(self.sigma_f)
        with torch.cuda.device(self.device):
            if ld and not y_is_beta:
                check(lib().gpmpc_pack_build_strided(self._h, ptr(self.X), ptr(self.Y), ctypes.c_void_p(Ky_inv.data_ptr()), ld, gstride,
                                                     lp, sp, stream_ptr()), "gpmpc_pack_build_strided")
            else:
                if ld:                                  # (beta given: the strided entry takes targets only)
                    Ky_inv = Ky_inv.contiguous() if Ky_inv.dim() == 3 else Ky_inv.contiguous().unsqueeze(0).expand(self.ds, -1, -1).contiguous()
                build = lib().gpmpc_pack_build_beta if y_is_beta else lib().gpmpc_pack_build
                check(build(self._h, ptr(self.X), ptr(self.Y), ptr(Ky_inv), lp, sp, stream_ptr()), "gpmpc_pack_build")
        n, npad, ds, da = (ctypes.c_int() for _ in range(4))
        lib().gpmpc_pack_dims(self._h, ctypes.byref(n), ctypes.byref(npad), ctypes.byref(ds), ctypes.byref(da))
        self.Np = npad.value
        self.generation += 1

    def rebuild(self, X, Y, Ky_inv, lambdas, sigma_f):
        """Refill THIS pack for new data / hyper-parameters when its allocation fits (same device, dimensions and padded
        size): no allocation, the library handle survives.  Returns False (pack untouched) when it does not fit."""
        Xs = X.shape
        n, D = int(Xs[0]), int(Xs[1])
        ds = int(Y.shape[1]) if len(Y.shape) > 1 else 1
        if self._h is None or D != self.D or ds != self.ds or ((n + 63) // 64) * 64 != self.Np:
            return False
        with torch.cuda.device(self.device):
            if n != self.N:
                check(lib().gpmpc_pack_resize(self._h, n), "gpmpc_pack_resize")
        self._fill(X, Y, Ky_inv, lambdas, sigma_f, False)      # (cross-covariance weights, if enabled, follow every build)
        return True

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                torch.cuda.synchronize(self.device)
                with torch.cuda.device(self.device):
                    lib().gpmpc_pack_destroy(h)
            except Exception:
                pass
            self._h = None

    def enable_fullcov(self):
        """Allocate the cross-covariance weight matrices (one N x N per GP pair): full-covariance rollout and
        analytic cross-covariance Jacobians."""
        with torch.cuda.device(self.device):
            check(lib().gpmpc_pack_enable_fullcov(self._h, stream_ptr()), "gpmpc_pack_enable_fullcov")
        self.fullcov = True
        return self

    def objective_gradient(self, x0_host, U_host, cost, want_grad=True):
        """One solver callback (C ABI ``gpmpc_objective_gradient``; reference src/mpc.py:202-255): host arrays in, host
        array out.  x0_host (ds,), U_host (H, da) float64 numpy -> numpy [cost, d cost / d U (H*da)].  Synchronous."""
        x0 = np.ascontiguousarray(x0_host, dtype=np.float64).reshape(-1)
        U = np.ascontiguousarray(U_host, dtype=np.float64).reshape(-1, self.da)
        if x0.shape[0] != self.ds or cost.ds != self.ds or cost.da != self.da:
            raise ValueError("shape mismatch between pack, x0, U and cost parameters")
        H = U.shape[0]
        out = np.empty(1 + (H * self.da if want_grad else 0), dtype=np.float64)
        dp = ctypes.POINTER(ctypes.c_double)
        with torch.cuda.device(self.device):
            check(lib().gpmpc_objective_gradient(self._h, H, x0.ctypes.data_as(dp), U.ctypes.data_as(dp), ctypes.byref(cost.c),
                                                 _lib.WANT_GRAD if want_grad else 0, out.ctypes.data_as(dp), stream_ptr(self.device)),
                  "gpmpc_objective_gradient")
        return out

    def reload_tuning(self):
        """Re-read the GPMPC_* tuning environment variables (read once at pack creation otherwise)."""
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            check(lib().gpmpc_pack_reload_tuning(self._h), "gpmpc_pack_reload_tuning")
        self._graph_bufs = {}
        self._ws = {}
        return self

    def plan(self, B, H, want_grad=True, graph=False):
        """What a rollout call of this shape launches (C ABI ``gpmpc_plan_describe``): dict with ``form``, ``kernel``, ``tiling``,
        ``workgroups`` (per horizon step), ``launches_per_step``, ``split`` ..."""
        buf = ctypes.create_string_buffer(512)
        flags = (_lib.WANT_GRAD if want_grad else 0) | (_lib.USE_GRAPH if graph else 0)
        check(lib().gpmpc_plan_describe(self._h, int(B), int(H), flags, buf, 512), "gpmpc_plan_describe")
        out = {}
        for kv in buf.value.decode().split():
            k, v = kv.split("=", 1)
            out[k] = int(v) if v.lstrip("-").isdigit() else v
        return out

    def plan_fullcov(self, B, H, want_grad=True):
        """What ``rollout_fullcov`` launches for this call shape (C ABI ``gpmpc_rollout_fullcov_describe``): dict with ``form``
        (``two_launch`` | ``four_launch``), ``tiling``, ``workgroups``, ``columns_per_iteration``, ``head_workgroups_per_unit``, ``kernel``."""
        buf = ctypes.create_string_buffer(512)
        check(lib().gpmpc_rollout_fullcov_describe(self._h, int(B), int(H), _lib.WANT_GRAD if want_grad else 0, buf, 512),
              "gpmpc_rollout_fullcov_describe")
        out = {}
        for kv in buf.value.decode().split():
            k, v = kv.split("=", 1)
            out[k] = int(v) if v.lstrip("-").isdigit() else v
        return out

    def autotune(self, B, H, want_grad=True, graph=False):
        """Time the candidate plans of this call shape on this device and keep the fastest for later calls (C ABI
        ``gpmpc_pack_autotune``).  Returns a list of dicts (``name``, ``ms``, ``winner``, plan fields), the default plan first."""
        buf = ctypes.create_string_buffer(8192)
        flags = (_lib.WANT_GRAD if want_grad else 0) | (_lib.USE_GRAPH if graph else 0)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            rc = lib().gpmpc_pack_autotune(self._h, int(B), int(H), flags, buf, 8192)
        if rc < 0:
            check(rc, "gpmpc_pack_autotune")
        self._graph_bufs = {}
        self._ws = {}
        out = []
        for item in buf.value.decode().split(";"):
            name, fields, ms = item.split(":")
            d = {"name": name.lstrip("*"), "winner": name.startswith("*"), "ms": float(ms)}
            for kv in fields.split(","):
                k, v = kv.split("=")
                d[k] = int(v)
            out.append(d)
        return out

    def autotune_clear(self):
        check(lib().gpmpc_pack_autotune_clear(self._h), "gpmpc_pack_autotune_clear")
        self._graph_bufs = {}
        self._ws = {}

    @property
    def shared_lambda(self):
        """True when every GP of the pack has bit-identical length-scales (the shared-lambda pair kernel applies)."""
        return lib().gpmpc_pack_shared_lambda(self._h) == 1

    @property
    def handle(self):
        return self._h

    def workspace(self, nbytes):
        """Scratch for one call, one buffer per stream: calls on different streams may overlap on the device (the C ABI is
        re-entrant across streams as long as the workspaces differ)."""
        key = torch.cuda.current_stream(self.device.index).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            if ws is None and len(self._ws) >= 8:            # bounded: drop the workspace of the longest-unused stream
                torch.cuda.synchronize(self.device)
                self._ws.pop(next(iter(self._ws)))
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        else:
            self._ws.pop(key)
        self._ws[key] = ws                                   # most recently used last
        return ws

    def beta(self):
        """(ds, N) copy of the cached beta vectors (for tests)."""
        out = torch.empty((self.ds, self.Np), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().gpmpc_pack_export(self._h, ptr(out), None, stream_ptr()), "gpmpc_pack_export")
        return out[:, :self.N]

    def weights(self):
        """(ds, Np, Np) copy of the folded weight matrices, element (i<=j) at [a, j, i] (for tests)."""
        out = torch.empty((self.ds, self.Np, self.Np), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().gpmpc_pack_export(self._h, None, ptr(out), stream_ptr()), "gpmpc_pack_export")
        return out


def rollout(pack, x0, U, cost, want_grad=True, want_traj=True, graph=False, precision="fp64"):
    """B shooting rollouts + cost (+ gradient) in one call (C ABI ``gpmpc_rollout``).

    x0: (B, ds) or (ds,); U: (B, H, da) or (H, da).  Returns a dict of CUDA tensors:
    cost (B,), grad (B, H, da), means (B, H+1, ds), vars (B, H+1, ds).

    graph=True replays the 2H+1 kernel launches as one hipGraph (for launch-latency-bound small batches, e.g. the
    B = 1 callbacks of a solver loop): inputs are copied into buffers owned by the pack and the returned tensors are
    views of buffers that the NEXT graph call with the same shape overwrites.

    precision: "fp64" (default), or -- objective only, for the tolerance sweep of BASELINE config 3 -- "fp32acc" (N^2
    products and sum in fp32) / "fp32" (exponent and exp in fp32 too).  Single precision fails the variance tolerance
    by orders of magnitude (profiles/r01/fp32_sweep.txt); it is a measurement aid, not a fast path."""
    dev = pack.device
    U_host = None
    if graph and not isinstance(U, torch.Tensor):          # solver callbacks hand numpy: stage through pinned memory
        U_host = np.ascontiguousarray(np.asarray(U, dtype=np.float64))
        if U_host.ndim == 2:
            U_host = U_host[None]
        B, H, da = U_host.shape
    else:
        U = _dev(U, dev)
        if U.dim() == 2:
            U = U.unsqueeze(0)
        B, H, da = U.shape
    x0 = _dev(x0, dev).reshape(-1, pack.ds)
    if x0.shape[0] == 1 and B > 1:
        x0 = x0.expand(B, pack.ds).contiguous()
    if da != pack.da or x0.shape[0] != B or cost.ds != pack.ds or cost.da != pack.da:
        raise ValueError("shape mismatch between pack, x0, U and cost parameters")
    prec = {"fp64": 0, "fp32acc": _lib.FP32_ACCUM, "fp32": _lib.FP32_ALL}[precision]
    if prec and want_grad:
        raise ValueError("the reduced-precision sweep modes are objective only: pass want_grad=False")
    flags = (_lib.WANT_GRAD if want_grad else 0) | (_lib.USE_GRAPH if graph else 0) | prec
    nbytes = lib().gpmpc_rollout_workspace_bytes(pack.handle, B, H, flags)
    e = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)  # noqa: E731
    if graph:
        key = (B, H, bool(want_grad), bool(want_traj))
        buf = pack._graph_bufs.get(key)
        if buf is None:
            # cost and grad are views of ONE block so that a caller can fetch both with a single device-to-host copy
            cg = e(B * (1 + (H * da if want_grad else 0)))
            buf = {"x0": e(B, pack.ds), "U": e(B, H, da), "cost_grad": cg, "cost": cg[:B],
                   "U_pinned": torch.empty((B, H, da), dtype=torch.float64).pin_memory(),
                   "ws": torch.empty(int(nbytes), dtype=torch.uint8, device=dev)}
            if want_grad:
                buf["grad"] = cg[B:].view(B, H, da)
            if want_traj:
                buf["means"], buf["vars"] = e(B, H + 1, pack.ds), e(B, H + 1, pack.ds)
            if len(pack._graph_bufs) >= 4:                   # the library keeps 4 captured shapes per pack
                pack._graph_bufs.pop(next(iter(pack._graph_bufs)))
            pack._graph_bufs[key] = buf
        if buf["ws"].numel() < nbytes:                          # the pack was refilled under a plan that needs more scratch
            buf["ws"] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)     # (new pointer: the library captures anew)
        buf["x0"].copy_(x0)
        if U_host is not None:
            # the previous copy out of the pinned buffer has completed: every graph call is followed by a synchronising
            # read of its results before the next one can be issued from the same host thread
            torch.cuda.current_stream(dev).synchronize()
            buf["U_pinned"].numpy()[...] = U_host
            buf["U"].copy_(buf["U_pinned"], non_blocking=True)
        else:
            buf["U"].copy_(U)
        x0, U, ws = buf["x0"], buf["U"], buf["ws"]
        out = {k: buf[k] for k in ("cost", "grad", "means", "vars", "cost_grad") if k in buf}
    else:
        out = {"cost": e(B)}
        if want_grad:
            out["grad"] = e(B, H, da)
        if want_traj:
            out["means"], out["vars"] = e(B, H + 1, pack.ds), e(B, H + 1, pack.ds)
        ws = pack.workspace(nbytes)
    with torch.cuda.device(dev):
        check(lib().gpmpc_rollout(pack.handle, B, H, ptr(x0), ptr(U), ctypes.byref(cost.c), flags,
                                  ptr(out.get("means")), ptr(out.get("vars")), ptr(out["cost"]), ptr(out.get("grad")),
                                  ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()), "gpmpc_rollout")
    return out


def rollout_fullcov(pack, x0, U, cost, want_grad=True):
    """Like :func:`rollout` but propagating the FULL state covariance (C ABI ``gpmpc_rollout_fullcov``;
    BASELINE config 5).  Returns cost (B,), grad (B,H,da), means (B,H+1,ds), covs (B,H+1,ds,ds)."""
    dev = pack.device
    if not pack.fullcov:
        pack.enable_fullcov()
    U = _dev(U, dev)
    if U.dim() == 2:
        U = U.unsqueeze(0)
    B, H, da = U.shape
    x0 = _dev(x0, dev).reshape(-1, pack.ds)
    if x0.shape[0] == 1 and B > 1:
        x0 = x0.expand(B, pack.ds).contiguous()
    if da != pack.da or x0.shape[0] != B or cost.ds != pack.ds or cost.da != pack.da:
        raise ValueError("shape mismatch between pack, x0, U and cost parameters")
    flags = _lib.WANT_GRAD if want_grad else 0
    e = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)  # noqa: E731
    out = {"cost": e(B), "means": e(B, H + 1, pack.ds), "covs": e(B, H + 1, pack.ds, pack.ds)}
    if want_grad:
        out["grad"] = e(B, H, da)
    nbytes = lib().gpmpc_rollout_fullcov_workspace_bytes(pack.handle, B, H, flags)
    ws = pack.workspace(nbytes)
    with torch.cuda.device(dev):
        check(lib().gpmpc_rollout_fullcov(pack.handle, B, H, ptr(x0), ptr(U), ctypes.byref(cost.c), flags,
                                          ptr(out["means"]), ptr(out["covs"]), ptr(out["cost"]), ptr(out.get("grad")),
                                          ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()),
              "gpmpc_rollout_fullcov")
    return out


def moment_match(pack, u, S, want_cov=False, want_grad=False, bug_compatible=False, want_l=False):
    """Exact moment matching of all ds GPs for nq Gaussian inputs N(u, S) with full S
    (C ABI ``gpmpc_moment_match``).  u: (nq, D) or (D,); S: (nq, D, D) or (D, D)."""
    dev = pack.device
    u = _dev(u, dev).reshape(-1, pack.D)
    nq = u.shape[0]
    S = _dev(S, dev).reshape(nq, pack.D, pack.D)
    flags = (_lib.WANT_GRAD if want_grad else 0) | (_lib.COV_BUG_COMPAT if bug_compatible else 0)
    ds, D = pack.ds, pack.D
    e = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)  # noqa: E731
    out = {"mean": e(nq, ds), "var": e(nq, ds)}
    if want_cov:
        out["cov"] = e(nq, ds, ds)
    if want_l:
        out["l"] = e(nq, ds, pack.N)
    if want_grad:
        out.update(dmean_du=e(nq, ds, D), dmean_dS=e(nq, ds, D, D), dvar_du=e(nq, ds, D), dvar_dS=e(nq, ds, D, D))
        if want_cov and ds > 1:
            # pair kernel's cross units when the pack has them (enable_fullcov) and the consistent form is asked for, else the direct
            # kernel (either form); diagonal (a, a) entries are not written: zero-filled
            out.update(dcov_du=torch.zeros((nq, ds, ds, D), dtype=torch.float64, device=dev),
                       dcov_dS=torch.zeros((nq, ds, ds, D, D), dtype=torch.float64, device=dev))
    nbytes = lib().gpmpc_moment_match_workspace_bytes(pack.handle, nq)
    ws = pack.workspace(nbytes)
    with torch.cuda.device(dev):
        check(lib().gpmpc_moment_match(pack.handle, nq, ptr(u), ptr(S), flags, ptr(out["mean"]), ptr(out["var"]),
                                       ptr(out.get("cov")), ptr(out.get("l")), ptr(out.get("dmean_du")), ptr(out.get("dmean_dS")),
                                       ptr(out.get("dvar_du")), ptr(out.get("dvar_dS")),
                                       ptr(out.get("dcov_du")), ptr(out.get("dcov_dS")),
                                       ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()), "gpmpc_moment_match")
    return out


def cost_full(cost, means, covs, U):
    """Risk-sensitive cost for given means (B,H+1,ds), FULL covariances (B,H+1,ds,ds), inputs (B,H,da)
    (C ABI ``gpmpc_cost``; reference src/mpc.py:156-200)."""
    dev = means.device if isinstance(means, torch.Tensor) and means.is_cuda else require_gpu()
    means, covs, U = _dev(means, dev), _dev(covs, dev), _dev(U, dev)
    if means.dim() == 2:
        means, covs, U = means.unsqueeze(0), covs.unsqueeze(0), U.unsqueeze(0)
    B, H1, ds = means.shape
    out = torch.empty(B, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib().gpmpc_cost(B, H1 - 1, ds, U.shape[2], ctypes.byref(cost.c), ptr(means), ptr(covs), ptr(U), ptr(out),
                               stream_ptr()), "gpmpc_cost")
    return out
