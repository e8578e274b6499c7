"""GaussianProcessRegression: host-side mirror of the reference class (src/gpr.py:5-370): training-data
state, hyper-parameter setters/getters, ``build_Ky_inv_mat``, the predict API and hyper-parameter
training (``compute_marginal_likelihood``, ``update_hyperparams``).  Arithmetic runs on the device
through the C ABI (``gpmpc_build_ky``, ``gpmpc_matvec``, ``gpmpc_predict``, ``gpmpc_ml_grad``,
``gpmpc_kinv_append``); the matrix inverse / Cholesky factor are torch's (rocSOLVER) on the device,
as in the reference (src/gpr.py:171).
"""
import ctypes

import numpy as np
import torch

from ._lib import check, host_doubles, lib, ptr, require_gpu, stream_ptr


def _exp_numpy(t):
    return torch.exp(t).numpy()


def _exp_item(t):
    return torch.exp(t).item()


def _exp_tuple(t):
    return tuple(float(v) for v in torch.exp(t).numpy())


def _noise_var_of(t):
    # src/gpr.py:170: the noise variance on the diagonal of Ky is float32(sigma_n^2)
    return float((torch.exp(t) ** 2 * torch.ones(1)).item())


class GaussianProcessRegression(object):
    def __init__(self, x_dim, nominal_model=None):
        self.device = require_gpu()
        self.x_dim = x_dim
        self.num_train = 0
        self.y_train = None
        self.X_train = None
        self.Kf = None
        self.Ky = None
        self.Ky_inv = None
        # log-hypers initialised to 0 => lambda = sigma_f = sigma_n = 1 (src/gpr.py:38-40)
        self.log_lambdas = torch.zeros(x_dim, device=self.device).type(torch.float64).requires_grad_()
        self.log_sigma_n = torch.tensor(0.0, device=self.device).type(torch.float64).requires_grad_()
        self.log_sigma_f = torch.tensor(0.0, device=self.device).type(torch.float64).requires_grad_()
        self.f_nom = nominal_model
        self.version = 0            # bumped whenever Ky_inv changes (Dynamics uses it to refresh its pack)
        self._beta = None
        self._adam = None           # (optimizer, host parameters) of update_hyperparams, created on first use
        self._hcache = {}           # host copies of the log-hypers, keyed on tensor identity and version
        self._vcache = {}           # values derived from them (exp, noise variance), same key
        self._built_hypers = None   # (lambdas, sigma_f, noise variance) the current Kf / Ky / Ky_inv were built with
        self._appends_since_rebuild = 0
        self.rebuild_every = 64     # incremental appends between two full rebuilds (bounds the accumulated round-off)
        # True: the periodic full rebuild runs on a SIDE stream while the appends go on (and the solver keeps using the
        # incrementally updated inverse); when it has finished, the observations appended meanwhile are re-applied to its
        # result (O(N^2) each) and the matrices are swapped -- the O(N^3) inverse never sits on an environment step.
        self.async_rebuild = False
        self._pending = None        # (n, hypers, Kf, Ky, Ky_inv, event, stream) of a rebuild in flight
        # How the incremental path bounds its round-off every `rebuild_every` appends: "rebuild" (default): Kf, Ky from scratch and
        # a fresh LU inverse, the reference's own update (O(N^3): 2.2 ms at N = 400, or `async_rebuild`); "newton": Newton-Schulz
        # steps X <- X + X (I - Ky X) on the incrementally updated inverse -- it is already within ~1e-4 of the true one, each step
        # squares the residual, and a step is two N x N GEMMs (a handful of launches instead of the LU's few hundred): the refresh
        # costs 0.1-0.2 ms on the step it falls on.  Falls back to the rebuild when the residual is not safely contractive.
        self.refresh = "rebuild"
        self.newton_accept = 1e-8   # measured Frobenius residual |I - Ky X| above which a Newton-Schulz refresh is discarded for a rebuild
        self.newton_residual_last = None
        # "lu": torch.linalg.inv, the reference's own call (src/gpr.py:171) and the default, so that Ky_inv carries the
        # reference's round-off.  "cholesky": potrf + potri (SURVEY 8 f1 as sketched): a third of the flops and a symmetric
        # result, but NOT the reference's numerics -- the variances move by ~1e-5 relative at sigma_n = 1e-5.
        self.inverse = "lu"

    # -- hyper-parameters: same expressions as the reference setters (src/gpr.py:51-88), including the
    #    dtype inference of torch.tensor (a Python float / list is float32 before the log).  Like the
    #    reference they do NOT rebuild the matrices.  The log / exp are evaluated on the HOST and the
    #    result moved to the device, so the values are those of the CPU reference the oracle is pinned
    #    to (float32 log differs by an ulp between host and device libm).
    def _log_param(self, value):
        self._adam = None           # the optimiser state belongs to the parameters it was created on
        return torch.log(torch.tensor(value)).type(torch.float64).to(self.device).requires_grad_()

    def _host(self, name):
        """Host copy of a log-hyper tensor, refreshed when the attribute is replaced or modified in place (the getters
        run on every solver callback through Dynamics.pack(); a device-to-host copy each time costs more than the
        rollout itself for small problems)."""
        t = getattr(self, name)
        c = self._hcache.get(name)
        if c is None or c[0] is not t or c[1] != t._version:     # the cache keeps t alive, so `is` cannot alias a new tensor
            c = (t, t._version, t.detach().cpu())
            self._hcache[name] = c
        return c[2]

    def _hval(self, name, fn):
        """`fn(host copy of the log tensor)`, evaluated once per value of the tensor (same key as `_host`): the closed loop reads the
        hyper-parameters ~15 times per environment step (up-to-date check of the incremental update, kernel arguments, pack key) and
        each read was a torch.exp + .item() / .numpy() on the host, 60-80 us per step together (profiles/r04/append_profile.txt)."""
        t = getattr(self, name)
        c = self._vcache.get((name, fn))
        if c is None or c[0] is not t or c[1] != t._version:
            c = (t, t._version, fn(self._host(name)))
            self._vcache[(name, fn)] = c
        return c[2]

    def set_lambdas(self, lambdas):
        self.log_lambdas = self._log_param(lambdas)

    def get_lambdas(self):
        return self._hval("log_lambdas", _exp_numpy).copy()

    def set_sigma_f(self, sigma_f):
        self.log_sigma_f = self._log_param(sigma_f)

    def get_sigma_f(self):
        return self._hval("log_sigma_f", _exp_item)

    def set_sigma_n(self, sigma_n):
        self.log_sigma_n = self._log_param(sigma_n)

    def get_sigma_n(self):
        return self._hval("log_sigma_n", _exp_item)

    # -- data
    def append_train_data(self, x, y, incremental=False):
        """x: (x_dim,) or (n, x_dim) numpy; y: scalar or (n,) numpy (src/gpr.py:90-122).
        incremental=True (extension, single observation, data already present): O(N^2) Schur-complement update of
        Ky_inv (C ABI ``gpmpc_kinv_append``) instead of the reference's O(N^3) rebuild."""
        if not np.isscalar(y):
            num_obs = len(y)
            y = np.asarray(y)[:, None]
        else:
            num_obs = 1
            y = np.array([y])[:, None]
        if num_obs == 1:
            x = np.reshape(x, (1, self.x_dim))
        mode = self._ingest(x, y, num_obs, incremental)
        if mode == "incremental":
            self._append_one_incremental(self.X_train[-1:])
        else:
            self.build_Ky_inv_mat()

    def _ingest(self, x, y, num_obs, incremental, shared=None, column=0):
        """Store the new rows (src/gpr.py:109-119) and say how the matrices have to follow: "incremental" or "full".
        shared (optional): a dict carried across the GPs of ONE Dynamics.append_train_data call, which feeds every GP the same
        input rows: the device copy of x, the concatenated X_train and the device copy of all targets (column `column` is this
        GP's) are then made once instead of once per GP (each was its own host-to-device copy on every environment step)."""
        if incremental and num_obs == 1 and self.num_train > 0 and self.x_dim * 8 <= 512:
            return self._ingest_one_in_place(x, y, shared, column)
        if shared is not None and "x" in shared:
            x = shared["x"]
        else:
            x = torch.tensor(np.asarray(x), requires_grad=False).type(torch.float64).to(self.device)
            if shared is not None:
                shared["x"] = x
        if shared is not None and ("y_all" in shared or "y_all_host" in shared):
            if "y_all" not in shared:
                shared["y_all"] = torch.tensor(shared["y_all_host"]).to(self.device)
            y = shared["y_all"][:, column:column + 1].contiguous()
        else:
            y = torch.tensor(y, requires_grad=False).type(torch.float64).to(self.device)
        if self.num_train == 0:
            self.X_train, self.y_train = x, y
        else:
            if shared is not None and shared.get("X_old") is self.X_train:
                self.X_train = shared["X_new"]                     # the same tensor for every GP that held the same one
            else:
                X_new = torch.cat((self.X_train, x), dim=0)
                if shared is not None:
                    shared["X_old"], shared["X_new"] = self.X_train, X_new
                self.X_train = X_new
            self.y_train = torch.cat((self.y_train, y), dim=0)
        # The O(N^2) append is only valid on matrices built with the CURRENT hyper-parameters (the setters do not
        # rebuild, src/gpr.py:53; the reference's append always does, so an edit takes effect there), and its round-off
        # accumulates: fall back to the reference's full rebuild when the hypers changed and every `rebuild_every` appends.
        inc = (incremental and num_obs == 1 and self.num_train > 0 and self.Ky_inv is not None
               and self._built_hypers == self._current_hypers()
               and (self._appends_since_rebuild < self.rebuild_every or self.async_rebuild or self.refresh == "newton"))
        if not inc:
            self._pending = None                               # a full rebuild supersedes one in flight
        self.num_train += num_obs
        return "incremental" if inc else "full"

    def _ingest_one_in_place(self, x, y, shared, column):
        """One new observation on the incremental path: X_train / y_train are views of capacity-padded, APPEND-ONLY device buffers (rows
        below num_train never change, so earlier views stay valid), and the new row travels as a kernel argument (C ABI
        ``gpmpc_store_host``) -- no host-to-device copy, no concatenation.  Same values as src/gpr.py:109-119."""
        n = self.num_train
        xh = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(self.x_dim))
        yh = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(1))
        vp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 8 * off)  # noqa: E731
        with torch.cuda.device(self.device):
            sp = stream_ptr(self.device)
            if shared is not None and shared.get("X_old") is self.X_train:
                self.X_train = shared["X_new"]                     # the leader of this call already appended the row
            else:
                buf = getattr(self, "_Xbuf", None)
                if buf is None or buf.shape[0] < n + 1 or self.X_train.untyped_storage().data_ptr() != buf.untyped_storage().data_ptr():
                    buf = torch.empty((((n + 1 + 255) // 256) * 256 + 256, self.x_dim), dtype=torch.float64, device=self.device)
                    buf[:n].copy_(self.X_train)
                    self._Xbuf = buf
                check(lib().gpmpc_store_host(vp(buf, n * self.x_dim), xh.ctypes.data_as(ctypes.c_void_p), 8 * self.x_dim, sp), "gpmpc_store_host")
                X_new = buf[:n + 1]
                if shared is not None:
                    shared["X_old"], shared["X_new"] = self.X_train, X_new
                self.X_train = X_new
            yb = getattr(self, "_ybuf", None)
            if yb is None or yb.shape[0] < n + 1 or self.y_train.untyped_storage().data_ptr() != yb.untyped_storage().data_ptr():
                yb = torch.empty((((n + 1 + 255) // 256) * 256 + 256, 1), dtype=torch.float64, device=self.device)
                yb[:n].copy_(self.y_train)
                self._ybuf = yb
            check(lib().gpmpc_store_host(vp(yb, n), yh.ctypes.data_as(ctypes.c_void_p), 8, sp), "gpmpc_store_host")
            self.y_train = yb[:n + 1]
        inc = (self.Ky_inv is not None and self._built_hypers == self._current_hypers()
               and (self._appends_since_rebuild < self.rebuild_every or self.async_rebuild or self.refresh == "newton"))
        if not inc:
            self._pending = None
        self.num_train += 1
        return "incremental" if inc else "full"

    def _adopt(self, other):
        """Share the matrices of `other`, a GP with bit-identical training inputs and hyper-parameters.  Invariant for sharers
        (src/gpr.py:159-171 assigns fresh tensors on every update): a matrix a GP holds is never modified under it.  The
        incremental append writes into two REUSED capacity-padded buffer sets, so the leader keeps weak references to the GPs
        that hold views of them and, before it overwrites a set that one of them still aliases -- a leader fed on its own through
        the public ``append_train_data`` instead of ``update_many`` --, gives that GP its own copies
        (`_append_one_incremental`); user code that keeps ``gp.Ky_inv`` across TWO later appends must clone it itself."""
        import weakref
        self.Kf, self.Ky, self.Ky_inv = other.Kf, other.Ky, other.Ky_inv
        sh = other.__dict__.setdefault("_sharers", weakref.WeakSet())
        sh.add(self)
        self._beta = None
        self.version += 1
        self._built_hypers = other._built_hypers
        self._appends_since_rebuild = other._appends_since_rebuild
        self._pending = None

    def _current_hypers(self):
        return (self._hval("log_lambdas", _exp_tuple), float(self.get_sigma_f()), self._noise_var())

    def se_kernel(self, x1, x2):
        """sigma_f^2 exp(-1/2 (x1 - x2)^T Lambda^-1 (x1 - x2)) for two points: 0-dim device tensor (src/gpr.py:124-135),
        evaluated by the same device kernel as K(X*, X) (C ABI ``gpmpc_predict``)."""
        a = torch.as_tensor(x1).to(self.device).type(torch.float64).reshape(1, self.x_dim).contiguous()
        b = torch.as_tensor(x2).to(self.device).type(torch.float64).reshape(1, self.x_dim).contiguous()
        out = torch.empty((1, 1), dtype=torch.float64, device=self.device)
        _, lp = host_doubles(self.get_lambdas())
        nb = lib().gpmpc_predict_workspace_bytes(1, self.x_dim, 1)
        ws = torch.empty(max(int(nb), 8), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().gpmpc_predict(1, self.x_dim, ptr(b), lp, self.get_sigma_f(), None, None, 0.0, 1, ptr(a), ptr(out), None, None,
                                      ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()), "gpmpc_predict")
        return out.reshape(())

    def _schur_append(self, X_old, x_new, Kf, Ky, Kinv):
        """(Kf, Ky, Ky_inv) of the n points X_old -> the same for X_old + x_new: one O(N^2) Schur-complement step
        (C ABI ``gpmpc_kinv_append``), on the current stream."""
        n = X_old.shape[0]
        X_old = X_old.contiguous()
        sigma_f = self.get_sigma_f()
        noise = self._noise_var()
        _, lp = host_doubles(self.get_lambdas())
        k = torch.empty((1, n), dtype=torch.float64, device=self.device)
        xn = x_new.reshape(1, self.x_dim).contiguous()
        nb = lib().gpmpc_predict_workspace_bytes(n, self.x_dim, 1)
        ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        kinv_old = Kinv.contiguous()
        out = torch.empty((n + 1, n + 1), dtype=torch.float64, device=self.device)
        nb2 = lib().gpmpc_kinv_append_workspace_bytes(n)
        ws2 = torch.empty(nb2, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().gpmpc_predict(n, self.x_dim, ptr(X_old), lp, sigma_f, None, None, 0.0, 1, ptr(xn), ptr(k), None, None,
                                      ctypes.c_void_p(ws.data_ptr()), nb, stream_ptr(self.device)), "gpmpc_predict")
            check(lib().gpmpc_kinv_append(n, ptr(kinv_old), ptr(k), sigma_f ** 2 + noise, ptr(out),
                                          ctypes.c_void_p(ws2.data_ptr()), nb2, stream_ptr(self.device)), "gpmpc_kinv_append")
        kff = torch.full((1, 1), sigma_f ** 2, dtype=torch.float64, device=self.device)
        Kf2 = torch.cat((torch.cat((Kf, k.t()), dim=1), torch.cat((k, kff), dim=1)), dim=0)
        Ky2 = torch.cat((torch.cat((Ky, k.t()), dim=1), torch.cat((k, kff + noise), dim=1)), dim=0)
        return Kf2, Ky2, out

    def _append_buffers(self, n1):
        """Two sets of capacity-padded (cap, cap) buffers for Kf, Ky, Ky_inv: an append reads the current matrices (views of one set, or
        the packed tensors of a full rebuild) and writes the (n + 1)-point ones into the OTHER set -- no allocation and no
        concatenation per step, and the n-point tensors stay intact for whoever still holds them (followers of update_many, a pack
        being built) until the append after next.  Grown in steps of 256 rows."""
        cap = getattr(self, "_cap", 0)
        if cap < n1:
            cap = ((n1 + 64 + 255) // 256) * 256
            self._cap = cap
            self._bufs = [[torch.empty((cap, cap), dtype=torch.float64, device=self.device) for _ in range(3)] for _ in range(2)]
            self._cur = 1                                       # the next write goes to set 0
            nb = lib().gpmpc_gp_append_workspace_bytes(cap, self.x_dim)
            self._append_ws = torch.empty(int(nb), dtype=torch.uint8, device=self.device)
        return cap

    def _append_one_incremental(self, x_new):
        """self.X_train / y_train already hold the new row (last) and num_train counts it; Kf, Ky, Ky_inv still have the
        old size n.  One library call (C ABI ``gpmpc_gp_append``: k = K_f(X, x_new), the Schur step on Ky_inv, the new row /
        column of Kf and Ky) from the current matrices into the other buffer set."""
        n = self.num_train - 1
        cap = self._append_buffers(n + 1)
        Kf, Ky, Kinv = self.Kf, self.Ky, self.Ky_inv
        if not (Kf.stride(1) == 1 and Ky.stride(1) == 1 and Kf.stride(0) == Ky.stride(0)):
            Kf, Ky = Kf.contiguous(), Ky.contiguous()
        if Kinv.stride(1) != 1:
            Kinv = Kinv.contiguous()
        dst = self._bufs[1 - self._cur]
        # never write over the set the inputs live in (after a rebuild / refresh the inputs are packed tensors: either set is free)
        base = [t.untyped_storage().data_ptr() for t in (Kf, Ky, Kinv)]
        if any(d.untyped_storage().data_ptr() in base for d in dst):
            dst = self._bufs[self._cur]
            if any(d.untyped_storage().data_ptr() in base for d in dst):
                raise RuntimeError("incremental append: both buffer sets alias the current matrices")
            self._cur = 1 - self._cur
        # a GP that adopted views of the set about to be overwritten (and was not re-fed together with this one) keeps its
        # n-point matrices: it gets copies first
        targets = {d.untyped_storage().data_ptr() for d in dst}
        for f in list(getattr(self, "_sharers", ())):
            if f is self:
                continue
            for name in ("Kf", "Ky", "Ky_inv"):
                t = getattr(f, name)
                if t is not None and t.untyped_storage().data_ptr() in targets:
                    setattr(f, name, t.clone())
        X_old = self.X_train[:n]
        if not X_old.is_contiguous():
            X_old = X_old.contiguous()
        xn = x_new.reshape(1, self.x_dim).contiguous()
        _, lp = host_doubles(self.get_lambdas())
        vp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(self.device):
            check(lib().gpmpc_gp_append(n, self.x_dim, ptr(X_old), ptr(xn), lp, self.get_sigma_f(), self._noise_var(),
                                        vp(Kf), vp(Ky), Kf.stride(0), vp(Kinv), Kinv.stride(0),
                                        vp(dst[0]), vp(dst[1]), vp(dst[2]), cap,
                                        ctypes.c_void_p(self._append_ws.data_ptr()), self._append_ws.numel(), stream_ptr(self.device)),
                  "gpmpc_gp_append")
        self._cur = 1 - self._cur
        self.Kf, self.Ky, self.Ky_inv = dst[0][:n + 1, :n + 1], dst[1][:n + 1, :n + 1], dst[2][:n + 1, :n + 1]
        self._beta = None
        self.version += 1
        self._appends_since_rebuild += 1
        if self.refresh == "newton":
            if self._appends_since_rebuild >= self.rebuild_every:
                self._newton_refresh()
        elif self.async_rebuild:
            self._service_async_rebuild()

    def _newton_refresh(self, max_steps=6):
        """Newton-Schulz polish of the incrementally updated inverse (see `refresh`).  The residual R = I - Ky X is measured once
        (one small device-to-host read); below 0.5 in the Frobenius norm the iteration X <- X + X R converges quadratically and is
        run until the PREDICTED residual (r, r^2, r^4, ...) is below 1e-13; otherwise the matrices are rebuilt from scratch."""
        n = self.num_train
        eye = torch.eye(n, dtype=torch.float64, device=self.device)
        X = self.Ky_inv
        R = eye - self.Ky @ X
        r = float(torch.linalg.matrix_norm(R).item())
        if not (r < 0.5):
            self.build_Ky_inv_mat()
            return
        steps = 0
        while r > 1e-13 and steps < max_steps:
            X = X + X @ R
            steps += 1
            r = r * r
            if r > 1e-13 and steps < max_steps:
                R = eye - self.Ky @ X
        # the loop runs on the PREDICTED residual (r, r^2, r^4, ...): measure the final one once -- an iteration that stalled in
        # round-off (cond(Ky) ~ 1e7 and beyond) must not be accepted and the refresh counter reset on a prediction
        if steps:
            r_final = float(torch.linalg.matrix_norm(eye - self.Ky @ X).item())
            self.newton_residual_last = r_final
            if not (r_final < self.newton_accept):
                self.build_Ky_inv_mat()
                self.newton_steps_last = -1
                return
        self.Ky_inv = X
        self._beta = None
        self.version += 1
        self._appends_since_rebuild = 0
        self.newton_steps_last = steps

    def _service_async_rebuild(self):
        """Side-stream rebuild: start one when `rebuild_every` appends have accumulated; once a started one has finished,
        bring its result up to date with the observations appended since and swap it in."""
        if self._pending is not None:
            n_p, hyp, Kf, Ky, Kinv, ev, side = self._pending
            if hyp != self._current_hypers():
                self._pending = None
            elif ev.query():
                main = torch.cuda.current_stream(self.device.index)
                for t in (Kf, Ky, Kinv):
                    t.record_stream(main)                          # allocated on the side stream, consumed here
                # the observations that arrived while it ran, one Schur step each.  (ONE block step for all of them -- S = K_kk + sn^2 I -
                # K_kn Ky_inv K_nk, k x k -- was tried: its result is as close to the true inverse as the sequential one, but the appends
                # that follow it blow up within a few steps at cond(Ky) ~ 1e7: abs error 1e-5 -> 1e+3 in ten appends, against a flat
                # ~1e-4 for the sequential form, on the problem of tests/test_gpu_api.py::test_side_stream_rebuild_catches_up_and_swaps.)
                for m in range(n_p, self.num_train):
                    Kf, Ky, Kinv = self._schur_append(self.X_train[:m], self.X_train[m:m + 1], Kf, Ky, Kinv)
                self.Kf, self.Ky, self.Ky_inv = Kf, Ky, Kinv
                self._beta = None
                self.version += 1
                self._appends_since_rebuild = self.num_train - n_p
                self._pending = None
            return
        if self._appends_since_rebuild < self.rebuild_every:
            return
        n = self.num_train
        X = self.X_train.contiguous()                              # immutable: later appends make new tensors
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device.index)
        side.wait_stream(main)                                     # X (and the hyper-parameter copies) are ready
        _, lp = host_doubles(self.get_lambdas())
        with torch.cuda.stream(side), torch.cuda.device(self.device):
            Kf = torch.empty((n, n), dtype=torch.float64, device=self.device)
            Ky = torch.empty((n, n), dtype=torch.float64, device=self.device)
            check(lib().gpmpc_build_ky(n, self.x_dim, ptr(X), lp, self.get_sigma_f(), self._noise_var(), ptr(Kf), ptr(Ky),
                                       ctypes.c_void_p(side.cuda_stream)), "gpmpc_build_ky")
            Kinv = self._invert(Ky, check=False)
            ev = torch.cuda.Event()
            ev.record(side)
        X.record_stream(side)
        self._pending = (n, self._current_hypers(), Kf, Ky, Kinv, ev, side)

    def finish_async_rebuild(self):
        """Wait for a side-stream rebuild in flight and swap it in (tests; not needed in the loop)."""
        if self._pending is not None:
            self._pending[5].synchronize()
            self._service_async_rebuild()

    def update_Ky_inv_mat(self, k_new):
        """Ky_inv of the n current points -> Ky_inv of the n + 1 points whose last one has the covariance column ``k_new``
        (n, 1) with the others (src/gpr.py:137-157: block inverse with the scalar Schur complement
        ``sigma_n^2 + sigma_f^2 - k^T Ky_inv k``; the reference marks it "don't use" and never calls it).  Like the
        reference's it only replaces ``Ky_inv``; here the update runs on the device (C ABI ``gpmpc_kinv_append``)."""
        n = self.Ky_inv.shape[0]
        k = torch.as_tensor(k_new).detach().to(self.device, torch.float64).reshape(1, n).contiguous()
        kinv_old = self.Ky_inv.detach().contiguous()
        out = torch.empty((n + 1, n + 1), dtype=torch.float64, device=self.device)
        nb = lib().gpmpc_kinv_append_workspace_bytes(n)
        ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        kappa = self.get_sigma_n() ** 2 + self.get_sigma_f() ** 2
        with torch.cuda.device(self.device):
            check(lib().gpmpc_kinv_append(n, ptr(kinv_old), ptr(k), kappa, ptr(out), ctypes.c_void_p(ws.data_ptr()), nb,
                                          stream_ptr(self.device)), "gpmpc_kinv_append")
        self.Ky_inv = out
        self._beta = None
        self.version += 1

    def build_Ky_inv_mat(self):
        """Kf, Ky, Ky_inv from scratch (src/gpr.py:159-171)."""
        self._build_kf_ky()
        self._finish_build(self._invert(self.Ky))

    def _build_kf_ky(self):
        n = self.num_train
        X = self.X_train.contiguous()
        _, lp = host_doubles(self.get_lambdas())
        # src/gpr.py:170: sigma_n**2 (0-dim float64) * torch.eye (float32) is a float32 tensor
        noise = self._noise_var()
        self.Kf = torch.empty((n, n), dtype=torch.float64, device=self.device)
        self.Ky = torch.empty((n, n), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().gpmpc_build_ky(n, self.x_dim, ptr(X), lp, self.get_sigma_f(), noise, ptr(self.Kf), ptr(self.Ky),
                                       stream_ptr()), "gpmpc_build_ky")

    def _invert(self, Ky, check=True):
        """Explicit inverse of one matrix or of a (k, n, n) stack (one batched factorisation).  check=False: no error check on
        the host (torch.linalg.inv waits for the factorisation's status word: a host synchronisation, which would put a
        side-stream rebuild back onto the step that starts it)."""
        if self.inverse == "lu":
            return torch.linalg.inv(Ky) if check else torch.linalg.inv_ex(Ky, check_errors=False).inverse
        if self.inverse == "cholesky":
            if not check:
                return torch.cholesky_inverse(torch.linalg.cholesky_ex(Ky, check_errors=False).L)
            return torch.cholesky_inverse(torch.linalg.cholesky(Ky))
        raise ValueError("GaussianProcessRegression.inverse must be 'lu' or 'cholesky', got %r" % (self.inverse,))

    def _finish_build(self, Ky_inv):
        self.Ky_inv = Ky_inv
        self._beta = None
        self.version += 1
        self._built_hypers = self._current_hypers()
        self._appends_since_rebuild = 0

    @staticmethod
    def update_many(gps, modes):
        """Bring the matrices of several GPs that were fed the SAME inputs (Dynamics.append_train_data) up to date:
        GPs with bit-identical hyper-parameters share ONE set of matrices (every experiment of the reference sets the
        same lambda / sigma_f / sigma_n on all GPs: ds identical O(N^3) inversions per Simulator step there,
        src/simulator.py:55, src/gpr.py:171), and the distinct ones that need a full rebuild are inverted as one batched
        factorisation of the (k, n, n) stack instead of k in a Python loop."""
        leaders = {}
        for g, mode in zip(gps, modes):
            # followers take over the leader's matrices AND its refresh state (_adopt): a group must agree on the refresh policy too
            leaders.setdefault((g._current_hypers(), g.inverse, mode, g.num_train, g.refresh, bool(g.async_rebuild), g.rebuild_every,
                                g._appends_since_rebuild), []).append(g)
        full = [grp[0] for key, grp in leaders.items() if key[2] == "full"]
        for key, grp in leaders.items():
            if key[2] == "incremental":
                grp[0]._append_one_incremental(grp[0].X_train[-1:])
        by_kind = {}
        for g in full:
            g._build_kf_ky()
            by_kind.setdefault((g.inverse, g.num_train), []).append(g)
        for grp in by_kind.values():
            if len(grp) == 1:
                grp[0]._finish_build(grp[0]._invert(grp[0].Ky))
            else:
                inv = grp[0]._invert(torch.stack([g.Ky for g in grp]))
                for k, g in enumerate(grp):
                    g._finish_build(inv[k])
        for grp in leaders.values():
            for g in grp[1:]:
                g._adopt(grp[0])

    # -- prediction
    def _targets(self):
        y = self.y_train
        if self.f_nom is not None:
            y = y - self.f_nom(self.X_train)
        return y.reshape(-1).contiguous()

    def beta(self):
        """Ky_inv (y - f_nom(X)), cached until the matrices change."""
        if self._beta is None:
            out = torch.empty(self.num_train, dtype=torch.float64, device=self.device)
            # keep every operand referenced until the launch returned: a temporary freed while the
            # argument list is still being built can be handed out again by the caching allocator
            kinv, tgt = self.Ky_inv.contiguous(), self._targets()
            with torch.cuda.device(self.device):
                check(lib().gpmpc_matvec(self.num_train, self.num_train, ptr(kinv), ptr(tgt), ptr(out), stream_ptr()),
                      "gpmpc_matvec")
            self._beta = out
        return self._beta

    def _predict(self, X_pred, want_mean, want_cov, noise_var):
        Xp = torch.tensor(np.asarray(X_pred), device=self.device).type(torch.float64)
        single = Xp.dim() == 1
        Xp = Xp.reshape(-1, self.x_dim).contiguous()
        p, n = Xp.shape[0], self.num_train
        K = torch.empty((p, n), dtype=torch.float64, device=self.device)
        mean = torch.empty(p, dtype=torch.float64, device=self.device) if want_mean else None
        cov = torch.empty((p, p), dtype=torch.float64, device=self.device) if want_cov else None
        _, lp = host_doubles(self.get_lambdas())
        nb = lib().gpmpc_predict_workspace_bytes(n, self.x_dim, p)
        ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        Xt = self.X_train.contiguous()
        beta = self.beta() if want_mean else None
        kinv = self.Ky_inv.contiguous() if want_cov else None
        with torch.cuda.device(self.device):
            check(lib().gpmpc_predict(n, self.x_dim, ptr(Xt), lp, self.get_sigma_f(), ptr(beta), ptr(kinv), noise_var, p, ptr(Xp),
                                      ptr(K), ptr(mean), ptr(cov), ctypes.c_void_p(ws.data_ptr()), nb, stream_ptr()),
                  "gpmpc_predict")
        return K, mean, cov, Xp, single

    def compute_pred_train_covariance(self, X_pred):
        """K(X*, X): (p, N) tensor, or (N,) for a single 1-D test point (src/gpr.py:253-283)."""
        K, _, _, _, single = self._predict(X_pred, False, False, 0.0)
        return K[0] if single else K

    def predict_latent_vars(self, X_pred, covar=False, targets=False):
        """Posterior mean (p,1) and optionally covariance (p,p), numpy (src/gpr.py:285-332)."""
        noise = self.get_sigma_n() ** 2 if (covar and targets) else 0.0
        _, mean, cov, Xp, _ = self._predict(X_pred, True, covar, noise)
        f = mean.reshape(-1, 1)
        if self.f_nom is not None:
            f = f + self.f_nom(Xp).reshape(-1, 1)
        if not covar:
            return f.cpu().detach().numpy(), None
        return f.cpu().detach().numpy(), cov.cpu().detach().numpy()

    # -- hyper-parameter training (src/gpr.py:173-251, 334-370)
    def _noise_var(self):
        # src/gpr.py:170: the noise variance on the diagonal of Ky is float32(sigma_n^2)
        return self._hval("log_sigma_n", _noise_var_of)

    def _ml_terms(self):
        """One device pass (C ABI ``gpmpc_ml_grad``): [d ml/d log lambda (D), d/d log sigma_f, d/d log sigma_n, r^T alpha]."""
        n, D = self.num_train, self.x_dim
        X, kinv, resid, alpha = self.X_train.contiguous(), self.Ky_inv.contiguous(), self._targets(), self.beta()
        _, lp = host_doubles(self.get_lambdas())
        out = torch.empty(D + 3, dtype=torch.float64, device=self.device)
        nb = lib().gpmpc_ml_grad_workspace_bytes(n, D)
        ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().gpmpc_ml_grad(n, D, ptr(X), ptr(kinv), ptr(alpha), ptr(resid), lp, self.get_sigma_f(), self._noise_var(),
                                      ptr(out), ctypes.c_void_p(ws.data_ptr()), nb, stream_ptr()), "gpmpc_ml_grad")
        return out

    def _logdet_Ky(self):
        """log det(Ky).  The reference takes log(det(Ky)) (src/gpr.py:245), which over/underflows for large N; the
        Cholesky form is the same number whenever that one is finite."""
        L, info = torch.linalg.cholesky_ex(self.Ky)
        if int(info.item()) == 0:
            return 2.0 * torch.log(torch.diagonal(L)).sum()
        sign, logabs = torch.linalg.slogdet(self.Ky)
        return logabs if sign.item() > 0 else torch.tensor(float("nan"), dtype=torch.float64, device=self.device)

    def compute_marginal_likelihood(self):
        """-1/2 r^T Ky_inv r - 1/2 log det Ky - N/2 log(2 pi), r = y - f_nom(X): (1,1) device tensor
        (src/gpr.py:240-251; no autograd graph -- the gradient is ``marginal_likelihood_gradient``)."""
        quad = self._ml_terms()[self.x_dim + 2]
        ml = -0.5 * quad - 0.5 * self._logdet_Ky() - self.num_train / 2 * np.log(2 * np.pi)
        return ml.reshape(1, 1)

    def marginal_likelihood_gradient(self):
        """Gradient of the likelihood w.r.t. the log hyper-parameters, as numpy: what ``.grad`` of log_lambdas /
        log_sigma_f / log_sigma_n holds after ``ml.backward()`` in the reference (src/gpr.py:337-338); also the
        'log_*' entries of marginal_likelihood_grad (src/gpr.py:200-238)."""
        g = self._ml_terms().cpu().numpy()
        D = self.x_dim
        return {"log_lambda": g[:D].copy(), "log_sigma_f": float(g[D]), "log_sigma_n": float(g[D + 1])}

    def kernel_matrix_gradient(self):
        """Dense dKy/dlambda_k (N,N,D), dKy/dsigma_f, dKy/dsigma_n as in src/gpr.py:173-198 (device tensors).  API
        compatibility only: the training path never materialises them."""
        lam = torch.exp(self.log_lambdas.detach())
        sf, sn = torch.exp(self.log_sigma_f.detach()), torch.exp(self.log_sigma_n.detach())
        diff2 = torch.square(self.X_train[:, None, :] - self.X_train[None, :, :])
        return {"lambda": self.Kf[:, :, None] * diff2 / (2 * lam ** 2), "sigma_f": 2 / sf * self.Kf,
                "sigma_n": 2 * sn * torch.eye(self.num_train, dtype=torch.float64, device=self.device)}

    def marginal_likelihood_grad(self, gradient_dict=None):
        """Same keys as src/gpr.py:200-238; computed from the one-pass kernel (``gradient_dict`` is accepted and ignored)."""
        g = self.marginal_likelihood_gradient()
        lam, sf, sn = self.get_lambdas(), self.get_sigma_f(), self.get_sigma_n()
        dev = lambda v: torch.as_tensor(v, dtype=torch.float64, device=self.device)   # noqa: E731
        return {"lambda": dev(g["log_lambda"] / lam), "sigma_f": dev(g["log_sigma_f"] / sf), "sigma_n": dev(g["log_sigma_n"] / sn),
                "log_lambda": dev(g["log_lambda"]), "log_sigma_f": dev(g["log_sigma_f"]), "log_sigma_n": dev(g["log_sigma_n"])}

    def update_hyperparams(self, num_iters=1000, verbose=False):
        """Maximise the marginal likelihood with Adam(lr=0.1, betas=(0.9, 0.999)) over [log_lambdas, log_sigma_n,
        log_sigma_f]; per iteration: gradient at the current matrices, Adam step, rebuild; stops when every
        |gradient| < 1e-5 (src/gpr.py:46-49, 334-370).  The gradient is the analytic trace form evaluated by
        ``gpmpc_ml_grad`` instead of autograd through inv / det.  Differences from the reference: nothing is
        printed unless ``verbose``; the optimiser follows hyper-parameters changed through the setters (in the
        reference the setters replace the tensors the optimiser was built on, so training after a setter call
        silently updates nothing).  Returns the list of per-iteration records."""
        if self._adam is None:
            host = [self.log_lambdas.detach().cpu().clone().requires_grad_(),
                    self.log_sigma_n.detach().cpu().clone().requires_grad_(),
                    self.log_sigma_f.detach().cpu().clone().requires_grad_()]
            adam = torch.optim.Adam(params=host, lr=0.1, betas=(0.9, 0.999), maximize=True)
            self._adam = (adam, host)
        adam, host = self._adam
        history = []
        for it in range(num_iters):
            g = self._ml_terms().cpu()
            D = self.x_dim
            ml = (-0.5 * g[D + 2] - 0.5 * self._logdet_Ky().cpu() - self.num_train / 2 * np.log(2 * np.pi)).item()
            host[0].grad = g[:D].clone()
            host[1].grad = g[D + 1].clone().reshape(())
            host[2].grad = g[D].clone().reshape(())
            adam.step()
            self.log_lambdas = host[0].detach().clone().to(self.device).requires_grad_()
            self.log_sigma_n = host[1].detach().clone().to(self.device).requires_grad_()
            self.log_sigma_f = host[2].detach().clone().to(self.device).requires_grad_()
            self.build_Ky_inv_mat()
            rec = {"ml": ml, "grad": {"log_lambdas": g[:D].numpy().copy(), "log_sigma_f": g[D].item(), "log_sigma_n": g[D + 1].item()},
                   "log_lambdas": host[0].detach().numpy().copy(), "log_sigma_f": host[2].item(), "log_sigma_n": host[1].item()}
            history.append(rec)
            if verbose:
                print("Iter: ", it, " ml: ", ml, " lambdas: ", self.get_lambdas(), " sigma_f: ", self.get_sigma_f(),
                      " sigma_n: ", self.get_sigma_n())
            gr = rec["grad"]
            if (np.abs(gr["log_lambdas"]) < 1e-5).all() and abs(gr["log_sigma_f"]) < 1e-5 and abs(gr["log_sigma_n"]) < 1e-5:
                break
        return history
