"""RiskSensitiveMPC: host-side mirror of the reference class (src/mpc.py:7-330).

The cyipopt problem-object surface is kept (``objective`` / ``gradient`` / ``constraints`` /
``jacobian``), as are the constructor, the setters and ``get_optimal_trajectory``.  Differences, all
forced by moving the rollout into one fused HIP call:

* ``objective(x)`` evaluates cost AND gradient in the same device pass and caches both keyed on the
  bytes of ``x`` (the reference keeps an autograd graph and back-propagates lazily, src/mpc.py:225-255,
  and its ``gradient`` ignores ``x``); Ipopt's buffer is copied, never aliased (src/mpc.py:217).
* ``objective_batch`` / ``evaluate_batch`` evaluate many candidate action sequences at once (the
  trajectory-sharded batch the multi-GPU path consumes) -- an extension, the reference has no batch API.
* ``gamma == 0`` is accepted and means the analytic risk-neutral limit.
* ``get_optimal_trajectory(curr_state, n_starts=K)`` (or ``mpc.n_starts = K``) with K > 1 runs K bounded quasi-Newton searches in
  lock-step (multistart.py): every solver iteration is ONE ``evaluate_batch`` call of K candidate plans -- sharded over the
  ranks when ``torch.distributed`` is initialised -- and the best local optimum is returned.  ``n_starts = 1`` (the default)
  is the reference's single zero-start solve (src/mpc.py:292-326).
"""
import numpy as np
import torch

from .autograd import CostFunction, wants_grad
from .dynamics import Dynamics
from .rollout import CostParams, cost_full, rollout, rollout_fullcov

try:                                    # the solver binding is optional (not installed in the build image)
    import cyipopt                      # noqa: F401
    HAVE_CYIPOPT = True
except Exception:                       # pragma: no cover
    cyipopt = None
    HAVE_CYIPOPT = False


class RiskSensitiveMPC:
    def __init__(self, gamma, horizon, state_dim, input_dim, Q, R, R_delta=None, full_covariance=False):
        # full_covariance=True propagates the whole state covariance (off-diagonal terms from the exact
        # cross-covariances): the extension the reference leaves as a TODO (src/dynamics.py:184), BASELINE config 5.
        self.full_covariance = full_covariance
        self.gamma = gamma
        self.horizon = horizon
        self.state_dim = state_dim
        self.input_dim = input_dim
        self.Q = Q
        self.R = R
        self.R_delta = R_delta
        self.dynamics = Dynamics(self.state_dim, self.input_dim, nominal_models=None)
        self.device = self.dynamics.device
        self.Q_tor = torch.tensor(np.asarray(self.Q), device=self.device).type(torch.float64)
        self.R_tor = torch.tensor(np.asarray(self.R), device=self.device).type(torch.float64)
        self.R_delta_tor = (torch.tensor(np.asarray(self.R_delta), device=self.device).type(torch.float64)
                            if self.R_delta is not None else None)
        self.x_ref = torch.zeros(self.state_dim, device=self.device)
        self.u_ref = torch.zeros(self.input_dim, device=self.device)
        self.curr_cost = None
        self.curr_grad = None
        self.curr_state = None
        self.backward_taken = False
        self.curr_u = None
        self._cache_key = None
        self.last_traj = np.random.standard_normal(size=(self.horizon * self.input_dim,))   # src/mpc.py:62
        self.ub = [1e16 for _ in range(self.input_dim)]
        self.lb = [-1e16 for _ in range(self.input_dim)]
        self.train_empty = True
        self.solver_used = None
        # lock-step multi-start solve (extension; multistart.py): K starts, tick budget, seed of the sampled starts
        self.n_starts = 1
        self.multistart_options = {"max_ticks": 150, "history": 6, "gtol": 1e-4, "ftol": 1e-10, "spread": 1.0, "seed": 0, "warm": True,
                                   "patience": 8, "line_points": 4}      # 4 step lengths per start and tick: one batch of 4 K plans
        self.last_solve_info = None
        self._solve_count = 0

    # -- setters (src/mpc.py:72-116)
    def set_ub(self, ub):
        assert len(ub) == self.input_dim
        self.ub = ub

    def set_lb(self, lb):
        assert len(lb) == self.input_dim
        self.lb = lb

    def set_xref(self, x_ref):
        assert len(x_ref) == self.state_dim
        self.x_ref = torch.tensor(np.asarray(x_ref), device=self.device).type(torch.float64)
        self._cache_key = None

    def set_uref(self, u_ref):
        assert len(u_ref) == self.input_dim
        self.u_ref = torch.tensor(np.asarray(u_ref), device=self.device).type(torch.float64)
        self._cache_key = None

    # -- cost
    def _cost_params(self, x_ref=None, u_ref=None):
        xr = self.x_ref if x_ref is None else x_ref
        ur = self.u_ref if u_ref is None else u_ref
        last_u = np.asarray(self.last_traj, dtype=np.float64)[0:self.input_dim] if self.R_delta is not None else None
        # rebuilt only when one of its inputs changes: the solver calls this once per callback and the device-to-host
        # copies of x_ref / u_ref alone cost as much as a small rollout
        objs = (self.Q, self.R, self.R_delta, xr, ur)             # kept alive by the cache: `is` cannot alias new objects
        vers = tuple(t._version if isinstance(t, torch.Tensor) else None for t in objs)
        key = (self.gamma, vers, None if last_u is None else last_u.tobytes())
        old = getattr(self, "_cp_state", None)
        if old is None or old[0] != key or any(a is not b for a, b in zip(old[1], objs)):
            to_np = lambda t: t.detach().cpu().numpy().astype(np.float64) if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)  # noqa: E731
            self._cp = CostParams(self.gamma, self.Q, self.R, R_delta=self.R_delta, x_ref=to_np(xr), u_ref=to_np(ur),
                                  last_u=last_u)
            self._cp_state = (key, objs)
        return self._cp

    def cost(self, x, u, sig, x_ref, u_ref):
        """numpy risk-sensitive cost, no input-rate term (src/mpc.py:118-154); host arithmetic in the
        reference too."""
        Q, R, g = np.asarray(self.Q, dtype=float), np.asarray(self.R, dtype=float), self.gamma
        Qi = np.linalg.inv(Q)
        eye = np.identity(self.state_dim)
        total = 0
        for i in range(self.horizon + 1):
            e = x[i, :] - x_ref
            total += np.log(np.linalg.det(eye + g * Q @ sig[i, :, :])) / g
            total += e.T @ np.linalg.inv(Qi + g * sig[i, :, :]) @ e
        for j in range(self.horizon):
            d = u[j, :] - u_ref
            total += d.T @ R @ d
        return total

    def cost_torch(self, x, u, sig, x_ref, u_ref):
        """Risk-sensitive cost incl. the input-rate term for FULL covariance matrices
        (src/mpc.py:156-200) on the device; x / sig may be lists of tensors or stacked tensors.  Differentiable
        like the reference's: if any of x, sig, u carries a graph the result does too (autograd.CostFunction), so
        ``cost_torch(...).backward()`` fills ``u.grad`` as src/mpc.py:251 does."""
        xs = torch.stack([t.reshape(-1) for t in x]) if isinstance(x, (list, tuple)) else torch.as_tensor(x)
        ss = torch.stack(list(sig)) if isinstance(sig, (list, tuple)) else torch.as_tensor(sig)
        ss = ss.reshape(xs.shape[0], self.state_dim, self.state_dim)
        xs = xs.reshape(-1, self.state_dim)
        ut = torch.as_tensor(u).reshape(-1, self.input_dim)
        cp = self._cost_params(x_ref, u_ref)
        if wants_grad(xs, ss, ut):
            to = lambda t: t.to(self.device, torch.float64)  # noqa: E731
            return CostFunction.apply(to(xs)[None], to(ss)[None], to(ut)[None], cp)[0]
        return cost_full(cp, xs, ss, ut)[0]

    # -- cyipopt problem object (src/mpc.py:202-267)
    def _evaluate(self, x):
        x = np.array(x, dtype=np.float64, copy=True).reshape(-1)
        cs, pack, cp = self.curr_state, self.dynamics.pack(), self._cost_params()
        key = (x.tobytes(), None if cs is None else cs._version, bool(self.full_covariance), pack.generation)
        held = getattr(self, "_cache_held", (None, None, None))    # kept alive so that `is` cannot alias later objects
        if key != self._cache_key or cs is not held[0] or pack is not held[1] or cp is not held[2]:
            if self.full_covariance:
                r = rollout_fullcov(pack, cs, x.reshape(self.horizon, self.input_dim), cp, want_grad=True)
                self.curr_cost = float(r["cost"][0].item())
                self.curr_grad = r["grad"][0].cpu().numpy()
            else:
                # B = 1 is pure latency: upload, the H + 1 kernels and the download are ONE captured hipGraph owned by the
                # pack (C ABI gpmpc_objective_gradient) -- one launch and one wait per callback pair
                # host copy of the state, keyed on ITS OWN source tensor + version (the full-covariance branch also
                # refreshes _cache_held, so that cannot vouch for this copy)
                if getattr(self, "_cs_host_src", None) is not cs or getattr(self, "_cs_host_version", None) != cs._version:
                    self._cs_host = cs.detach().cpu().numpy().astype(np.float64).reshape(-1)
                    self._cs_host_src, self._cs_host_version = cs, cs._version
                cg = pack.objective_gradient(self._cs_host, x.reshape(self.horizon, self.input_dim), cp)
                self.curr_cost = float(cg[0])
                self.curr_grad = cg[1:].reshape(self.horizon, self.input_dim).copy()
            self.curr_u = x.reshape(self.horizon, self.input_dim)
            self.backward_taken = True
            self._cache_key = key
            self._cache_held = (cs, pack, cp)
        return self.curr_cost, self.curr_grad

    def objective(self, x):
        return self._evaluate(x)[0]

    def gradient(self, x):
        """d cost / d U, shape (horizon, input_dim) like the reference (cyipopt flattens it)."""
        return self._evaluate(x)[1]

    def constraints(self, x):
        return 0

    def jacobian(self, x):
        return np.zeros(x.shape)

    # -- batched evaluation (extension)
    def evaluate_batch(self, U, curr_state=None, want_grad=True):
        """U: (B, H, da) candidates from one (ds,) or per-candidate (B, ds) start state.
        Returns the rollout dict (device tensors: cost (B,), grad (B,H,da), means, vars)."""
        cs = self.curr_state if curr_state is None else curr_state
        if self.full_covariance:
            return rollout_fullcov(self.dynamics.pack(), cs, U, self._cost_params(), want_grad=want_grad)
        return rollout(self.dynamics.pack(), cs, U, self._cost_params(), want_grad=want_grad, want_traj=True)

    def objective_batch(self, U, curr_state=None):
        r = self.evaluate_batch(U, curr_state)
        return r["cost"].cpu().numpy(), r["grad"].cpu().numpy()

    # -- solve (src/mpc.py:269-330)
    def get_optimal_trajectory(self, curr_state, n_starts=None):
        if self.train_empty:
            if self.dynamics.gpr_err[0].num_train > 0:
                self.train_empty = False
            else:
                return np.zeros((self.horizon, self.input_dim))
        self.curr_state = torch.tensor(np.asarray(curr_state), device=self.device).type(torch.float64)
        self._cache_key = None
        x0 = np.zeros(shape=len(self.last_traj))          # warm start deliberately off, src/mpc.py:292-293
        lb, ub = self.horizon * list(self.lb), self.horizon * list(self.ub)
        K = self.n_starts if n_starts is None else n_starts
        if K > 1:
            x = self._solve_multistart(K, lb, ub)
            self.last_traj = x
            return np.reshape(x, (self.horizon, self.input_dim))
        if HAVE_CYIPOPT:
            nlp = cyipopt.Problem(n=len(x0), m=0, problem_obj=self, lb=lb, ub=ub, cl=[0], cu=[0])
            for k, v in (("mu_strategy", "adaptive"), ("accept_every_trial_step", "yes"), ("max_iter", 300),
                         ("tol", 1e-4), ("acceptable_tol", 1e-4), ("constr_viol_tol", 1e-4), ("compl_inf_tol", 1e-4),
                         ("dual_inf_tol", 1e-4), ("mu_target", 1e-4), ("acceptable_iter", 3), ("sb", "yes"),
                         ("print_level", 0)):
                nlp.add_option(k, v)
            x, _ = nlp.solve(x0)
            self.solver_used = "ipopt"
        else:
            x = self._solve_without_ipopt(x0, lb, ub)
        self.last_traj = x
        return np.reshape(x, (self.horizon, self.input_dim))

    def _solve_without_ipopt(self, x0, lb, ub):
        """Stand-in driver when cyipopt is not installed: bounded L-BFGS (scipy) on the same
        objective/gradient callbacks.  Not Ipopt: optimiser results are not parity-pinned."""
        from scipy.optimize import minimize
        big = 1e15
        bounds = [(None if l <= -big else l, None if u >= big else u) for l, u in zip(lb, ub)]
        res = minimize(lambda v: self.objective(v), x0, jac=lambda v: np.asarray(self.gradient(v)).reshape(-1),
                       method="L-BFGS-B", bounds=bounds, options={"maxiter": 300, "ftol": 1e-10, "gtol": 1e-4})
        self.solver_used = "scipy-lbfgsb"
        return res.x

    def _solve_multistart(self, K, lb, ub):
        """K starts advanced together (multistart.lockstep_lbfgs): one batched rollout of K plans per solver iteration, replayed as
        one hipGraph; with torch.distributed initialised the K plans are sharded over the ranks (parallel.sharded_rollout) and every
        rank takes the same decisions from the same gathered [cost | grad].  Optimiser results unpinned, like the Ipopt stand-in."""
        from .multistart import lockstep_lbfgs, make_starts
        opt = self.multistart_options
        n, H, da = self.horizon * self.input_dim, self.horizon, self.input_dim
        pack, cp, cs = self.dynamics.pack(), self._cost_params(), self.curr_state
        warm = None
        if opt.get("warm", True) and self.solver_used is not None:        # the previous plan shifted by one step, last input repeated
            prev = np.asarray(self.last_traj, dtype=np.float64).reshape(H, da)
            warm = np.concatenate((prev[1:], prev[-1:]), axis=0).reshape(-1)
        rng = np.random.default_rng([int(opt.get("seed", 0)), self._solve_count])
        self._solve_count += 1
        X0 = make_starts(K, n, lb, ub, rng, warm=warm, spread=float(opt.get("spread", 1.0)))
        dist = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            dist = torch.distributed

        def run(x0b, Ub, graph):
            if self.full_covariance:
                return rollout_fullcov(pack, x0b, Ub, cp, want_grad=True)
            return rollout(pack, x0b, Ub, cp, want_grad=True, want_traj=False, graph=graph)

        def evaluate(X):                                                  # (K, n), or (S K, n): S step lengths per start
            nb = X.shape[0]
            U = np.ascontiguousarray(X.reshape(nb, H, da))
            if dist is not None:
                from .parallel import sharded_rollout
                Ud = torch.as_tensor(U, device=self.device)
                c, g = sharded_rollout(lambda x0b, Ub: run(x0b, Ub, False), cs, Ud, dist)
                return c.cpu().numpy(), g.cpu().numpy().reshape(nb, n)
            r = run(cs, U, True)
            if "cost_grad" in r:                                          # cost and gradient in ONE device-to-host copy
                cg = r["cost_grad"].cpu().numpy()
                return cg[:nb], cg[nb:].reshape(nb, n)
            return r["cost"].cpu().numpy(), r["grad"].cpu().numpy().reshape(nb, n)

        x, info = lockstep_lbfgs(evaluate, X0, np.asarray(lb, dtype=np.float64), np.asarray(ub, dtype=np.float64),
                                 max_ticks=int(opt.get("max_ticks", 150)), history=int(opt.get("history", 8)),
                                 gtol=float(opt.get("gtol", 1e-4)), ftol=float(opt.get("ftol", 1e-10)), patience=opt.get("patience"),
                                 line_points=int(opt.get("line_points", 1)))
        info["starts"] = K
        info["sharded_over"] = dist.get_world_size() if dist is not None else 1
        self.last_solve_info = info
        self.solver_used = f"lockstep-lbfgs x{K}"
        self._cache_key = None
        return x
