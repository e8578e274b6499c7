"""Dynamics: host-side mirror of the reference class (src/dynamics.py:8-191) -- a bundle of
``state_dim`` GPs sharing X_train -- whose rollout runs in the HIP library.

``forward_propagate_torch`` keeps the reference signature and return types (lists of H+1 tensors, graph
attached when the actions require grad).
The batched entry ``rollout`` is the build's extension (the reference handles one trajectory per call).
"""
import numpy as np
import torch

from .gpr import GaussianProcessRegression
from .autograd import RolloutFunction, wants_grad
from .rollout import CostParams, GPPack, rollout, rollout_fullcov


class Dynamics(object):
    def __init__(self, state_dim, action_dim, nominal_models=None):
        self.state_dim = state_dim
        self.action_dim = action_dim
        self.nominal_models = nominal_models
        if nominal_models is None:
            self.gpr_err = [GaussianProcessRegression(state_dim + action_dim) for _ in range(state_dim)]
        else:
            self.gpr_err = [GaussianProcessRegression(state_dim + action_dim, nominal_models[i])
                            for i in range(state_dim)]
        self.device = self.gpr_err[0].device
        self._pack = None
        self._pack_key = None

    def append_train_data(self, state, action, next_state, incremental=False, async_rebuild=None, refresh=None):
        """(state, action, next_state) observations, one or many (src/dynamics.py:39-60).  incremental=True: O(N^2)
        update of every Ky_inv for a single new observation (see GaussianProcessRegression.append_train_data);
        async_rebuild (True / False; None leaves the GPs' setting): the periodic full rebuild of the incremental path on a
        side stream instead of on the step that reaches `rebuild_every`; refresh ("rebuild" / "newton"; None leaves the GPs'
        setting): a Newton-Schulz polish of the updated inverse instead of that rebuild (GaussianProcessRegression.refresh)."""
        if async_rebuild is not None:
            for g in self.gpr_err:
                g.async_rebuild = bool(async_rebuild)
        if refresh is not None:
            if refresh not in ("rebuild", "newton"):
                raise ValueError("refresh must be 'rebuild' or 'newton', got %r" % (refresh,))
            for g in self.gpr_err:
                g.refresh = refresh
        state, action, next_state = np.asarray(state), np.asarray(action), np.asarray(next_state)
        # Every GP receives the same input rows; GPs whose hyper-parameters are bit-identical then have identical Ky /
        # Ky_inv and share one build (GaussianProcessRegression.update_many).  That only holds while ALL data went through
        # this method: once a GP was fed on its own (its X_train is no longer the tensor left here), each is updated by itself.
        uniform = getattr(self, "_uniform_ok", True)
        if uniform:
            seen = getattr(self, "_seen_X", None)       # the input tensors this method left in the GPs last time
            if seen is None:
                uniform = all(g.num_train == 0 for g in self.gpr_err)
            else:
                uniform = all(g.X_train is sx for g, sx in zip(self.gpr_err, seen))
        self._uniform_ok = uniform                       # once a GP was fed on its own, never again
        if len(state.shape) == 1:
            x = np.concatenate((state, action))
            ys = [next_state[i] for i in range(self.state_dim)]
        else:
            if len(action.shape) == 1:
                action = action[:, None]
            x = np.concatenate((state, action), axis=1)
            ys = [next_state[:, i] for i in range(self.state_dim)]
            incremental = False
        if not uniform:
            for g, y in zip(self.gpr_err, ys):
                g.append_train_data(x, y, incremental=incremental)
        else:
            modes = []
            # one device copy of the new input rows and one of ALL targets for the ds GPs (they were 2 ds host-to-device copies)
            # (made on first use: the in-place path of a single observation sends its targets as kernel arguments instead)
            shared = {"y_all_host": np.asarray(next_state, dtype=np.float64).reshape(-1, self.state_dim)}
            for a, (g, y) in enumerate(zip(self.gpr_err, ys)):
                if not np.isscalar(y) and np.ndim(y) > 0:
                    n_obs, yy = len(y), np.asarray(y)[:, None]
                    xx = x
                else:
                    n_obs, yy = 1, np.array([y])[:, None]
                    xx = np.reshape(x, (1, g.x_dim))
                modes.append(g._ingest(xx, yy, n_obs, incremental, shared=shared, column=a))
            GaussianProcessRegression.update_many(self.gpr_err, modes)
        self._seen_X = [g.X_train for g in self.gpr_err]

    # -- device pack -----------------------------------------------------------------------
    def _key(self):
        # what forward_propagate_torch reads at call time in the reference (src/dynamics.py:150, :170-173):
        # Ky_inv, exp(log_lambdas), y_train, sigma_f of every GP, X_train of GP 0.  Tensors are keyed by OBJECT and
        # autograd version (the key holds the references, so an id cannot be reused by a later tensor): this runs on
        # every solver callback, and exp / tolist of the hyper-parameters per call cost as much as a small rollout.
        return [(g.version, g.num_train, g.log_lambdas, g.log_lambdas._version, g.log_sigma_f, g.log_sigma_f._version)
                for g in self.gpr_err]

    @staticmethod
    def _same_key(a, b):
        if a is None or b is None or len(a) != len(b):
            return False
        return all(x[0] == y[0] and x[1] == y[1] and x[2] is y[2] and x[3] == y[3] and x[4] is y[4] and x[5] == y[5]
                   for x, y in zip(a, b))

    def pack(self):
        """The device-resident constants of the rollout, rebuilt only when data or hypers changed."""
        key = self._key()
        if self._pack is None or not self._same_key(key, self._pack_key):
            g0 = self.gpr_err[0]
            if g0.num_train == 0:
                raise RuntimeError("no training data")
            Y = torch.cat([g.y_train.reshape(-1, 1) for g in self.gpr_err], dim=1)
            if all(g.Ky_inv is g0.Ky_inv for g in self.gpr_err):
                Kinv = g0.Ky_inv.detach()               # GPs with identical hyper-parameters share ONE inverse: read in place, no stack
            else:
                Kinv = torch.stack([g.Ky_inv.detach() for g in self.gpr_err])
            lam = np.stack([g.get_lambdas() for g in self.gpr_err])
            sf = np.array([g.get_sigma_f() for g in self.gpr_err])
            # the closed loop appends one observation per step: refill the existing pack while its padded size fits
            if self._pack is None or not self._pack.rebuild(g0.X_train, Y, Kinv, lam, sf):
                self._pack = GPPack(g0.X_train, Y, Kinv, lam, sf, device=self.device)
            self._pack_key = key
        return self._pack

    # -- rollout ---------------------------------------------------------------------------
    def rollout(self, curr_state, actions, cost=None, want_grad=False, full_covariance=False):
        """Batched rollout: curr_state (ds,) or (B, ds); actions (H, da) or (B, H, da).
        Returns the dict of gaussian_process_mpc_amd.rollout.rollout (or rollout_fullcov: 'covs' instead of 'vars')."""
        if cost is None:     # propagation only: a zero cost keeps the fused tail trivial
            cost = CostParams(0.0, np.zeros((self.state_dim, self.state_dim)), np.zeros((self.action_dim, self.action_dim)))
        if full_covariance:
            return rollout_fullcov(self.pack(), curr_state, actions, cost, want_grad=want_grad)
        return rollout(self.pack(), curr_state, actions, cost, want_grad=want_grad, want_traj=True)

    def forward_propagate(self, horizon, curr_state, actions):
        """The reference's numpy rollout (src/dynamics.py:62-124; its slow double-loop oracle, sigma_f = 1 only) with the same
        signature and return types -- numpy (H+1, ds) means and (H+1, ds, ds) diagonal covariances -- evaluated by the HIP
        path.  One documented difference: the numpy version adds an action-noise variance of exactly 1e-3, the torch version
        (and this library) float32(1e-3) (src/dynamics.py:162); the reference's own test holds the two to 1e-7
        (src/test/test_dynamics.py:134-196) and so do these values."""
        r = self.rollout(np.asarray(curr_state, dtype=np.float64).reshape(self.state_dim),
                         np.asarray(actions, dtype=np.float64).reshape(horizon, self.action_dim))
        means = r["means"][0].cpu().numpy()
        covars = np.zeros((horizon + 1, self.state_dim, self.state_dim))
        v = r["vars"][0].cpu().numpy()
        for t in range(horizon + 1):
            covars[t] = np.diag(v[t])
        return means, covars

    def forward_propagate_torch(self, horizon, curr_state, actions):
        """Means and (diagonal) covariances of the H-step shooting rollout (src/dynamics.py:126-191).
        Returns (list of H+1 (ds,) tensors, list of H+1 (ds,ds) tensors) on the device.  Like the reference's, the
        tensors carry the autograd graph when ``actions`` (or ``curr_state``) requires grad: the reference pattern
        ``forward_propagate_torch -> cost_torch -> backward()`` (src/mpc.py:217-255) works on the mirror classes; the
        graph has ONE node for the whole rollout (autograd.RolloutFunction: HIP forward, analytic step Jacobians, reverse
        sweep on the device) instead of ~40 torch ops per (step, GP)."""
        U = torch.as_tensor(actions)
        x0 = torch.as_tensor(curr_state)
        if wants_grad(U, x0):
            Ud = U.to(self.device, torch.float64).reshape(1, horizon, self.action_dim)
            xd = x0.to(self.device, torch.float64).reshape(1, self.state_dim)
            means, vars_ = RolloutFunction.apply(xd, Ud, self.pack())
            means, vars_ = means[0], vars_[0]
        else:
            U = U.detach().to(self.device, torch.float64).reshape(horizon, self.action_dim)
            x0 = x0.detach().to(self.device, torch.float64).reshape(self.state_dim)
            r = self.rollout(x0, U)
            means, vars_ = r["means"][0], r["vars"][0]
        return [means[t] for t in range(horizon + 1)], [torch.diag(vars_[t]) for t in range(horizon + 1)]
