#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run once in the build container (where the upstream repository is mounted
read-only at /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

The reference is imported unmodified; the only shim is an empty ``cyipopt``
module (the solver binding is not installed and none of the functions
exercised here touch it).  The outputs are plain data (seeded inputs and the
reference's results) stored as ``.npz``; no reference source is stored.

Fixtures:
  g1_single_step.npz   test-suite geometry, N=100/200, D=2, full S
  g2_adversarial.npz   N=48, D=3..5, non-proportional lambdas, diag and full S, autograd d/du, d/dS of mean, variance and
                       (round 4) of covariance_prop_torch, with graph-attached and with constant means
  g3_rollout_c1.npz    N=100, ds=2, da=2, H=10, gamma in {-1, 1e-5, 1}
  g4_rollout_c2.npz    N=128, ds=3, da=1, H=20, R_delta + nonzero references
  g5_cost.npz          literal cost cases of the reference's tests
  g6_gp.npz            Ky / Ky_inv / K* / predict
  g7_quirks.npz        dtype quirks of the rollout
  g8_hyper.npz         marginal likelihood at set hyper-parameters; update_hyperparams trajectories (Adam)
  g9_closed_loop.npz   the callers either side of the path: plant updates of the reference's two
                       environments (cart-pole stepPhysics, pendulum step / step_static) on seeded inputs, and the data
                       file triple of the README experiment (src/experiments/data/*.npy, stored as arrays)

  g10_readme_regime.npz  the reference's own regime: README experiment data, lambda = 0.5 for every GP, H = 6,
                       sigma_n in {1e-3, 1e-5}: means / variances / cost / gradient of four candidate plans
  g11_fullsize_pin.npz reference-produced mean / variance of every GP at N = 2048 (C3 training set re-derived from the
                       seed) for two single-step queries

  g12_fullsize_rollout.npz reference-produced objective / gradient / trajectory of two plans at N = 2048, H = 2

g9 imports src/environments/*, which need ``gym`` (absent here).  A minimal stand-in module (``gym.Env`` as a bare base
class, ``spaces.Box`` recording its arguments, ``seeding.np_random``, ``error.DependencyNotInstalled``) is registered
for that import only; none of the arithmetic captured below goes through it.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("GPMPC_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.modules.setdefault("cyipopt", types.ModuleType("cyipopt"))

from src.gpr import GaussianProcessRegression                      # noqa: E402
from src.dynamics import Dynamics                                  # noqa: E402
from src.mpc import RiskSensitiveMPC                               # noqa: E402
from src.tools.uncertainty_prop import (                           # noqa: E402
    mean_prop, variance_prop, covariance_prop,
    mean_prop_torch, variance_prop_torch, covariance_prop_torch)

OUT = os.path.dirname(os.path.abspath(__file__))
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)     # noqa: E731


def se_K(X, lam, sf=1.0):
    d = (X[:, None, :] - X[None, :, :])
    return sf ** 2 * np.exp(-0.5 * np.sum(d * d / lam, axis=2))


def g1():
    out = {}
    rng = np.random.default_rng(101)
    u = np.array([2.0, 1.0])
    S = np.array([[1.0, 0.5], [0.5, 2.0]])
    lam1, lam2 = np.array([1.0, 1.0]), np.array([2.0, 2.0])
    sigma = 0.5
    for tag, N, sf1, sf2 in (("a", 100, 1.0, 1.0), ("b", 200, 1.0, 1.0), ("c", 100, 0.8, 1.5)):
        X = rng.multivariate_normal(u, S, N)
        y = np.sum(X * X, axis=1) + rng.normal(0, sigma, N)
        Ky1 = se_K(X, lam1, sf1) + sigma ** 2 * np.eye(N)
        Ky2 = se_K(X, lam2, sf2) + sigma ** 2 * np.eye(N)
        Ki1, Ki2 = torch.linalg.inv(T(Ky1)), torch.linalg.inv(T(Ky2))
        m1, d1 = mean_prop_torch(Ki1, T(lam1), T(u), T(S), T(X), T(y), sf1)
        m2, d2 = mean_prop_torch(Ki2, T(lam2), T(u), T(S), T(X), T(y), sf2)
        v1 = variance_prop_torch(Ki1, T(lam1), T(u), T(S), T(X), m1, d1["beta"], sf1)
        v2 = variance_prop_torch(Ki2, T(lam2), T(u), T(S), T(X), m2, d2["beta"], sf2)
        cv = covariance_prop_torch(T(lam1), T(lam2), T(u), T(S), T(X), m1, m2, d1["beta"], d2["beta"], sf1, sf2)
        out.update({f"{tag}_X": X, f"{tag}_y": y, f"{tag}_Kinv1": Ki1.numpy(), f"{tag}_Kinv2": Ki2.numpy(),
                    f"{tag}_sf": np.array([sf1, sf2]),
                    f"{tag}_mu": np.array([m1.item(), m2.item()]),
                    f"{tag}_beta1": d1["beta"].numpy(), f"{tag}_l1": d1["l"].numpy(),
                    f"{tag}_var": np.array([v1.item(), v2.item()]), f"{tag}_cov": np.array(cv.item())})
        if sf1 == 1.0 and N == 100:   # numpy loop formulas only support sigma_f = 1
            nm, nd = mean_prop(Ky1, np.diag(lam1), u, S, X, y)
            nv = variance_prop(Ky1, np.diag(lam1), u, S, X, y)
            nc = covariance_prop(Ky1, Ky2, np.diag(lam1), np.diag(lam2), u, S, X, y)
            out.update({f"{tag}_np_mu": np.array(nm), f"{tag}_np_var": np.array(nv), f"{tag}_np_cov": np.array(nc),
                        f"{tag}_np_beta": nd["beta"], f"{tag}_np_l": nd["l"]})
    out.update({"u": u, "S": S, "lam1": lam1, "lam2": lam2, "sigma": np.array(sigma)})
    np.savez(os.path.join(OUT, "g1_single_step.npz"), **out)


def g2():
    out = {}
    rng = np.random.default_rng(202)
    N = 48
    cases = []
    for D in (3, 4, 5):
        for full in (False, True):
            cases.append((D, full))
    out["cases"] = np.array([[D, int(f)] for D, f in cases])
    for k, (D, full) in enumerate(cases):
        X = rng.uniform(-2, 2, (N, D))
        y1 = np.sin(X).sum(axis=1) + 0.05 * rng.normal(size=N)
        y2 = np.cos(X[:, 0]) * X[:, 1] + 0.05 * rng.normal(size=N)
        lam1, lam2 = rng.uniform(0.5, 3.0, D), rng.uniform(0.5, 3.0, D)
        sf1, sf2, sn = 1.3, 0.7, 0.05
        u = rng.uniform(-1, 1, D)
        if full:
            Aq = rng.normal(size=(D, D))
            S = 0.05 * (Aq @ Aq.T) + 0.01 * np.eye(D)
        else:
            S = np.diag(rng.uniform(0.001, 0.2, D))
        Ki1 = torch.linalg.inv(T(se_K(X, lam1, sf1) + sn ** 2 * np.eye(N)))
        Ki2 = torch.linalg.inv(T(se_K(X, lam2, sf2) + sn ** 2 * np.eye(N)))
        ut = T(u).requires_grad_(True)
        St = T(S).requires_grad_(True)
        m1, d1 = mean_prop_torch(Ki1, T(lam1), ut, St, T(X), T(y1), sf1)
        v1 = variance_prop_torch(Ki1, T(lam1), ut, St, T(X), m1, d1["beta"], sf1)
        dm_du, dm_dS = torch.autograd.grad(m1, (ut, St), retain_graph=True)
        dv_du, dv_dS = torch.autograd.grad(v1, (ut, St), retain_graph=True)
        m2, d2 = mean_prop_torch(Ki2, T(lam2), ut, St, T(X), T(y2), sf2)
        v2 = variance_prop_torch(Ki2, T(lam2), ut, St, T(X), m2, d2["beta"], sf2)
        cv = covariance_prop_torch(T(lam1), T(lam2), ut, St, T(X), m1, m2, d1["beta"], d2["beta"], sf1, sf2)
        # autograd of the covariance: total (through the graph-attached means) and with the means as constants
        dc_du, dc_dS = torch.autograd.grad(cv, (ut, St), retain_graph=True)
        cv0 = covariance_prop_torch(T(lam1), T(lam2), ut, St, T(X), m1.detach(), m2.detach(), d1["beta"], d2["beta"], sf1, sf2)
        dc0_du, dc0_dS = torch.autograd.grad(cv0, (ut, St), retain_graph=True)
        # numpy loop covariance (sigma_f = 1 and one shared y): the mathematically consistent form
        Ky1u = se_K(X, lam1) + sn ** 2 * np.eye(N)
        Ky2u = se_K(X, lam2) + sn ** 2 * np.eye(N)
        nc = covariance_prop(Ky1u, Ky2u, np.diag(lam1), np.diag(lam2), u, S, X, y1)
        Ki1u, Ki2u = torch.linalg.inv(T(Ky1u)), torch.linalg.inv(T(Ky2u))
        mu1u, du1 = mean_prop_torch(Ki1u, T(lam1), T(u), T(S), T(X), T(y1))
        mu2u, du2 = mean_prop_torch(Ki2u, T(lam2), T(u), T(S), T(X), T(y1))
        cvu = covariance_prop_torch(T(lam1), T(lam2), T(u), T(S), T(X), mu1u, mu2u, du1["beta"], du2["beta"])
        p = f"c{k}_"
        out.update({p + "X": X, p + "y1": y1, p + "y2": y2, p + "lam1": lam1, p + "lam2": lam2,
                    p + "hyp": np.array([sf1, sf2, sn]), p + "u": u, p + "S": S,
                    p + "Kinv1": Ki1.numpy(), p + "Kinv2": Ki2.numpy(),
                    p + "mu": np.array([m1.item(), m2.item()]), p + "var": np.array([v1.item(), v2.item()]),
                    p + "cov_torch": np.array(cv.item()),
                    p + "dcov_du": dc_du.numpy(), p + "dcov_dS": dc_dS.numpy(),
                    p + "dcov_du_means_const": dc0_du.numpy(), p + "dcov_dS_means_const": dc0_dS.numpy(),
                    p + "dm_du": dm_du.numpy(), p + "dm_dS": dm_dS.numpy(),
                    p + "dv_du": dv_du.numpy(), p + "dv_dS": dv_dS.numpy(),
                    p + "unit_Kinv1": Ki1u.numpy(), p + "unit_Kinv2": Ki2u.numpy(),
                    p + "unit_cov_numpy": np.array(nc), p + "unit_cov_torch": np.array(cvu.item())})
    np.savez(os.path.join(OUT, "g2_adversarial.npz"), **out)


def _build_mpc(rng, N, ds, da, H, gamma, lam, sn, Q, R, R_delta=None):
    D = ds + da
    S = rng.uniform(-2, 2, (N, ds))
    A = rng.uniform(-1, 1, (N, da))
    Y = S + 0.1 * np.tanh(S) + 0.1 * A.sum(axis=1, keepdims=True)
    mpc = RiskSensitiveMPC(gamma, H, ds, da, Q, R, R_delta)
    for a in range(ds):
        g = mpc.dynamics.gpr_err[a]
        g.set_lambdas(lam[a])
        g.set_sigma_n(sn)
        g.set_sigma_f(1.0)
    mpc.dynamics.append_train_data(S, A, Y)
    return mpc, np.concatenate((S, A), axis=1), Y


def _rollout_fixture(name, seed, N, ds, da, H, gammas, lam_rng, sn, Q, R, R_delta=None,
                     x_ref=None, u_ref=None, last_u=None, n_traj=2):
    rng = np.random.default_rng(seed)
    D = ds + da
    lam = rng.uniform(lam_rng[0], lam_rng[1], (ds, D))
    out = {"lambdas": lam, "sigma_f": np.ones(ds), "sigma_n": np.full(ds, sn), "Q": Q, "R": R,
           "gammas": np.array(gammas), "dims": np.array([N, ds, da, H])}
    mpc, X, Y = _build_mpc(rng, N, ds, da, H, gammas[0], lam, sn, Q, R, R_delta)
    out.update({"X": X, "Y": Y,
                "Ky_inv": np.stack([g.Ky_inv.detach().numpy() for g in mpc.dynamics.gpr_err])})
    if x_ref is not None:
        mpc.set_xref(x_ref); out["x_ref"] = x_ref
    if u_ref is not None:
        mpc.set_uref(u_ref); out["u_ref"] = u_ref
    if R_delta is not None:
        out["R_delta"] = R_delta
        mpc.last_traj = np.asarray(last_u, dtype=np.float64)
        out["last_traj"] = mpc.last_traj
    x0 = rng.uniform(-1, 1, (n_traj, ds))
    U = rng.uniform(-1, 1, (n_traj, H, da))
    out.update({"x0": x0, "U": U})
    means = np.zeros((n_traj, H + 1, ds)); vars_ = np.zeros((n_traj, H + 1, ds))
    costs = np.zeros((len(gammas), n_traj)); grads = np.zeros((len(gammas), n_traj, H, da))
    for b in range(n_traj):
        mpc.curr_state = torch.tensor(x0[b]).type(torch.float64)
        for gi, gm in enumerate(gammas):
            mpc.gamma = gm
            mpc.curr_cost = None
            c = mpc.objective(U[b].reshape(-1).copy())
            g = mpc.gradient(U[b].reshape(-1).copy())
            costs[gi, b] = c
            grads[gi, b] = np.asarray(g).reshape(H, da)
        sm, sc = mpc.dynamics.forward_propagate_torch(H, mpc.curr_state, torch.tensor(U[b]).type(torch.float64))
        means[b] = torch.stack(sm).detach().numpy()
        vars_[b] = torch.stack([torch.diag(s) for s in sc]).detach().numpy()
        offd = max(float((s - torch.diag(torch.diag(s))).abs().max()) for s in sc)
        assert offd == 0.0
    out.update({"means": means, "vars": vars_, "costs": costs, "grads": grads})
    np.savez(os.path.join(OUT, name), **out)


def g3():
    _rollout_fixture("g3_rollout_c1.npz", 303, N=100, ds=2, da=2, H=10, gammas=[-1.0, 1e-5, 1.0],
                     lam_rng=(2.0, 6.0), sn=1e-2, Q=0.1 * np.eye(2), R=0.01 * np.eye(2))


def g4():
    _rollout_fixture("g4_rollout_c2.npz", 404, N=128, ds=3, da=1, H=20, gammas=[-1.0, 1e-5],
                     lam_rng=(2.0, 6.0), sn=1e-2, Q=np.diag([0.1, 0.2, 0.05]), R=np.array([[0.01]]),
                     R_delta=np.array([[0.03]]), x_ref=np.array([0.3, -0.2, 0.1]), u_ref=np.array([0.1]),
                     last_u=0.25 * np.ones(20))


def g5():
    out = {}
    # test_mpc.py:15-57 (cost / cost_torch, full non-symmetric Sigma)
    Q = np.array([[2.0, 0], [0, 2.0]]); R = np.array([[1.0, 1], [1, 1.0]])
    x = np.array([[1.0, 1], [3, 3]]); u = np.array([[2.0, 2]])
    sig = np.array([[[1.0, 2], [3, 4]], [[5.0, 6], [7, 8]]])
    xr, ur = np.array([0.5, 0.5]), np.array([0.6, 0.6])
    mpc = RiskSensitiveMPC(1, 1, 2, 2, Q, R)
    out.update({"a_Q": Q, "a_R": R, "a_x": x, "a_u": u, "a_sig": sig, "a_xref": xr, "a_uref": ur,
                "a_gamma": np.array(1.0),
                "a_cost_np": np.array(mpc.cost(x, u, sig, xr, ur)),
                "a_cost_torch": np.array(mpc.cost_torch(T(x), T(u), T(sig), T(xr), T(ur)).item())})
    # test_mpc.py:169-243 (R_delta)
    Rd = np.array([[0.5, 0], [0, 1.5]])
    x = np.array([[1.0, 1], [2, 2], [3, 3]]); u = np.array([[2.0, 2], [4, 4]])
    sig = np.array([[[1.0, 2], [3, 4]], [[5.0, 6], [7, 8]], [[9.0, 10], [11, 12]]])
    mpc = RiskSensitiveMPC(1.1, 2, 2, 2, Q, R, Rd)
    mpc.last_traj = [0 for _ in range(4)]
    c = mpc.cost_torch([T(x[i]) for i in range(3)], T(u), [T(sig[i]) for i in range(3)], T(xr), T(ur))
    out.update({"b_Rd": Rd, "b_x": x, "b_u": u, "b_sig": sig, "b_gamma": np.array(1.1),
                "b_last": np.zeros(4), "b_cost_torch": np.array(c.item())})
    # test_mpc.py:245-274 (gamma = -1 scalar closed form)
    H = 5
    xs = np.array([5.0, 4, 3, 2, 1, 0]); sg = np.array([1 / 6, 1 / 7, 1 / 8, 1 / 9, 1 / 10, 1 / 11])
    closed = sum(-np.log(1 - 2 * sg[i]) + xs[i] ** 2 / (0.5 - sg[i]) for i in range(H + 1))
    mpc = RiskSensitiveMPC(-1, H, 1, 1, 2 * np.eye(1), np.array([[0.0]]), np.array([[0.0]]))
    c = mpc.cost_torch(T(xs).reshape(H + 1, 1), torch.zeros((H, 1), dtype=torch.float64),
                       T(sg).reshape(H + 1, 1, 1), torch.zeros(1, dtype=torch.float64),
                       torch.zeros(1, dtype=torch.float64))
    out.update({"c_x": xs, "c_sig": sg, "c_closed": np.array(closed), "c_cost_torch": np.array(c.item())})
    np.savez(os.path.join(OUT, "g5_cost.npz"), **out)


def g6():
    rng = np.random.default_rng(606)
    N, D, p = 64, 3, 7
    X = rng.uniform(-2, 2, (N, D)); y = np.sin(X).sum(axis=1) + 0.1 * rng.normal(size=N)
    lam = np.array([0.7, 1.9, 3.1]); sf, sn = 1.4, 0.2
    g = GaussianProcessRegression(D)
    g.set_lambdas(lam); g.set_sigma_f(sf); g.set_sigma_n(sn)
    g.append_train_data(X, y)
    Xp = rng.uniform(-2, 2, (p, D))
    Ks = g.compute_pred_train_covariance(Xp).detach().numpy()
    Ks1 = g.compute_pred_train_covariance(Xp[0]).detach().numpy()
    f0, _ = g.predict_latent_vars(Xp)
    f1, cf = g.predict_latent_vars(Xp, covar=True)
    f2, cy = g.predict_latent_vars(Xp, covar=True, targets=True)
    # append in two chunks + one single point gives the same state as the reference's cat path
    g2_ = GaussianProcessRegression(D)
    g2_.set_lambdas(lam); g2_.set_sigma_f(sf); g2_.set_sigma_n(sn)
    g2_.append_train_data(X[:40], y[:40]); g2_.append_train_data(X[40:63], y[40:63])
    g2_.append_train_data(X[63], float(y[63]))
    assert np.allclose(g2_.Ky.detach().numpy(), g.Ky.detach().numpy())
    np.savez(os.path.join(OUT, "g6_gp.npz"), X=X, y=y, lam=lam, hyp=np.array([sf, sn]), Xp=Xp,
             Kf=g.Kf.detach().numpy(), Ky=g.Ky.detach().numpy(), Ky_inv=g.Ky_inv.detach().numpy(),
             Ks=Ks, Ks_single=Ks1, f=f0, f_cov=f1, cov_f=cf, cov_y=cy)


def g7():
    d = Dynamics(2, 1)
    rng = np.random.default_rng(707)
    S = rng.uniform(-1, 1, (6, 2)); A = rng.uniform(-1, 1, (6, 1))
    d.append_train_data(S, A, S + 0.1 * A)
    sm, sc = d.forward_propagate_torch(1, torch.zeros(2, dtype=torch.float64), torch.zeros((1, 1), dtype=torch.float64))
    act = (1e-3 * torch.eye(1)).type(torch.float64)[0, 0].item()    # dynamics.py:162 promoted to fp64
    np.savez(os.path.join(OUT, "g7_quirks.npz"), init_var=np.array(sc[0][0, 0].item()),
             action_var=np.array(act), float32_1e3=np.array(float(np.float32(1e-3))))


def g8():
    """compute_marginal_likelihood (src/gpr.py:240) and update_hyperparams (:334), one iteration per call so that
    the per-iteration state can be recorded (the Adam state lives in the object, so k calls of one iteration are
    k iterations)."""
    import contextlib
    import io
    out = {}
    # likelihood at set hyper-parameters (the g6 problem)
    rng = np.random.default_rng(606)
    N, D = 64, 3
    X = rng.uniform(-2, 2, (N, D)); y = np.sin(X).sum(axis=1) + 0.1 * rng.normal(size=N)
    g = GaussianProcessRegression(D)
    g.set_lambdas(np.array([0.7, 1.9, 3.1])); g.set_sigma_f(1.4); g.set_sigma_n(0.2)
    g.append_train_data(X, y)
    out["ml_X"], out["ml_y"] = X, y
    out["ml_lam"], out["ml_hyp"] = np.array([0.7, 1.9, 3.1]), np.array([1.4, 0.2])
    out["ml_value"] = np.array(g.compute_marginal_likelihood().item())

    def traj(tag, X, y, x_dim, nominal, iters):
        gp = GaussianProcessRegression(x_dim, nominal_model=nominal)
        gp.append_train_data(X, y)
        ml, ll, lf, ln, gl, gf, gn = [], [], [], [], [], [], []
        for _ in range(iters):
            ml.append(gp.compute_marginal_likelihood().item())
            with contextlib.redirect_stdout(io.StringIO()):
                gp.update_hyperparams(num_iters=1)
            ll.append(gp.log_lambdas.detach().numpy().copy()); lf.append(gp.log_sigma_f.item()); ln.append(gp.log_sigma_n.item())
            gl.append(gp.log_lambdas.grad.numpy().copy()); gf.append(gp.log_sigma_f.grad.item()); gn.append(gp.log_sigma_n.grad.item())
        out[tag + "_X"], out[tag + "_y"] = X, y
        out[tag + "_ml"] = np.array(ml)
        out[tag + "_log_lambdas"], out[tag + "_log_sigma_f"], out[tag + "_log_sigma_n"] = np.array(ll), np.array(lf), np.array(ln)
        out[tag + "_g_log_lambdas"], out[tag + "_g_log_sigma_f"], out[tag + "_g_log_sigma_n"] = np.array(gl), np.array(gf), np.array(gn)

    rng = np.random.default_rng(808)
    X1 = np.arange(-5, 6, dtype=float)[:, None]                      # the reference's own test data shape (test_gpr.py:984-997)
    traj("t1", X1, X1[:, 0] ** 2 + rng.normal(size=11), 1, None, 12)
    traj("t2", X1, X1[:, 0] + np.sin(X1[:, 0]) + rng.normal(size=11), 1, (lambda x: x), 12)     # test_gpr.py:1065-1080
    X3 = rng.uniform(-2, 2, (40, 3))
    traj("t3", X3, np.sin(X3).sum(axis=1) + 0.1 * rng.normal(size=40), 3, None, 10)
    np.savez(os.path.join(OUT, "g8_hyper.npz"), **out)


def _gym_stand_in():
    """Just enough of gym's module tree for ``import src.environments.*`` to succeed (class bodies only reference these
    names at construction time; no gym arithmetic exists on the plant updates captured by g9)."""
    gym = types.ModuleType("gym")

    class Env(object):
        np_random = None

        def reset(self, *, seed=None, options=None):
            self.np_random = np.random.default_rng(seed)

    class Box(object):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class DependencyNotInstalled(Exception):
        pass

    gym.Env = Env
    spaces = types.ModuleType("gym.spaces"); spaces.Box = Box
    logger = types.ModuleType("gym.logger"); logger.warn = lambda *a, **k: None
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding"); seeding.np_random = lambda seed=None: (np.random.default_rng(seed), seed)
    utils.seeding = seeding
    error = types.ModuleType("gym.error"); error.DependencyNotInstalled = DependencyNotInstalled
    envs = types.ModuleType("gym.envs")
    cc = types.ModuleType("gym.envs.classic_control")
    ccu = types.ModuleType("gym.envs.classic_control.utils")
    cc.utils = ccu; envs.classic_control = cc
    gym.spaces, gym.logger, gym.utils, gym.error, gym.envs = spaces, logger, utils, error, envs
    for name, mod in (("gym", gym), ("gym.spaces", spaces), ("gym.logger", logger), ("gym.utils", utils),
                      ("gym.utils.seeding", seeding), ("gym.error", error), ("gym.envs", envs),
                      ("gym.envs.classic_control", cc), ("gym.envs.classic_control.utils", ccu)):
        sys.modules.setdefault(name, mod)


def g9():
    _gym_stand_in()
    from src.environments.continuous_cartpole import ContinuousCartPoleEnv
    from src.environments.adjustable_pendulum import AdjustablePendulumEnv
    out = {}
    rng = np.random.default_rng(909)
    # cart-pole: stepPhysics on explicit states (src/environments/continuous_cartpole.py:71-87) and a step() chain
    env = ContinuousCartPoleEnv()
    st = rng.uniform(-1, 1, (12, 4)) * np.array([2.0, 3.0, 0.6, 3.0])
    fc = rng.uniform(-30, 30, 12)
    out["cp_states"], out["cp_forces"] = st, fc
    out["cp_next"] = np.array([env.stepPhysics(float(f), tuple(s)) for s, f in zip(st, fc)])
    env.state = np.array([0.05, -0.1, 0.08, 0.2])
    acts = rng.uniform(-0.99, 0.99, 25)
    chain = []
    for a in acts:
        obs, rew, term, trunc, _ = env.step(np.array([a]))
        chain.append(np.array(obs, dtype=float))
    out["cp_chain_x0"], out["cp_chain_actions"], out["cp_chain"] = np.array([0.05, -0.1, 0.08, 0.2]), acts, np.array(chain)
    out["cp_params"] = np.array([env.gravity, env.masscart, env.masspole, env.length, env.force_mag, env.tau])
    # pendulum: step_static (src/environments/adjustable_pendulum.py:158-178) and step() (:135-156), incl. clipping
    opts = {"g": 9.81, "m": 1.0, "l": 1.0, "dt": 0.05, "max_torque": 2.0, "max_speed": 8.0}
    ps = rng.uniform(-1, 1, (12, 2)) * np.array([4.0, 9.0])
    pu = rng.uniform(-3, 3, (12, 1))
    out["pd_states"], out["pd_u"] = ps, pu
    out["pd_static_next"] = np.array([AdjustablePendulumEnv.step_static(s, u, opts) for s, u in zip(ps, pu)])
    out["pd_opts"] = np.array([opts[k] for k in ("g", "m", "l", "dt", "max_torque", "max_speed")])
    penv = AdjustablePendulumEnv(g=10.0, max_speed=8, max_torque=2.0, init_state={"th_init": np.pi, "thdot_init": 0.0})
    obs, _ = penv.reset()
    pacts = rng.uniform(-2.5, 2.5, (30, 1))
    pch, prew = [np.array(obs, dtype=float)], []
    for u in pacts:
        obs, r, _, _, _ = penv.step(u)
        pch.append(np.array(obs, dtype=float)); prew.append(float(r))
    out["pd_chain_actions"], out["pd_chain"], out["pd_chain_reward"] = pacts, np.array(pch), np.array(prew)
    # the data files of the README experiment (src/experiments/pretrain_uncertainty.py:86-88)
    for k in ("states", "actions", "next_states"):
        out["exp_" + k] = np.load(os.path.join(REF, "src", "experiments", "data", k + ".npy"))
    np.savez(os.path.join(OUT, "g9_closed_loop.npz"), **out)


def g10():
    """The reference's OWN regime (src/experiments/pretrain_uncertainty.py:86-121): its data files, one lambda = 0.5 for
    every GP, sigma_f = 1, Q = 2 I, R = 0, H = 6, curr_state = [4, -4], gamma in {-1, 1e-5}, at the experiment's
    sigma_n = 1e-5 and at 1e-3.  Every GP has the same hyper-parameters and the same X, so Ky_inv is ONE matrix per
    sigma_n (stored once).  Candidates: the zero plan, a plan along the data corridor, and seeded plans in the action box."""
    data = {k: np.load(os.path.join(REF, "src", "experiments", "data", k + ".npy")) for k in ("states", "actions", "next_states")}
    ds = da = 2
    H = 6
    Q, R = 2 * np.identity(2), np.zeros((2, 2))
    rng = np.random.default_rng(1010)
    U = np.stack([np.zeros((H, da)),
                  np.array([[0.0, 1.0]] * 4 + [[-1.0, 0.0]] * 2),               # up the corridor, then left
                  rng.uniform(-1, 1, (H, da)), 0.5 * rng.uniform(-1, 1, (H, da))])
    x0 = np.array([4.0, -4.0])
    gammas = [-1.0, 1e-5]
    out = {"X": np.concatenate((data["states"], data["actions"]), axis=1), "Y": data["next_states"], "x0": x0, "U": U,
           "Q": Q, "R": R, "gammas": np.array(gammas), "lambdas": np.full((ds, ds + da), 0.5), "sigma_f": np.ones(ds),
           "dims": np.array([data["states"].shape[0], ds, da, H]), "sigma_ns": np.array([1e-3, 1e-5])}
    for si, sn in enumerate(out["sigma_ns"]):
        mpc = RiskSensitiveMPC(gammas[0], H, ds, da, Q, R, None)
        for a in range(ds):
            g = mpc.dynamics.gpr_err[a]
            g.set_sigma_n(float(sn)); g.set_lambdas([0.5, 0.5, 0.5, 0.5]); g.set_sigma_f(1.)
        mpc.dynamics.append_train_data(data["states"], data["actions"], data["next_states"])
        mpc.set_xref(np.array([0., 0.])); mpc.set_uref(np.array([0., 0.]))
        k0 = mpc.dynamics.gpr_err[0].Ky_inv.detach().numpy()
        assert np.array_equal(k0, mpc.dynamics.gpr_err[1].Ky_inv.detach().numpy())
        # what the rollout reads (src/dynamics.py:170): exp of the float32 log the setter stored (src/gpr.py:51-60) -- NOT 0.5
        out["lambdas"] = np.stack([torch.exp(g.log_lambdas).detach().numpy() for g in mpc.dynamics.gpr_err])
        out["lambdas_set"] = np.full((ds, ds + da), 0.5)
        out[f"s{si}_Ky_inv"] = k0
        nb = U.shape[0]
        means = np.zeros((nb, H + 1, ds)); vars_ = np.zeros((nb, H + 1, ds))
        costs = np.zeros((len(gammas), nb)); grads = np.zeros((len(gammas), nb, H, da))
        mpc.curr_state = torch.tensor(x0).type(torch.float64)
        for b in range(nb):
            for gi, gm in enumerate(gammas):
                mpc.gamma = gm
                mpc.curr_cost = None
                costs[gi, b] = mpc.objective(U[b].reshape(-1).copy())
                grads[gi, b] = np.asarray(mpc.gradient(U[b].reshape(-1).copy())).reshape(H, da)
            sm, sc = mpc.dynamics.forward_propagate_torch(H, mpc.curr_state, torch.tensor(U[b]).type(torch.float64))
            means[b] = torch.stack(sm).detach().numpy()
            vars_[b] = torch.stack([torch.diag(s) for s in sc]).detach().numpy()
        out.update({f"s{si}_means": means, f"s{si}_vars": vars_, f"s{si}_costs": costs, f"s{si}_grads": grads})
    np.savez(os.path.join(OUT, "g10_readme_regime.npz"), **out)


def g11():
    """Reference-produced scalars at FULL size (N = 2048, the C3 training set of gaussian_process_mpc_amd/synth.py, seed
    1003): mean_prop_torch + variance_prop_torch (N^3 trace and all) of every GP for the first-step input of trajectory 0
    and for a later-step-like input with distinct state variances.  Inputs are re-derived from the seed by the tests; only
    the queries and the reference's outputs are stored (Ky_inv would be 4 x 33.5 MB)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from gaussian_process_mpc_amd.synth import synth_problem
    N, ds, da = 2048, 4, 1
    pb = synth_problem(3, N, ds, da, 20, 8)
    gps = []
    for a in range(ds):
        g = GaussianProcessRegression(ds + da)
        g.set_lambdas(pb["lambdas"][a]); g.set_sigma_f(1.0); g.set_sigma_n(float(pb["sigma_n"][a]))
        g.append_train_data(pb["X"], pb["Y"][:, a])
        gps.append(g)
    act_var = float(np.float32(1e-3))
    queries = [(np.concatenate((pb["x0"][0], pb["U"][0, 0])), np.diag([1e-3] * ds + [act_var])),
               (np.concatenate((0.7 * pb["x0"][1], pb["U"][1, 3])), np.diag([2.3e-3, 4.1e-3, 1.7e-3, 3.2e-3, act_var]))]
    out = {"u": np.stack([q[0] for q in queries]), "S": np.stack([q[1] for q in queries]),
           "mean": np.zeros((2, ds)), "var": np.zeros((2, ds)), "beta_sum": np.zeros(ds), "beta_abs_sum": np.zeros(ds),
           "dims": np.array([N, ds, da]), "seed": np.array([3])}
    for qi, (u, S) in enumerate(queries):
        for a, g in enumerate(gps):
            m, aux = mean_prop_torch(g.Ky_inv.detach(), torch.exp(g.log_lambdas).detach(), T(u), T(S), g.X_train, g.y_train.squeeze(),
                                     sigma_f=g.get_sigma_f())
            v = variance_prop_torch(g.Ky_inv.detach(), torch.exp(g.log_lambdas).detach(), T(u), T(S), g.X_train, m,
                                    aux["beta"], sigma_f=g.get_sigma_f())
            out["mean"][qi, a], out["var"][qi, a] = float(m), float(v)
            out["beta_sum"][a], out["beta_abs_sum"][a] = float(aux["beta"].sum()), float(aux["beta"].abs().sum())
    np.savez(os.path.join(OUT, "g11_fullsize_pin.npz"), **out)


def g12():
    """The WHOLE path at full size, produced by the reference: objective + gradient (rollout, cost, autograd backward) of two
    candidate plans at N = 2048, ds = 4, da = 1 (the C3 training set of gaussian_process_mpc_amd/synth.py, seed 1003), H = 2
    (a full-horizon reference run needs ~46 GiB), gamma = -1.  Inputs are re-derived from the seed by the tests."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from gaussian_process_mpc_amd.synth import synth_problem
    N, ds, da, H = 2048, 4, 1, 2
    pb = synth_problem(3, N, ds, da, 20, 8)
    mpc = RiskSensitiveMPC(-1.0, H, ds, da, pb["Q"], pb["R"], None)
    for a in range(ds):
        g = mpc.dynamics.gpr_err[a]
        g.set_lambdas(pb["lambdas"][a]); g.set_sigma_n(float(pb["sigma_n"][a])); g.set_sigma_f(1.0)
    mpc.dynamics.append_train_data(pb["X"][:, :ds], pb["X"][:, ds:], pb["Y"])
    out = {"dims": np.array([N, ds, da, H]), "seed": np.array([3]), "traj": np.array([0, 5]), "gamma": np.array([-1.0]),
           "means": np.zeros((2, H + 1, ds)), "vars": np.zeros((2, H + 1, ds)), "cost": np.zeros(2), "grad": np.zeros((2, H, da))}
    for k, b in enumerate(out["traj"]):
        mpc.curr_state = torch.tensor(pb["x0"][b]).type(torch.float64)
        mpc.curr_cost = None
        x = pb["U"][b, :H].reshape(-1).copy()
        out["cost"][k] = mpc.objective(x)
        out["grad"][k] = np.asarray(mpc.gradient(x)).reshape(H, da)
        sm, sc = mpc.dynamics.forward_propagate_torch(H, mpc.curr_state, torch.tensor(pb["U"][b, :H]).type(torch.float64))
        out["means"][k] = torch.stack(sm).detach().numpy()
        out["vars"][k] = torch.stack([torch.diag(c) for c in sc]).detach().numpy()
    np.savez(os.path.join(OUT, "g12_fullsize_rollout.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12):
        if only and fn.__name__ not in only:
            continue
        fn()
        print("wrote", fn.__name__)
