"""CPU oracle (torch fp64) for the GP-MPC rollout path.  TEST INFRASTRUCTURE ONLY.

A restatement of the reference algorithm, written functionally over plain
tensors.  Every function cites the reference lines it follows
(paths relative to the upstream repository root).

Two evaluation modes for the variance term:

* ``mode="faithful"`` keeps the reference's op sequence and asymptotics: the
  per-call ``Ky_inv @ y`` GEMV, the per-call kernel-distance matrix and the
  dense ``(Ky_inv - beta beta^T) @ L`` product followed by ``trace``
  (src/tools/uncertainty_prop.py:399).  This is what ``bench.py`` times as the
  "reference CPU path".
* ``mode="o2"`` evaluates the same trace as the elementwise sum
  ``sum_ij W_ij L_ji`` (numerically equivalent, O(N^2)); used for larger parity
  cases and as the "algorithmic" CPU baseline.

Gradients come from torch autograd, exactly as in the reference
(src/mpc.py:251).
"""
from __future__ import annotations

import numpy as np
import torch

F64 = torch.float64

# src/dynamics.py:162 -- `1e-3 * torch.eye(action_dim)` is a float32 tensor that is
# promoted to float64 on concatenation, so the action-noise variance is the
# float32 rounding of 1e-3; src/dynamics.py:148 casts to float64 first, so the
# initial state variance is the float64 1e-3.
ACTION_NOISE_VAR = float(np.float32(1e-3))
INIT_STATE_VAR = 1e-3


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a), dtype=F64)


# --------------------------------------------------------------------------
# GP state (src/gpr.py)
# --------------------------------------------------------------------------
def effective_hyper(value):
    """Value a hyper-parameter actually takes after a reference setter
    (src/gpr.py:51-88): ``torch.log(torch.tensor(v)).type(float64)`` then ``exp``.
    ``torch.tensor`` of a Python float / list of floats is float32, so the log is
    taken in float32 (relative error ~1e-8); a float64 numpy array stays float64."""
    t = torch.tensor(value)
    return torch.exp(torch.log(t).type(F64)).numpy()


def scaled_sqdist(Xa, Xb, lambdas):
    """Squared distance in the 1/sqrt(lambda)-scaled space, via cdist as in
    src/gpr.py:167-168 / :272-275."""
    w = torch.sqrt(1.0 / lambdas)
    return torch.square(torch.cdist(Xa * w, Xb * w, p=2))


def kernel_matrices(X, lambdas, sigma_f, sigma_n):
    """Kf, Ky, Ky_inv exactly as GaussianProcessRegression.build_Ky_inv_mat
    (src/gpr.py:159-171): explicit inverse, no Cholesky."""
    X, lambdas = _t(X), _t(lambdas)
    Kf = (sigma_f ** 2) * torch.exp(-0.5 * scaled_sqdist(X, X, lambdas))
    # :170 multiplies a 0-dim float64 tensor with a float32 `torch.eye`: the product is
    # float32, so the noise variance on the diagonal is float32(sigma_n^2).
    noise = torch.as_tensor(sigma_n, dtype=F64) ** 2 * torch.eye(X.shape[0])
    assert noise.dtype == torch.float32
    Ky = Kf + noise
    return Kf, Ky, torch.linalg.inv(Ky)


def cross_kernel(X_pred, X, lambdas, sigma_f):
    """K(X*, X) as compute_pred_train_covariance (src/gpr.py:253-283).
    2-D input -> (p, N) by cdist; 1-D input -> (N,) evaluated pointwise."""
    X_pred, X, lambdas = _t(X_pred), _t(X), _t(lambdas)
    if X_pred.dim() == 2:
        return (sigma_f ** 2) * torch.exp(-0.5 * scaled_sqdist(X_pred, X, lambdas))
    diff = X_pred[None, :] - X
    return (sigma_f ** 2) * torch.exp(-0.5 * torch.sum(diff * diff / lambdas, dim=1))


def predict(X_pred, X, y, Ky_inv, lambdas, sigma_f, sigma_n, covar=False, targets=False):
    """predict_latent_vars without nominal model (src/gpr.py:285-332).
    Returns numpy (p,1) mean and (p,p) covariance or None."""
    X_pred, X, y, Ky_inv, lambdas = _t(X_pred), _t(X), _t(y), _t(Ky_inv), _t(lambdas)
    y = y.reshape(-1, 1)
    Ks = cross_kernel(X_pred, X, lambdas, sigma_f)
    mean = Ks @ Ky_inv @ y
    if not covar:
        return mean.numpy(), None
    Kss = (sigma_f ** 2) * torch.exp(-0.5 * scaled_sqdist(X_pred, X_pred, lambdas))
    cov = Kss - Ks @ Ky_inv @ Ks.mT
    if targets:
        p = X_pred.shape[0] if X_pred.dim() == 2 else 1
        cov = cov + (sigma_n ** 2) * torch.eye(p, dtype=F64)
    return mean.numpy(), cov.numpy()


# --------------------------------------------------------------------------
# Marginal likelihood and hyper-parameter training (src/gpr.py:240-251, 334-370)
# --------------------------------------------------------------------------
def marginal_likelihood(X, y, lambdas, sigma_f, sigma_n, nominal=None):
    """compute_marginal_likelihood (src/gpr.py:240-251) on matrices built as build_Ky_inv_mat builds
    them (:159-171): -1/2 r^T Ky_inv r - 1/2 log(det(Ky)) - N/2 log(2 pi), r = y - f_nom(X).
    Returns a (1,1) tensor (graph attached when the hyper-parameters require grad)."""
    X, y = _t(X), _t(y).reshape(-1, 1)
    r = y if nominal is None else y - _t(nominal).reshape(-1, 1)
    _, Ky, Ky_inv = kernel_matrices(X, lambdas, sigma_f, sigma_n)
    return (-0.5 * r.mT @ Ky_inv @ r - 0.5 * torch.log(torch.linalg.det(Ky))
            - X.shape[0] / 2 * np.log(2 * np.pi))


class HyperTrainer:
    """update_hyperparams (src/gpr.py:334-370) with the optimiser of the constructor (:46-49):
    Adam(lr=0.1, betas=(0.9, 0.999), maximize=True) over [log_lambdas, log_sigma_n, log_sigma_f],
    gradients by autograd through inv / det; one iteration = likelihood at the current matrices,
    backward, Adam step, rebuild; stops when every |d ml / d log-hyper| < 1e-5 (:366-370).
    ``step()`` returns what the reference prints per iteration: the likelihood BEFORE the step, the
    gradients, and the log-hypers AFTER the step."""

    def __init__(self, X, y, x_dim, nominal=None, log_lambdas=None, log_sigma_f=0.0, log_sigma_n=0.0):
        self.X, self.y, self.nominal = _t(X), _t(y), nominal
        ll = np.zeros(x_dim) if log_lambdas is None else np.asarray(log_lambdas, dtype=float)
        self.log_lambdas = torch.tensor(ll, dtype=F64).requires_grad_()
        self.log_sigma_n = torch.tensor(float(log_sigma_n), dtype=F64).requires_grad_()
        self.log_sigma_f = torch.tensor(float(log_sigma_f), dtype=F64).requires_grad_()
        self.optimizer = torch.optim.Adam(params=[self.log_lambdas, self.log_sigma_n, self.log_sigma_f],
                                          lr=0.1, betas=(0.9, 0.999), maximize=True)

    def likelihood(self):
        return marginal_likelihood(self.X, self.y, torch.exp(self.log_lambdas), torch.exp(self.log_sigma_f),
                                   torch.exp(self.log_sigma_n), self.nominal)

    def step(self):
        self.optimizer.zero_grad()
        ml = self.likelihood()
        ml.backward()
        self.optimizer.step()
        g = {"log_lambdas": self.log_lambdas.grad.detach().numpy().copy(),
             "log_sigma_f": self.log_sigma_f.grad.item(), "log_sigma_n": self.log_sigma_n.grad.item()}
        return {"ml": ml.item(), "grad": g,
                "log_lambdas": self.log_lambdas.detach().numpy().copy(),
                "log_sigma_f": self.log_sigma_f.item(), "log_sigma_n": self.log_sigma_n.item()}

    def run(self, num_iters=1000):
        hist = []
        for _ in range(num_iters):
            h = self.step()
            hist.append(h)
            g = h["grad"]
            if (np.abs(g["log_lambdas"]) < 1e-5).all() and abs(g["log_sigma_f"]) < 1e-5 and abs(g["log_sigma_n"]) < 1e-5:
                break
        return hist


# --------------------------------------------------------------------------
# Exact moment matching (src/tools/uncertainty_prop.py:296-465)
# --------------------------------------------------------------------------
def mean_prop(Ky_inv, lambdas, u, S, X, y, sigma_f=1.0):
    """mean_prop_torch (src/tools/uncertainty_prop.py:296-338).
    Returns (mu, beta, l)."""
    beta = Ky_inv.flatten() * y if y.dim() == 0 else Ky_inv @ y      # :324-327
    D = S.shape[0]
    B = torch.linalg.inv(S + torch.diag(lambdas))                   # :331
    diff = u - X
    quad = torch.sum((diff @ B) * diff, dim=1)                      # :334
    det = torch.linalg.det(torch.diag(1.0 / lambdas) @ S + torch.eye(D, dtype=F64))
    l = (det ** (-0.5)) * torch.exp(-0.5 * quad) * sigma_f ** 2     # :335-336
    return torch.dot(beta, l), beta, l


def _pair_matrix_L(lambdas, u, S, X, sigma_f):
    """The N x N matrix L of variance_prop_torch (:372-397), built with the
    reference's expanded quadratic form (u A u + X A X^T - uAX_i - uAX_j)."""
    D = S.shape[0]
    A = torch.linalg.inv(torch.diag(lambdas) / 2 + S)               # :376
    det = torch.linalg.det(2 * torch.diag(1.0 / lambdas) @ S + torch.eye(D, dtype=F64)) ** (-0.5)
    uAX = (u @ A @ X.mT)[:, None]                                   # :380
    G = u @ A @ u + X @ A @ X.mT - uAX - uAX.mT                     # :383
    g = torch.diag(G)[:, None]                                      # :386
    A_part = torch.exp(-0.125 * (g + 2 * G + g.mT))                 # :389
    Lam_part = torch.exp(-0.25 * scaled_sqdist(X, X, lambdas))      # :392-394
    return det * A_part * Lam_part * sigma_f ** 4                   # :397


def variance_prop(Ky_inv, lambdas, u, S, X, mean, beta, sigma_f=1.0, mode="faithful"):
    """variance_prop_torch (src/tools/uncertainty_prop.py:341-399).
    No clamp: the result may be negative, as in the reference."""
    L = _pair_matrix_L(lambdas, u, S, X, sigma_f)
    W = Ky_inv - torch.outer(beta, beta)
    if mode == "faithful":
        tr = torch.trace(W @ L)                                     # :399 (2 N^3 flops)
    else:
        tr = torch.sum(W * L.mT)                                    # same trace, O(N^2)
    return sigma_f ** 2 - tr - mean ** 2


def covariance_prop(lam1, lam2, u, S, X, mean1, mean2, beta1, beta2,
                    sigma_f1=1.0, sigma_f2=1.0, bug_compatible=True):
    """covariance_prop_torch (src/tools/uncertainty_prop.py:402-465).

    ``bug_compatible=True`` reproduces the reference's cross term
    ``z2^T A z1`` (:446), which is index-transposed relative to the row/column
    terms and is exact only for diagonal S or proportional lambdas.
    ``bug_compatible=False`` uses ``z1^T A z2``, which agrees with the numpy
    double loop ``covariance_prop`` (:187-237)."""
    D = lam1.shape[0]
    Li1, Li2 = torch.diag(1.0 / lam1), torch.diag(1.0 / lam2)
    R = S @ (Li1 + Li2) + torch.eye(D, dtype=F64)
    det = torch.linalg.det(R) ** (-0.5)                             # :437
    z1 = Li1 @ (X - u).mT                                           # :439
    z2 = Li2 @ (X - u).mT
    Am = torch.linalg.inv(R) @ S                                    # :441
    q1 = torch.sum((Am @ z1) * z1, dim=0)[:, None]
    q2 = torch.sum((Am @ z2) * z2, dim=0)[:, None]
    cross = z2.mT @ Am @ z1 if bug_compatible else z1.mT @ Am @ z2  # :444
    expo = torch.exp(0.5 * (q1 + 2 * cross + q2.mT))
    k1 = scaled_sqdist(X, u[None, :], lam1)                         # :450-453
    k2 = scaled_sqdist(X, u[None, :], lam2)                         # :455-458
    cov_part = torch.exp(-0.5 * (k1 + k2.mT)) * sigma_f1 ** 2 * sigma_f2 ** 2
    return beta1 @ (det * cov_part * expo) @ beta2 - mean1 * mean2   # :460-463


# numpy double-loop formulas (src/tools/uncertainty_prop.py:6-44, 91-136, 187-237);
# sigma_f = 1 only, small N only.
def mean_prop_loops(Ky, lambdas, u, S, X, y):
    beta = np.linalg.solve(Ky, y)
    B = np.linalg.inv(S + np.diag(lambdas))
    det = np.linalg.det(np.diag(1 / lambdas) @ S + np.eye(len(u))) ** (-0.5)
    l = np.array([det * np.exp(-0.5 * (u - x) @ B @ (u - x)) for x in X])
    return float(beta @ l), beta, l


def variance_prop_loops(Ky, lambdas, u, S, X, y):
    mu, beta, _ = mean_prop_loops(Ky, lambdas, u, S, X, y)
    N = X.shape[0]
    A = np.linalg.inv(np.diag(lambdas) / 2 + S)
    Li = np.diag(1 / lambdas)
    det = np.linalg.det(2 * Li @ S + np.eye(len(u))) ** (-0.5)
    L = np.empty((N, N))
    for i in range(N):
        for j in range(N):
            xm = 0.5 * (X[i] + X[j])
            dx = X[i] - X[j]
            L[i, j] = det * np.exp(-0.5 * (u - xm) @ A @ (u - xm) - 0.25 * dx @ Li @ dx)
    return float(1 - np.trace((np.linalg.inv(Ky) - np.outer(beta, beta)) @ L) - mu ** 2)


def covariance_prop_loops(Ky1, Ky2, lam1, lam2, u, S, X, y1, y2=None):
    """Reference double loop (:187-237).  The reference passes ONE y_train to
    both GPs; ``y2`` defaults to ``y1`` to mirror that."""
    y2 = y1 if y2 is None else y2
    m1, b1, _ = mean_prop_loops(Ky1, lam1, u, S, X, y1)
    m2, b2, _ = mean_prop_loops(Ky2, lam2, u, S, X, y2)
    Li1, Li2 = np.diag(1 / lam1), np.diag(1 / lam2)
    D = len(u)
    R = S @ (Li1 + Li2) + np.eye(D)
    det = np.linalg.det(R) ** (-0.5)
    RiS = np.linalg.inv(R) @ S
    N = X.shape[0]
    Qt = np.empty((N, N))
    for i in range(N):
        ki = np.exp(-0.5 * (X[i] - u) @ Li1 @ (X[i] - u))
        for j in range(N):
            kj = np.exp(-0.5 * (X[j] - u) @ Li2 @ (X[j] - u))
            z = Li1 @ (X[i] - u) + Li2 @ (X[j] - u)
            Qt[i, j] = ki * kj * det * np.exp(0.5 * z @ RiS @ z)
    return float(b1 @ Qt @ b2 - m1 * m2)


# --------------------------------------------------------------------------
# Rollout (src/dynamics.py:126-191) and cost (src/mpc.py:156-200)
# --------------------------------------------------------------------------
class GPBundle:
    """Plain container for the state of ``state_dim`` GPs sharing X
    (src/dynamics.py:33-37, src/gpr.py:24-40)."""

    def __init__(self, X, Y, lambdas, sigma_f, sigma_n, Ky_inv=None):
        self.X = _t(X)                                  # (N, D)
        self.Y = _t(Y)                                  # (N, ds), column a = targets of GP a
        self.lambdas = _t(lambdas)                      # (ds, D)
        self.sigma_f = [float(v) for v in np.atleast_1d(sigma_f)]
        self.sigma_n = [float(v) for v in np.atleast_1d(sigma_n)]
        self.ds = self.Y.shape[1]
        self.D = self.X.shape[1]
        self.da = self.D - self.ds
        if Ky_inv is None:
            Ky_inv = torch.stack([kernel_matrices(self.X, self.lambdas[a], self.sigma_f[a],
                                                  self.sigma_n[a])[2] for a in range(self.ds)])
        self.Ky_inv = _t(Ky_inv)                        # (ds, N, N)


def forward_propagate(gp: GPBundle, horizon, x0, U, mode="faithful"):
    """Dynamics.forward_propagate_torch (src/dynamics.py:126-191): shooting
    rollout of means and DIAGONAL covariances.  Returns lists like the
    reference (H+1 tensors (ds,), H+1 tensors (ds,ds))."""
    means = [_t(x0)]
    covs = [INIT_STATE_VAR * torch.eye(gp.ds, dtype=F64)]           # :148
    for t in range(1, horizon + 1):
        u = torch.cat((means[t - 1], U[t - 1, :]))                  # :154
        S = torch.zeros((gp.D, gp.D), dtype=F64)                    # :159-163
        S[:gp.ds, :gp.ds] = covs[t - 1]
        S = S + torch.diag(torch.cat((torch.zeros(gp.ds, dtype=F64),
                                      torch.full((gp.da,), ACTION_NOISE_VAR, dtype=F64))))
        mu_t, var_t = [], []
        for a in range(gp.ds):                                      # :166-181
            m, beta, _ = mean_prop(gp.Ky_inv[a], gp.lambdas[a], u, S, gp.X, gp.Y[:, a], gp.sigma_f[a])
            v = variance_prop(gp.Ky_inv[a], gp.lambdas[a], u, S, gp.X, m, beta, gp.sigma_f[a], mode)
            mu_t.append(m)
            var_t.append(v)
        means.append(torch.stack(mu_t))                             # :185-186
        covs.append(torch.diag(torch.stack(var_t)))                 # :188-189
    return means, covs


def forward_propagate_fullcov(gp: GPBundle, horizon, x0, U, mode="o2"):
    """Extension oracle for BASELINE config 5: the rollout of forward_propagate (src/dynamics.py:126-191) with the
    off-diagonal covariances the reference leaves as a TODO (:184), filled in with covariance_prop_torch in its
    consistent form (src/tools/uncertainty_prop.py:402-465; the reference's own numpy covariance_prop :187-237)."""
    means = [_t(x0)]
    covs = [INIT_STATE_VAR * torch.eye(gp.ds, dtype=F64)]
    for t in range(1, horizon + 1):
        u = torch.cat((means[t - 1], U[t - 1, :]))
        S = torch.zeros((gp.D, gp.D), dtype=F64)
        S[:gp.ds, :gp.ds] = covs[t - 1]
        S = S + torch.diag(torch.cat((torch.zeros(gp.ds, dtype=F64), torch.full((gp.da,), ACTION_NOISE_VAR, dtype=F64))))
        mu_t, beta_t = [], []
        rows = [[None] * gp.ds for _ in range(gp.ds)]
        for a in range(gp.ds):
            m, beta, _ = mean_prop(gp.Ky_inv[a], gp.lambdas[a], u, S, gp.X, gp.Y[:, a], gp.sigma_f[a])
            rows[a][a] = variance_prop(gp.Ky_inv[a], gp.lambdas[a], u, S, gp.X, m, beta, gp.sigma_f[a], mode)
            mu_t.append(m)
            beta_t.append(beta)
        for a in range(gp.ds):
            for b in range(a + 1, gp.ds):
                c = covariance_prop(gp.lambdas[a], gp.lambdas[b], u, S, gp.X, mu_t[a], mu_t[b], beta_t[a], beta_t[b],
                                    gp.sigma_f[a], gp.sigma_f[b], bug_compatible=False)
                rows[a][b] = c
                rows[b][a] = c
        means.append(torch.stack(mu_t))
        covs.append(torch.stack([torch.stack(r) for r in rows]))
    return means, covs


def objective_and_gradient_fullcov(gp, horizon, x0, U, x_ref, u_ref, Q, R, gamma, R_delta=None, last_u=None,
                                   mode="o2", want_grad=True):
    """cost_torch (src/mpc.py:156-200) on the full-covariance rollout, gradient by autograd."""
    Ut = _t(U).clone().reshape(horizon, -1).requires_grad_(want_grad)
    means, covs = forward_propagate_fullcov(gp, horizon, x0, Ut, mode)
    if gamma == 0:
        c = cost_risk_neutral(means, Ut, covs, _t(x_ref), _t(u_ref), Q, R, R_delta, last_u)
    else:
        c = cost(means, Ut, covs, _t(x_ref), _t(u_ref), Q, R, gamma, R_delta, last_u)
    out = {"cost": float(c.item()), "means": torch.stack([m.detach() for m in means]).numpy(),
           "covs": torch.stack([s.detach() for s in covs]).numpy()}
    if want_grad:
        c.backward()
        out["grad"] = Ut.grad.detach().numpy().copy()
    return out


def cost(means, U, covs, x_ref, u_ref, Q, R, gamma, R_delta=None, last_u=None):
    """RiskSensitiveMPC.cost_torch (src/mpc.py:156-200).  gamma != 0."""
    Q, R = _t(Q), _t(R)
    ds = Q.shape[0]
    H = U.shape[0]
    Q_inv = torch.linalg.inv(Q)                                     # :179
    total = 0
    for i in range(H + 1):                                          # :182-185
        total = total + (1 / gamma) * torch.log(torch.linalg.det(torch.eye(ds, dtype=F64) + gamma * Q @ covs[i]))
        e = means[i] - x_ref
        total = total + e @ torch.linalg.inv(Q_inv + gamma * covs[i]) @ e
    for j in range(H):                                              # :188-189
        d = U[j, :] - u_ref
        total = total + d @ R @ d
    if R_delta is not None:                                         # :191-198
        Rd = _t(R_delta)
        prev = _t(last_u).reshape(1, -1)
        dU = torch.diff(torch.cat((prev, U), dim=0), dim=0)
        for j in range(H):
            total = total + dU[j, :] @ Rd @ dU[j, :]
    return total


def cost_risk_neutral(means, U, covs, x_ref, u_ref, Q, R, R_delta=None, last_u=None):
    """gamma -> 0 limit of the cost (the reference cannot evaluate gamma = 0,
    src/mpc.py:183): tr(Q Sigma) + e^T Q e.  Build extension, oracled here."""
    Q, R = _t(Q), _t(R)
    H = U.shape[0]
    total = 0
    for i in range(H + 1):
        e = means[i] - x_ref
        total = total + torch.trace(Q @ covs[i]) + e @ Q @ e
    for j in range(H):
        d = U[j, :] - u_ref
        total = total + d @ R @ d
    if R_delta is not None:
        Rd = _t(R_delta)
        dU = torch.diff(torch.cat((_t(last_u).reshape(1, -1), U), dim=0), dim=0)
        for j in range(H):
            total = total + dU[j, :] @ Rd @ dU[j, :]
    return total


def objective_and_gradient(gp, horizon, x0, U, x_ref, u_ref, Q, R, gamma,
                           R_delta=None, last_u=None, mode="faithful", want_grad=True):
    """objective + gradient callback pair (src/mpc.py:202-255): rollout, cost,
    autograd backward.  Returns dict(cost, grad(H,da), means(H+1,ds), vars(H+1,ds))."""
    Ut = _t(U).clone().reshape(horizon, -1).requires_grad_(want_grad)   # :217-218
    means, covs = forward_propagate(gp, horizon, x0, Ut, mode)          # :221
    if gamma == 0:
        c = cost_risk_neutral(means, Ut, covs, _t(x_ref), _t(u_ref), Q, R, R_delta, last_u)
    else:
        c = cost(means, Ut, covs, _t(x_ref), _t(u_ref), Q, R, gamma, R_delta, last_u)  # :222
    out = {
        "cost": float(c.item()),
        "means": torch.stack([m.detach() for m in means]).numpy(),
        "vars": torch.stack([torch.diag(s).detach() for s in covs]).numpy(),
    }
    if want_grad:
        c.backward()                                                    # :251
        out["grad"] = Ut.grad.detach().numpy().copy()                   # :253
    return out
