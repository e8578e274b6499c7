"""Functional mirrors of the reference's single-step moment-matching functions
(src/tools/uncertainty_prop.py:296-465), evaluated by the HIP library.

The reference functions take raw tensors of ONE GP per call; here the constant part of a call
(beta, the folded weight matrix: an O(N^2) build) lives in a device pack.  The pack of the last few
distinct argument sets is kept, keyed by the OBJECTS handed in and their autograd versions, so the
reference's own calling pattern -- ``mean_prop_torch`` then ``variance_prop_torch`` on the same
``Ky_inv`` / ``X_train`` for every step of a rollout (src/dynamics.py:166-183) -- builds it once.
numpy arguments are not cached (no version counter to detect in-place edits): they build per call.
Like the reference's, the results carry an autograd graph when ``u`` or ``S`` requires grad.
"""
import numpy as np
import torch

from .autograd import CrossCovFunction, MomentMatchFunction, wants_grad
from .rollout import GPPack, moment_match


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)


_PACKS = []          # [(key objects, versions, extra, pack)], most recent last
_PACKS_MAX = 4


def clear_pack_cache():
    """Drop the device packs kept for the functional interface (each holds the N x N weights of one GP)."""
    _PACKS.clear()


def _cached_pack(objs, extra, build):
    """Pack for the tensors `objs` (identity + version) and the hashable `extra`; `build()` makes it."""
    if not all(isinstance(o, torch.Tensor) for o in objs):
        return build()
    vers = tuple(o._version for o in objs)
    for q, (ko, kv, ke, pack) in enumerate(_PACKS):
        if len(ko) == len(objs) and all(a is b for a, b in zip(ko, objs)) and kv == vers and ke == extra:
            _PACKS.append(_PACKS.pop(q))
            return pack
    pack = build()
    _PACKS.append((tuple(objs), vers, extra, pack))      # the key holds the references: ids cannot be reused
    if len(_PACKS) > _PACKS_MAX:
        _PACKS.pop(0)
    return pack


def _match(pack, u, S, **kw):
    """moment_match, through the autograd Function when u / S carry a graph."""
    if wants_grad(u, S) and not kw:
        dev = pack.device
        ud = torch.as_tensor(u).to(dev, torch.float64).reshape(1, pack.D)
        Sd = torch.as_tensor(S).to(dev, torch.float64).reshape(1, pack.D, pack.D)
        mean, var = MomentMatchFunction.apply(ud, Sd, pack)
        return {"mean": mean, "var": var}
    return moment_match(pack, u, S, **kw)


def mean_prop_torch(Ky_inv, lambdas, u, S, X_train, y_train, sigma_f=1):
    """(mean, {'beta', 'l'}) -- src/tools/uncertainty_prop.py:296-338."""
    lam = _np(lambdas)
    pack = _cached_pack((Ky_inv, X_train, y_train), ("y", lam.tobytes(), float(sigma_f)),
                        lambda: GPPack(X_train, _np(y_train).reshape(-1, 1), _np(Ky_inv)[None], lam[None],
                                       np.array([float(sigma_f)])))
    r = moment_match(pack, u, S, want_l=True)
    mean = _match(pack, u, S)["mean"][0, 0] if wants_grad(u, S) else r["mean"][0, 0]
    return mean, {"beta": pack.beta()[0], "l": r["l"][0, 0]}


def variance_prop_torch(Ky_inv, lambdas, u, S, X_train, mean, beta, sigma_f=1):
    """Predictive variance ``sigma_f^2 - tr((Ky_inv - beta beta^T) L) - mean^2`` -- src/tools/uncertainty_prop.py:341-399.
    ``mean`` enters as the reference uses it (:399): the device evaluates the variance with the mean of its own O(N)
    sum, and a caller-supplied ``mean`` that differs from it replaces that term."""
    lam = _np(lambdas)
    pack = _cached_pack((Ky_inv, X_train, beta), ("beta", lam.tobytes(), float(sigma_f)),
                        lambda: GPPack(X_train, _np(beta).reshape(-1, 1), _np(Ky_inv)[None], lam[None],
                                       np.array([float(sigma_f)]), y_is_beta=True))
    r = _match(pack, u, S)
    var, mu = r["var"][0, 0], r["mean"][0, 0]
    m = torch.as_tensor(mean).to(var.device, torch.float64).reshape(())
    return torch.where(m == mu, var, var + (mu * mu - m * m))


def covariance_prop_torch(lambdas1, lambdas2, u, S, X_train, mean1, mean2, beta1, beta2, sigma_f1=1, sigma_f2=1,
                          bug_compatible=True):
    """Cov[f1, f2] -- src/tools/uncertainty_prop.py:402-465.  The reference's cross term (:446) is
    index-transposed; ``bug_compatible=True`` (default) reproduces it, ``False`` gives the consistent
    form that matches the reference's numpy ``covariance_prop``.  Only the betas enter (rank-one
    weights), so the pack is built without Ky_inv.  ``mean1 * mean2`` is subtracted as given (:465).  When ``u``, ``S``,
    ``mean1`` or ``mean2`` requires grad the result carries the reference's graph (``autograd.CrossCovFunction``: analytic
    d/du, d/dS of beta1^T Qt beta2 from the device, minus the product of the caller's means in torch)."""
    l1, l2 = _np(lambdas1), _np(lambdas2)
    pack = _cached_pack((X_train, beta1, beta2), ("cov", l1.tobytes(), l2.tobytes(), float(sigma_f1), float(sigma_f2)),
                        lambda: GPPack(X_train, np.stack((_np(beta1).reshape(-1), _np(beta2).reshape(-1)), axis=1), None,
                                       np.stack((l1, l2)), np.array([float(sigma_f1), float(sigma_f2)]), y_is_beta=True))
    if wants_grad(u, S, mean1, mean2):
        # the graph of the reference's expression: beta1^T Qt beta2 as a function of (u, S) on the device, the product of the CALLER's
        # means subtracted in torch -- attached means carry their own graph, detached ones are constants, as in the reference (:465)
        dev = pack.device
        ud = torch.as_tensor(u).to(dev, torch.float64).reshape(1, pack.D)
        Sd = torch.as_tensor(S).to(dev, torch.float64).reshape(1, pack.D, pack.D)
        q = CrossCovFunction.apply(ud, Sd, pack, bool(bug_compatible))[0]
        m1 = torch.as_tensor(mean1).to(dev, torch.float64).reshape(())
        m2 = torch.as_tensor(mean2).to(dev, torch.float64).reshape(())
        return q - m1 * m2
    r = moment_match(pack, u, S, want_cov=True, bug_compatible=bug_compatible)
    # the reference subtracts the product of the means it is GIVEN (:465): beta1^T Qt beta2 from the device, the caller's means here
    q = r["cov"][0, 0, 1] + r["mean"][0, 0] * r["mean"][0, 1]
    m1 = torch.as_tensor(mean1).detach().to(q.device, torch.float64).reshape(())
    m2 = torch.as_tensor(mean2).detach().to(q.device, torch.float64).reshape(())
    return q - m1 * m2
