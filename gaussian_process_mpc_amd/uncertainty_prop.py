"""Functional mirrors of the reference's single-step moment-matching functions
(src/tools/uncertainty_prop.py:296-465), evaluated by the HIP library.

The reference functions take raw tensors of ONE GP per call; here each call folds them into a
temporary device pack (an O(N^2) build, the same order as the evaluation itself).  Code that calls
them repeatedly for fixed data should build a :class:`GPPack` once and use :func:`moment_match`.
"""
import numpy as np
import torch

from .rollout import GPPack, moment_match


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)


def mean_prop_torch(Ky_inv, lambdas, u, S, X_train, y_train, sigma_f=1):
    """(mean, {'beta', 'l'}) -- src/tools/uncertainty_prop.py:296-338."""
    pack = GPPack(X_train, _np(y_train).reshape(-1, 1), _np(Ky_inv)[None], _np(lambdas)[None],
                  np.array([float(sigma_f)]))
    r = moment_match(pack, u, S, want_l=True)
    return r["mean"][0, 0], {"beta": pack.beta()[0], "l": r["l"][0, 0]}


def variance_prop_torch(Ky_inv, lambdas, u, S, X_train, mean, beta, sigma_f=1):
    """Predictive variance -- src/tools/uncertainty_prop.py:341-399.  ``mean`` is accepted for
    signature parity and recomputed on the device (it is a function of the other arguments)."""
    pack = GPPack(X_train, _np(beta).reshape(-1, 1), _np(Ky_inv)[None], _np(lambdas)[None],
                  np.array([float(sigma_f)]), y_is_beta=True)
    return moment_match(pack, u, S)["var"][0, 0]


def covariance_prop_torch(lambdas1, lambdas2, u, S, X_train, mean1, mean2, beta1, beta2, sigma_f1=1, sigma_f2=1,
                          bug_compatible=True):
    """Cov[f1, f2] -- src/tools/uncertainty_prop.py:402-465.  The reference's cross term (:446) is
    index-transposed; ``bug_compatible=True`` (default) reproduces it, ``False`` gives the consistent
    form that matches the reference's numpy ``covariance_prop``.  Only the betas enter (rank-one
    weights), so the pack is built without Ky_inv; mean1/mean2 are recomputed on the device."""
    B = np.stack((_np(beta1).reshape(-1), _np(beta2).reshape(-1)), axis=1)
    pack = GPPack(X_train, B, None, np.stack((_np(lambdas1), _np(lambdas2))),
                  np.array([float(sigma_f1), float(sigma_f2)]), y_is_beta=True)
    return moment_match(pack, u, S, want_cov=True, bug_compatible=bug_compatible)["cov"][0, 0, 1]
