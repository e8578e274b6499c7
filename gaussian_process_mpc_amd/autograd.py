"""Autograd attachment of the HIP path: the reference returns graph-attached tensors from
``Dynamics.forward_propagate_torch`` (src/dynamics.py:126-191) and ``RiskSensitiveMPC.cost_torch``
(src/mpc.py:156-200), and its ``gradient`` is ``curr_cost.backward(retain_graph=True)`` (src/mpc.py:251).

The three ``torch.autograd.Function``s below give the mirror classes the same property without any
torch arithmetic on the path: each forward is one library call that also leaves the analytic
derivatives on the device, each backward is a contraction of them with the upstream gradient

* :class:`RolloutFunction`       ``gpmpc_rollout_jac`` / ``gpmpc_rollout_vjp`` (step Jacobians, reverse sweep)
* :class:`CostFunction`          ``gpmpc_cost_grad`` (d cost / d mu, d Sigma, d U)
* :class:`MomentMatchFunction`   ``gpmpc_moment_match`` with ``GPMPC_WANT_GRAD`` (d/du, d/dS of one step)

First derivatives only (the reference's solver path never differentiates twice).
"""
import ctypes

import torch
from torch.autograd.function import once_differentiable

from ._lib import check, lib, ptr, stream_ptr
from .rollout import moment_match


def _c(t):
    return t.detach().to(torch.float64).contiguous()


class RolloutFunction(torch.autograd.Function):
    """(x0 (B, ds), U (B, H, da)) -> means (B, H+1, ds), vars (B, H+1, ds)."""

    @staticmethod
    def forward(ctx, x0, U, pack):
        dev = pack.device
        x0c, Uc = _c(x0).to(dev), _c(U).to(dev)
        B, H, da = Uc.shape
        ds = pack.ds
        if da != pack.da or x0c.shape != (B, ds):
            raise ValueError("shape mismatch between pack, x0 and U")
        e = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)  # noqa: E731
        means, vars_, jac = e(B, H + 1, ds), e(B, H + 1, ds), e(B, H, 2 * ds, 2 * ds + da)
        ws = pack.workspace(lib().gpmpc_rollout_jac_workspace_bytes(pack.handle, B, H))
        with torch.cuda.device(dev):
            check(lib().gpmpc_rollout_jac(pack.handle, B, H, ptr(x0c), ptr(Uc), ptr(means), ptr(vars_), ptr(jac),
                                          ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()), "gpmpc_rollout_jac")
        ctx.save_for_backward(jac)
        ctx.dims = (B, H, ds, da, dev)
        ctx.need_x0 = x0.requires_grad
        return means, vars_

    @staticmethod
    @once_differentiable
    def backward(ctx, g_means, g_vars):
        (jac,) = ctx.saved_tensors
        B, H, ds, da, dev = ctx.dims
        gm = None if g_means is None else _c(g_means)
        gv = None if g_vars is None else _c(g_vars)
        gU = torch.empty((B, H, da), dtype=torch.float64, device=dev)
        gx0 = torch.empty((B, ds), dtype=torch.float64, device=dev) if ctx.need_x0 else None
        with torch.cuda.device(dev):
            check(lib().gpmpc_rollout_vjp(B, H, ds, da, ptr(jac), ptr(gm), ptr(gv), ptr(gU), ptr(gx0), stream_ptr()),
                  "gpmpc_rollout_vjp")
        return gx0, gU, None


class CostFunction(torch.autograd.Function):
    """(means (B, H+1, ds), covs (B, H+1, ds, ds), U (B, H, da)) -> cost (B,)."""

    @staticmethod
    def forward(ctx, means, covs, U, cost):
        dev = means.device
        m, c, u = _c(means), _c(covs), _c(U).to(dev)
        B, H1, ds = m.shape
        da = u.shape[2]
        out = torch.empty(B, dtype=torch.float64, device=dev)
        dm, dc, du = torch.empty_like(m), torch.empty_like(c), torch.empty_like(u)
        with torch.cuda.device(dev):
            check(lib().gpmpc_cost_grad(B, H1 - 1, ds, da, ctypes.byref(cost.c), ptr(m), ptr(c), ptr(u), ptr(out), ptr(dm),
                                        ptr(dc), ptr(du), stream_ptr()), "gpmpc_cost_grad")
        ctx.save_for_backward(dm, dc, du)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        dm, dc, du = ctx.saved_tensors
        return g[:, None, None] * dm, g[:, None, None, None] * dc, g[:, None, None] * du, None


class MomentMatchFunction(torch.autograd.Function):
    """(u (nq, D), S (nq, D, D)) -> mean (nq, ds), var (nq, ds): single-step moment matching of all GPs of a pack
    (mean_prop_torch / variance_prop_torch, src/tools/uncertainty_prop.py:296-399)."""

    @staticmethod
    def forward(ctx, u, S, pack):
        r = moment_match(pack, _c(u), _c(S), want_grad=True)
        ctx.save_for_backward(r["dmean_du"], r["dmean_dS"], r["dvar_du"], r["dvar_dS"])
        return r["mean"], r["var"]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_mean, g_var):
        dm_du, dm_dS, dv_du, dv_dS = ctx.saved_tensors
        gu = (g_mean[:, :, None] * dm_du).sum(1) + (g_var[:, :, None] * dv_du).sum(1)
        gS = (g_mean[:, :, None, None] * dm_dS).sum(1) + (g_var[:, :, None, None] * dv_dS).sum(1)
        return gu, gS, None


class CrossCovFunction(torch.autograd.Function):
    """(u (nq, D), S (nq, D, D)) -> q (nq,) = Cov[f_0, f_1] + mu_0 mu_1 = beta_0^T Qt beta_1 of a two-GP pack: the part of
    covariance_prop_torch (src/tools/uncertainty_prop.py:402-465) that does not go through the caller's mean1 / mean2 (the caller
    subtracts their product, so autograd treats them exactly as the reference's graph does).  Either cross-term form."""

    @staticmethod
    def forward(ctx, u, S, pack, bug_compatible):
        r = moment_match(pack, _c(u), _c(S), want_cov=True, want_grad=True, bug_compatible=bug_compatible)
        m0, m1 = r["mean"][:, 0], r["mean"][:, 1]
        # d q = d cov + mu_1 d mu_0 + mu_0 d mu_1
        dq_du = r["dcov_du"][:, 0, 1] + m1[:, None] * r["dmean_du"][:, 0] + m0[:, None] * r["dmean_du"][:, 1]
        dq_dS = r["dcov_dS"][:, 0, 1] + m1[:, None, None] * r["dmean_dS"][:, 0] + m0[:, None, None] * r["dmean_dS"][:, 1]
        ctx.save_for_backward(dq_du, dq_dS)
        return r["cov"][:, 0, 1] + m0 * m1

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        dq_du, dq_dS = ctx.saved_tensors
        return g[:, None] * dq_du, g[:, None, None] * dq_dS, None, None


def wants_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)
