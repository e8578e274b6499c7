"""The boundary is differentiable like the reference's: forward_propagate_torch / cost_torch return
graph-attached tensors (reference src/dynamics.py:126-191, src/mpc.py:156-200) and the reference's own
pattern -- rollout -> cost_torch -> backward() -> u.grad (src/mpc.py:217-255) -- runs on the mirror classes.

The graph's nodes are HIP calls (autograd.RolloutFunction / CostFunction / MomentMatchFunction); every gradient here is
checked against the reference's own autograd outputs stored in the golden fixtures (g2: single step, g3 / g4: whole
objective), against RiskSensitiveMPC.gradient, and against finite differences of the HIP forward.
"""
import numpy as np
import pytest
import torch

from test_gpu_api import _mpc_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gaussian_process_mpc_amd as g
    g.require_gpu()
    return g


@pytest.mark.parametrize("name", ["g3_rollout_c1.npz", "g4_rollout_c2.npz"])
def test_reference_pattern_rollout_cost_backward(G, golden, name):
    """src/mpc.py:217-255 verbatim on the mirror: u.requires_grad_(), forward_propagate_torch, cost_torch, backward."""
    z = golden(name)
    N, ds, da, H = (int(v) for v in z["dims"])
    for gi, gamma in enumerate(z["gammas"]):
        mpc = _mpc_from(G, z, float(gamma))
        for b in range(z["x0"].shape[0]):
            x = z["U"][b].reshape(-1).copy()
            mpc.curr_state = torch.tensor(z["x0"][b]).type(torch.float64).to(mpc.device)
            # -- the reference's objective(), line by line
            u = torch.as_tensor(x.reshape((H, da)), device=mpc.device).type(torch.float64).requires_grad_(True)
            state_means, state_covars = mpc.dynamics.forward_propagate_torch(H, mpc.curr_state, u)
            assert state_means[1].requires_grad and state_covars[1].requires_grad       # graph attached
            cost = mpc.cost_torch(state_means, u, state_covars, mpc.x_ref, mpc.u_ref)
            assert cost.dim() == 0 and cost.requires_grad
            # -- the reference's gradient()
            cost.backward(retain_graph=True)
            grad = u.grad.cpu().numpy()
            assert grad.shape == (H, da)
            np.testing.assert_allclose(cost.item(), z["costs"][gi, b], rtol=1e-6)
            np.testing.assert_allclose(grad, z["grads"][gi, b], rtol=1e-4, atol=1e-7)           # the reference's autograd
            mpc._cache_key = None
            np.testing.assert_allclose(grad, mpc.gradient(x), rtol=1e-9, atol=1e-12)            # the fused adjoint
            np.testing.assert_allclose(cost.item(), mpc.objective(x), rtol=1e-12)
            # retain_graph=True: a second backward accumulates, as torch does
            cost.backward()
            np.testing.assert_allclose(u.grad.cpu().numpy(), 2 * grad, rtol=1e-12, atol=0)


def test_graph_through_means_and_covariances_separately(G, golden):
    """Arbitrary downstream torch code on the returned lists: d/dU of a function of one mean and one covariance entry
    against central differences of the HIP forward; d/d curr_state too."""
    z = golden("g4_rollout_c2.npz")
    N, ds, da, H = (int(v) for v in z["dims"])
    mpc = _mpc_from(G, z, -1.0)
    dyn = mpc.dynamics
    x0 = torch.tensor(z["x0"][0], dtype=torch.float64, device=mpc.device).requires_grad_(True)
    U = torch.tensor(z["U"][0], dtype=torch.float64, device=mpc.device).requires_grad_(True)

    def f(x0_, U_):
        means, covs = dyn.forward_propagate_torch(H, x0_, U_)
        return (means[H] * means[3]).sum() + 50.0 * covs[H][1, 1] * means[2][0] + covs[5].diagonal().sum()

    val = f(x0, U)
    val.backward()
    gU, gx = U.grad.cpu().numpy().copy(), x0.grad.cpu().numpy().copy()
    eps = 1e-4
    with torch.no_grad():
        for (t, k) in [(0, 0), (3, 0), (H - 1, da - 1), (H // 2, 0)]:
            Up, Um = U.detach().clone(), U.detach().clone()
            Up[t, k] += eps
            Um[t, k] -= eps
            fd = (f(x0.detach(), Up).item() - f(x0.detach(), Um).item()) / (2 * eps)
            assert abs(fd - gU[t, k]) <= 2e-5 * max(1.0, abs(fd)), (t, k, fd, gU[t, k])
        for k in range(ds):
            xp, xm = x0.detach().clone(), x0.detach().clone()
            xp[k] += eps
            xm[k] -= eps
            fd = (f(xp, U.detach()).item() - f(xm, U.detach()).item()) / (2 * eps)
            assert abs(fd - gx[k]) <= 2e-5 * max(1.0, abs(fd)), (k, fd, gx[k])


def test_no_grad_inputs_take_the_plain_path_and_agree(G, golden):
    z = golden("g3_rollout_c1.npz")
    N, ds, da, H = (int(v) for v in z["dims"])
    mpc = _mpc_from(G, z, -1.0)
    x0 = torch.tensor(z["x0"][0], dtype=torch.float64)
    U = torch.tensor(z["U"][0], dtype=torch.float64)
    m0, c0 = mpc.dynamics.forward_propagate_torch(H, x0, U)
    assert not m0[1].requires_grad
    m1, c1 = mpc.dynamics.forward_propagate_torch(H, x0, U.clone().requires_grad_(True))
    # the two launch shapes (first-step variant / full Jacobians) agree to rounding
    np.testing.assert_allclose(torch.stack(m1).detach().cpu().numpy(), torch.stack(m0).cpu().numpy(), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(torch.stack(c1).detach().cpu().numpy(), torch.stack(c0).cpu().numpy(), rtol=1e-9, atol=1e-15)
    with torch.no_grad():
        m2, _ = mpc.dynamics.forward_propagate_torch(H, x0, U.clone().requires_grad_(True))
    assert not m2[1].requires_grad
    c = mpc.cost_torch(m0, U, c0, mpc.x_ref, mpc.u_ref)
    assert not c.requires_grad
    np.testing.assert_allclose(c.item(), z["costs"][list(z["gammas"]).index(-1.0), 0], rtol=1e-6)


def test_cost_torch_gradients_general_covariance(G):
    """d cost / d (x, Sigma, u) of cost_torch for full NON-symmetric Sigma, R_delta on: against torch autograd of the
    reference's expression (src/mpc.py:179-198) evaluated on the CPU in this test."""
    rng = np.random.default_rng(5)
    ds, da, H = 3, 2, 4
    A = rng.standard_normal((ds, ds))
    Q = A @ A.T + ds * np.eye(ds)
    R = np.diag(rng.uniform(0.1, 1.0, da))
    Rd = np.diag(rng.uniform(0.1, 1.0, da))
    for gamma in (-0.05, 1e-5, 0.7):
        mpc = G.RiskSensitiveMPC(gamma, H, ds, da, Q, R, R_delta=Rd)
        mpc.last_traj = rng.standard_normal(H * da)
        x_ref = torch.tensor(rng.standard_normal(ds))
        u_ref = torch.tensor(rng.standard_normal(da))
        xs = rng.standard_normal((H + 1, ds))
        ss = 0.05 * rng.standard_normal((H + 1, ds, ds)) + 0.2 * np.eye(ds)
        us = rng.standard_normal((H, da))
        dev = mpc.device
        x = [torch.tensor(v, device=dev, requires_grad=True) for v in xs]
        s = [torch.tensor(v, device=dev, requires_grad=True) for v in ss]
        u = torch.tensor(us, device=dev, requires_grad=True)
        c = mpc.cost_torch(x, u, s, x_ref, u_ref)
        c.backward()
        # reference expression, CPU torch autograd
        xc = torch.tensor(xs, requires_grad=True)
        sc = torch.tensor(ss, requires_grad=True)
        uc = torch.tensor(us, requires_grad=True)
        Qt, Rt, Rdt = torch.tensor(Q), torch.tensor(R), torch.tensor(Rd)
        Qi, I = torch.linalg.inv(Qt), torch.eye(ds, dtype=torch.float64)
        ref = 0.0
        for i in range(H + 1):
            e = xc[i] - x_ref
            ref = ref + (1 / gamma) * torch.log(torch.det(I + gamma * Qt @ sc[i])) + e @ torch.linalg.inv(Qi + gamma * sc[i]) @ e
        for j in range(H):
            d = uc[j] - u_ref
            ref = ref + d @ Rt @ d
        du = torch.diff(torch.cat((torch.tensor(mpc.last_traj[0:da])[None, :], uc), dim=0), dim=0)
        for j in range(H):
            ref = ref + du[j] @ Rdt @ du[j]
        ref.backward()
        np.testing.assert_allclose(c.item(), ref.item(), rtol=1e-10)
        np.testing.assert_allclose(torch.stack([v.grad for v in x]).cpu().numpy(), xc.grad.numpy(), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(torch.stack([v.grad for v in s]).cpu().numpy(), sc.grad.numpy(), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(u.grad.cpu().numpy(), uc.grad.numpy(), rtol=1e-9, atol=1e-11)


def test_batched_function_matches_rollout_gradient(G, golden):
    """RolloutFunction + CostFunction on a batch: dU equals the fused adjoint of gpmpc_rollout for every trajectory."""
    from gaussian_process_mpc_amd.autograd import CostFunction, RolloutFunction
    z = golden("g4_rollout_c2.npz")
    pack = G.GPPack(z["X"], z["Y"], z["Ky_inv"], z["lambdas"], z["sigma_f"])
    ds, da = pack.ds, pack.da
    cp = G.CostParams(-1.0, z["Q"], z["R"], R_delta=z["R_delta"] if "R_delta" in z else None,
                      x_ref=z["x_ref"] if "x_ref" in z else np.zeros(ds), u_ref=z["u_ref"] if "u_ref" in z else np.zeros(da),
                      last_u=z["last_traj"][:da] if "last_traj" in z else None)
    x0 = torch.tensor(z["x0"], device=pack.device)
    U = torch.tensor(z["U"], device=pack.device, requires_grad=True)
    means, vars_ = RolloutFunction.apply(x0, U, pack)
    cost = CostFunction.apply(means, torch.diag_embed(vars_), U, cp)
    cost.sum().backward()
    r = G.rollout(pack, z["x0"], z["U"], cp)
    np.testing.assert_allclose(cost.detach().cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(U.grad.cpu().numpy(), r["grad"].cpu().numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("N,ds,da,B", [(300, 2, 1, 48), (520, 3, 2, 20)])
def test_batched_function_on_the_mid_size_form(G, N, ds, da, B):
    """The same on shapes the plan runs as ONE launch per horizon step on 256x64 tiles (step_fused.h, Q = 0: >= 400 tile
    workgroups of a training set of at least one 256-row tile): the Jacobian-storing forward (gpmpc_rollout_jac, step 1 with
    its state derivatives) and the reverse sweep give the fused adjoint's dU, and d/dx0 matches central differences."""
    from gaussian_process_mpc_amd.autograd import CostFunction, RolloutFunction
    from gaussian_process_mpc_amd.synth import synth_problem
    from oracle import gpmpc_oracle as O
    H = 5
    pb = synth_problem(77, N, ds, da, H, B)
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    cp = G.CostParams(-1.0, pb["Q"], pb["R"])
    x0 = torch.tensor(pb["x0"], device=pack.device, requires_grad=True)
    U = torch.tensor(pb["U"], device=pack.device, requires_grad=True)
    means, vars_ = RolloutFunction.apply(x0, U, pack)
    cost = CostFunction.apply(means, torch.diag_embed(vars_), U, cp)
    cost.sum().backward()
    r = G.rollout(pack, pb["x0"], pb["U"], cp)
    np.testing.assert_allclose(cost.detach().cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-11)
    np.testing.assert_allclose(U.grad.cpu().numpy(), r["grad"].cpu().numpy(), rtol=1e-8, atol=1e-11)
    eps = 1e-4                                  # (the objective's own round-off, ~1e-10 of O(1), bounds the quotient's accuracy)
    for k in range(ds):
        d = np.zeros_like(pb["x0"]); d[:, k] = eps
        cp_ = G.rollout(pack, pb["x0"] + d, pb["U"], cp, want_grad=False)["cost"].cpu().numpy()
        cm_ = G.rollout(pack, pb["x0"] - d, pb["U"], cp, want_grad=False)["cost"].cpu().numpy()
        np.testing.assert_allclose(x0.grad[:, k].cpu().numpy(), (cp_ - cm_) / (2 * eps), rtol=2e-5, atol=5e-6)


@pytest.mark.parametrize("k", range(6))
def test_covariance_prop_torch_carries_the_graph(G, golden, k):
    """covariance_prop_torch with u, S requiring grad (src/tools/uncertainty_prop.py:402-465): value, d/du and symmetrised d/dS against
    the REFERENCE's autograd values in g2 -- with graph-attached means (the total derivative) and with the means as constants --; the
    consistent cross-term form against central differences of itself."""
    from gaussian_process_mpc_amd import uncertainty_prop as up
    z = golden("g2_adversarial.npz")
    p = f"c{k}_"
    X = torch.tensor(z[p + "X"])
    lam1, lam2 = torch.tensor(z[p + "lam1"]), torch.tensor(z[p + "lam2"])
    K1, K2 = torch.tensor(z[p + "Kinv1"]), torch.tensor(z[p + "Kinv2"])
    y1, y2 = torch.tensor(z[p + "y1"]), torch.tensor(z[p + "y2"])
    sf1, sf2 = float(z[p + "hyp"][0]), float(z[p + "hyp"][1])
    u = torch.tensor(z[p + "u"], requires_grad=True)
    S = torch.tensor(z[p + "S"], requires_grad=True)
    m1, a1 = up.mean_prop_torch(K1, lam1, u, S, X, y1, sf1)
    m2, a2 = up.mean_prop_torch(K2, lam2, u, S, X, y2, sf2)
    sym = lambda a: 0.5 * (a + a.T)  # noqa: E731
    cov = up.covariance_prop_torch(lam1, lam2, u, S, X, m1, m2, a1["beta"], a2["beta"], sf1, sf2)
    np.testing.assert_allclose(cov.item(), z[p + "cov_torch"], rtol=1e-7, atol=1e-12)
    g_u, g_S = torch.autograd.grad(cov, (u, S), retain_graph=True)
    scale = np.abs(z[p + "dcov_du"]).max()
    np.testing.assert_allclose(g_u.cpu().numpy(), z[p + "dcov_du"], rtol=1e-6, atol=1e-9 * max(1.0, scale))
    np.testing.assert_allclose(sym(g_S.cpu().numpy()), sym(z[p + "dcov_dS"]), rtol=1e-6, atol=1e-9 * max(1.0, np.abs(z[p + "dcov_dS"]).max()))
    cov0 = up.covariance_prop_torch(lam1, lam2, u, S, X, m1.detach(), m2.detach(), a1["beta"], a2["beta"], sf1, sf2)
    h_u, h_S = torch.autograd.grad(cov0, (u, S))
    np.testing.assert_allclose(h_u.cpu().numpy(), z[p + "dcov_du_means_const"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(sym(h_S.cpu().numpy()), sym(z[p + "dcov_dS_means_const"]), rtol=1e-6, atol=1e-9)
    # without a graph: a plain value, as before
    assert not up.covariance_prop_torch(lam1, lam2, u.detach(), S.detach(), X, m1.detach(), m2.detach(), a1["beta"], a2["beta"], sf1, sf2).requires_grad
    # the consistent form (bug_compatible=False): analytic gradient against central differences of the same function
    f = lambda uu, SS: up.covariance_prop_torch(lam1, lam2, uu, SS, X, m1.detach(), m2.detach(), a1["beta"], a2["beta"], sf1, sf2,  # noqa: E731
                                                bug_compatible=False)
    c1 = f(u, S)
    q_u, q_S = torch.autograd.grad(c1, (u, S))
    eps = 1e-6
    D = u.shape[0]
    for kk in range(D):
        e = torch.zeros(D, dtype=torch.float64); e[kk] = eps
        fd = (f(u.detach() + e, S.detach()).item() - f(u.detach() - e, S.detach()).item()) / (2 * eps)
        np.testing.assert_allclose(q_u[kk].item(), fd, rtol=2e-5, atol=1e-8)
    E = torch.zeros((D, D), dtype=torch.float64); E[0, D - 1] = E[D - 1, 0] = eps; E[1, 1] = eps
    fd = (f(u.detach(), S.detach() + E).item() - f(u.detach(), S.detach() - E).item()) / (2 * eps)
    qs = sym(q_S.cpu().numpy())
    np.testing.assert_allclose(qs[0, D - 1] + qs[D - 1, 0] + qs[1, 1], fd, rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("k", range(6))
def test_single_step_functions_carry_the_graph(G, golden, k):
    """mean_prop_torch / variance_prop_torch with u, S requiring grad: d/du and (diagonal / symmetrised) d/dS against the
    reference's autograd values in g2; the pack behind the functional interface is built once per argument set."""
    from gaussian_process_mpc_amd import uncertainty_prop as up
    z = golden("g2_adversarial.npz")
    p = f"c{k}_"
    X = torch.tensor(z[p + "X"])
    lam = torch.tensor(z[p + "lam1"])
    Kinv = torch.tensor(z[p + "Kinv1"])
    y = torch.tensor(z[p + "y1"])
    u = torch.tensor(z[p + "u"], requires_grad=True)
    S = torch.tensor(z[p + "S"], requires_grad=True)
    sf = float(z[p + "hyp"][0])
    mean, aux = up.mean_prop_torch(Kinv, lam, u, S, X, y, sf)
    n1 = len(up._PACKS)
    mean2, _ = up.mean_prop_torch(Kinv, lam, u, S, X, y, sf)
    assert len(up._PACKS) == n1 and mean2.item() == mean.item()              # same objects: the pack is reused
    var = up.variance_prop_torch(Kinv, lam, u, S, X, mean, aux["beta"], sf)
    np.testing.assert_allclose(mean.item(), z[p + "mu"][0], rtol=1e-9)
    np.testing.assert_allclose(var.item(), z[p + "var"][0], rtol=1e-6)
    gm_u, gm_S = torch.autograd.grad(mean, (u, S), retain_graph=True)
    gv_u, gv_S = torch.autograd.grad(var, (u, S))
    sym = lambda a: 0.5 * (a + a.T)  # noqa: E731
    np.testing.assert_allclose(gm_u.numpy(), z[p + "dm_du"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(gv_u.numpy(), z[p + "dv_du"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sym(gm_S.numpy()), sym(z[p + "dm_dS"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(sym(gv_S.numpy()), sym(z[p + "dv_dS"]), rtol=1e-5, atol=1e-7)
    # a caller-supplied mean that differs from the device's own enters the variance as in the reference (:399)
    var_shift = up.variance_prop_torch(Kinv, lam, u.detach(), S.detach(), X, mean.detach() + 0.25, aux["beta"], sf)
    m = mean.item()
    np.testing.assert_allclose(var_shift.item(), var.item() + m * m - (m + 0.25) ** 2, rtol=1e-9, atol=1e-12)
