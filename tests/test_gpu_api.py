"""GPU tests of the host-side mirror of the reference's Python surface (GaussianProcessRegression,
Dynamics, RiskSensitiveMPC, *_prop_torch): same calls as the reference's own tests, checked against
the golden vectors the reference produced."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gaussian_process_mpc_amd as g
    g.require_gpu()
    return g


def test_gpr_matrices_and_predict(G, golden):
    z = golden("g6_gp.npz")
    gp = G.GaussianProcessRegression(3)
    gp.set_lambdas(z["lam"])
    gp.set_sigma_f(1.4)            # Python floats: float32 log, as in the fixture
    gp.set_sigma_n(0.2)
    assert gp.get_sigma_f() != 1.4 and abs(gp.get_sigma_f() - 1.4) < 1e-7
    gp.append_train_data(z["X"][:40], z["y"][:40])
    gp.append_train_data(z["X"][40:63], z["y"][40:63])
    gp.append_train_data(z["X"][63], float(z["y"][63]))          # single observation path
    assert gp.num_train == 64 and gp.X_train.shape == (64, 3) and gp.y_train.shape == (64, 1)
    np.testing.assert_allclose(gp.Kf.cpu().numpy(), z["Kf"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(gp.Ky.cpu().numpy(), z["Ky"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(gp.Ky_inv.cpu().numpy(), z["Ky_inv"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(gp.compute_pred_train_covariance(z["Xp"]).cpu().numpy(), z["Ks"], rtol=1e-12)
    np.testing.assert_allclose(gp.compute_pred_train_covariance(z["Xp"][0]).cpu().numpy(), z["Ks_single"], rtol=1e-12)
    f, none = gp.predict_latent_vars(z["Xp"])
    assert none is None and f.shape == (7, 1)
    np.testing.assert_allclose(f, z["f"], rtol=1e-9)
    f, cov = gp.predict_latent_vars(z["Xp"], covar=True)
    np.testing.assert_allclose(cov, z["cov_f"], rtol=1e-7, atol=1e-9)
    _, covy = gp.predict_latent_vars(z["Xp"], covar=True, targets=True)
    np.testing.assert_allclose(covy, z["cov_y"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(covy - cov, gp.get_sigma_n() ** 2 * np.eye(7), atol=1e-12)   # test_gpr.py identity


def test_gpr_closed_form_kernel(G):
    """Closed-form check in the style of the reference's test_gpr.py:563-609: two points, known kernel."""
    gp = G.GaussianProcessRegression(2)
    gp.set_lambdas(np.array([2.0, 0.5]))
    gp.set_sigma_n(np.array(0.3))
    gp.set_sigma_f(np.array(1.5))
    X = np.array([[0.0, 0.0], [1.0, 2.0]])
    gp.append_train_data(X, np.array([1.0, -1.0]))
    k01 = 1.5 ** 2 * np.exp(-0.5 * (1.0 / 2.0 + 4.0 / 0.5))
    Ky = np.array([[2.25, k01], [k01, 2.25]]) + float(np.float32(0.3 ** 2)) * np.eye(2)
    np.testing.assert_allclose(gp.Ky.cpu().numpy(), Ky, rtol=1e-14)
    np.testing.assert_allclose(gp.Ky_inv.cpu().numpy(), np.linalg.inv(Ky), rtol=1e-12)
    # se_kernel (src/gpr.py:124-135): two points -> 0-dim tensor, squeezing (1, D) inputs like the reference
    k = gp.se_kernel(torch.tensor(X[0:1]), torch.tensor(X[1]))
    assert k.shape == () and k.is_cuda
    np.testing.assert_allclose(k.item(), k01, rtol=1e-14)
    np.testing.assert_allclose(gp.se_kernel(torch.tensor(X[1]), torch.tensor(X[1])).item(), 2.25, rtol=1e-15)
    with_nom = G.GaussianProcessRegression(2, nominal_model=lambda x: x[:, 0:1] * 2.0)
    with_nom.set_lambdas(np.array([2.0, 0.5]))
    with_nom.append_train_data(X, np.array([1.0, -1.0]))
    f, _ = with_nom.predict_latent_vars(X)
    Kf = with_nom.Kf.cpu().numpy()
    resid = np.array([1.0, -1.0]) - 2.0 * X[:, 0]
    np.testing.assert_allclose(f[:, 0], Kf @ with_nom.Ky_inv.cpu().numpy() @ resid + 2.0 * X[:, 0], rtol=1e-12)


def test_gpr_cholesky_inverse_option(G, golden):
    """inverse="cholesky" (potrf + potri, SURVEY 8 f1 as sketched) is an opt-in: same matrices, a symmetric inverse that
    agrees with the reference's LU inverse to round-off x condition number, and the default stays LU."""
    z = golden("g6_gp.npz")
    gps = []
    for mode in ("lu", "cholesky"):
        gp = G.GaussianProcessRegression(3)
        assert gp.inverse == "lu"
        gp.inverse = mode
        gp.set_lambdas(z["lam"])
        gp.set_sigma_f(1.4)
        gp.set_sigma_n(0.2)
        gp.append_train_data(z["X"], z["y"])
        gps.append(gp)
    lu, ch = (g.Ky_inv.cpu().numpy() for g in gps)
    np.testing.assert_allclose(lu, z["Ky_inv"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(ch, z["Ky_inv"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(ch, ch.T, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(ch @ z["Ky"], np.eye(64), atol=1e-10)
    f_lu, _ = gps[0].predict_latent_vars(z["Xp"])
    f_ch, _ = gps[1].predict_latent_vars(z["Xp"])
    np.testing.assert_allclose(f_ch, f_lu, rtol=1e-9)
    bad = G.GaussianProcessRegression(3)
    bad.inverse = "qr"
    with pytest.raises(ValueError):
        bad.append_train_data(z["X"][:4], z["y"][:4])


@pytest.mark.parametrize("tag", ["a", "c"])
def test_prop_torch_mirrors(G, golden, tag):
    """mean_prop_torch / variance_prop_torch / covariance_prop_torch called like the reference's tests
    (test_uncertainty_prop.py:182-385) with the reference's tolerances (and tighter)."""
    z = golden("g1_single_step.npz")
    T = lambda a: torch.tensor(a)      # noqa: E731
    sf1, sf2 = z[f"{tag}_sf"]
    X, y, u, S = T(z[f"{tag}_X"]), T(z[f"{tag}_y"]), T(z["u"]), T(z["S"])
    m1, d1 = G.mean_prop_torch(T(z[f"{tag}_Kinv1"]), T(z["lam1"]), u, S, X, y, sf1)
    m2, d2 = G.mean_prop_torch(T(z[f"{tag}_Kinv2"]), T(z["lam2"]), u, S, X, y, sf2)
    assert abs(m1.item() - z[f"{tag}_mu"][0]) < 1e-7 * abs(z[f"{tag}_mu"][0])
    assert np.linalg.norm(d1["beta"].cpu().numpy() - z[f"{tag}_beta1"]) < 1e-5
    assert np.linalg.norm(d1["l"].cpu().numpy() - z[f"{tag}_l1"]) < 1e-5
    np.testing.assert_allclose(d1["l"].cpu().numpy(), z[f"{tag}_l1"], rtol=1e-10, atol=1e-300)
    v1 = G.variance_prop_torch(T(z[f"{tag}_Kinv1"]), T(z["lam1"]), u, S, X, m1, d1["beta"], sf1)
    assert abs(v1.item() - z[f"{tag}_var"][0]) < 1e-5 * abs(z[f"{tag}_var"][0])
    cv = G.covariance_prop_torch(T(z["lam1"]), T(z["lam2"]), u, S, X, m1, m2, d1["beta"], d2["beta"], sf1, sf2)
    assert abs(cv.item() - z[f"{tag}_cov"]) < 1e-5 * abs(z[f"{tag}_cov"])


def _mpc_from(G, z, gamma):
    N, ds, da, H = (int(v) for v in z["dims"])
    mpc = G.RiskSensitiveMPC(gamma, H, ds, da, z["Q"], z["R"], z["R_delta"] if "R_delta" in z else None)
    for a in range(ds):
        g = mpc.dynamics.gpr_err[a]
        g.set_lambdas(z["lambdas"][a])
        g.set_sigma_n(float(z["sigma_n"][a]))
        g.set_sigma_f(1.0)
    mpc.dynamics.append_train_data(z["X"][:, :ds], z["X"][:, ds:], z["Y"])
    if "x_ref" in z:
        mpc.set_xref(z["x_ref"])
    if "u_ref" in z:
        mpc.set_uref(z["u_ref"])
    if "last_traj" in z:
        mpc.last_traj = z["last_traj"].copy()
    return mpc


@pytest.mark.parametrize("name", ["g3_rollout_c1.npz", "g4_rollout_c2.npz"])
def test_dynamics_and_mpc_callbacks(G, golden, name):
    """End to end through the mirrored classes, Ky_inv rebuilt on the device (explicit inverse as
    src/gpr.py:171): forward_propagate_torch, objective, gradient vs the reference's outputs."""
    z = golden(name)
    N, ds, da, H = (int(v) for v in z["dims"])
    mpc = _mpc_from(G, z, float(z["gammas"][0]))
    for a in range(ds):
        scale = np.abs(z["Ky_inv"][a]).max()
        np.testing.assert_allclose(mpc.dynamics.gpr_err[a].Ky_inv.cpu().numpy(), z["Ky_inv"][a], rtol=0, atol=1e-6 * scale)
    for b in range(z["x0"].shape[0]):
        x0 = torch.tensor(z["x0"][b]).type(torch.float64)
        means, covs = mpc.dynamics.forward_propagate_torch(H, x0, torch.tensor(z["U"][b]))
        assert len(means) == H + 1 and len(covs) == H + 1 and covs[1].shape == (ds, ds)
        np.testing.assert_allclose(torch.stack(means).cpu().numpy(), z["means"][b], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(torch.stack([torch.diag(c) for c in covs]).cpu().numpy(), z["vars"][b], rtol=1e-4)
        assert float((covs[3] - torch.diag(torch.diag(covs[3]))).abs().max()) == 0.0
        mpc.curr_state = x0.to(mpc.device)
        for gi, gamma in enumerate(z["gammas"]):
            mpc.gamma = float(gamma)
            mpc._cache_key = None
            x = z["U"][b].reshape(-1).copy()
            c = mpc.objective(x)
            g = mpc.gradient(x)
            assert isinstance(c, float) and g.shape == (H, da)
            np.testing.assert_allclose(c, z["costs"][gi, b], rtol=1e-6)
            np.testing.assert_allclose(g, z["grads"][gi, b], rtol=1e-4, atol=1e-7)
    assert mpc.constraints(x) == 0 and np.all(mpc.jacobian(x) == 0) and mpc.jacobian(x).shape == x.shape


def test_state_host_copy_survives_a_full_covariance_detour(G, golden):
    """The host copy of curr_state behind the B = 1 callback entry is keyed on its own source tensor: toggling
    full_covariance after curr_state was REPLACED by a new tensor of the same version must not upload the old state; and
    the value cache behind objective / gradient follows the full_covariance switch (the two propagations differ by ~5e-5
    in this cost: a full-covariance value must not be served to a diagonal-covariance call)."""
    z = golden("g3_rollout_c1.npz")
    mpc = _mpc_from(G, z, -1.0)
    x = z["U"][0].reshape(-1).copy()
    mpc.curr_state = torch.tensor(z["x0"][0]).to(mpc.device)
    c_a = mpc.objective(x)
    mpc.full_covariance = True
    mpc.curr_state = torch.tensor(z["x0"][1]).to(mpc.device)            # new tensor, _version 0 like the old one
    c_full = mpc.objective(x)                                            # full-covariance branch refreshes _cache_held only
    mpc.full_covariance = False
    c_b = mpc.objective(x)
    assert c_b != c_full
    fresh = _mpc_from(G, z, -1.0)
    fresh.curr_state = torch.tensor(z["x0"][1]).to(fresh.device)
    assert c_b == pytest.approx(fresh.objective(x), rel=1e-12) and c_b != c_a


def test_objective_cache_semantics(G, golden):
    z = golden("g3_rollout_c1.npz")
    mpc = _mpc_from(G, z, -1.0)
    mpc.curr_state = torch.tensor(z["x0"][0]).to(mpc.device)
    x = z["U"][0].reshape(-1).copy()
    c0 = mpc.objective(x)
    x_alias = x                       # Ipopt reuses its buffer: mutate after the call
    g0 = mpc.gradient(x).copy()
    x_alias[0] += 0.25
    c1 = mpc.objective(x_alias)       # new bytes -> new evaluation (the buffer is copied, never aliased)
    assert c1 != c0
    g1 = mpc.gradient(x_alias)
    assert not np.allclose(g0, g1)
    x_alias[0] -= 0.25
    assert mpc.objective(x_alias) == c0
    # the caches behind the callbacks follow every input that the reference reads at call time
    g_keep = mpc.gradient(x_alias).copy()
    mpc.curr_state = torch.tensor(z["x0"][0] + 0.1).to(mpc.device)       # attribute replaced
    c2 = mpc.objective(x_alias)
    assert c2 != c0
    mpc.curr_state.add_(-0.1)                                              # modified in place
    assert mpc.objective(x_alias) == pytest.approx(c0, rel=1e-12)
    np.testing.assert_allclose(mpc.gradient(x_alias), g_keep, rtol=1e-10)
    mpc.set_xref(np.full(mpc.state_dim, 0.3))                               # cost parameters
    c3 = mpc.objective(x_alias)
    assert c3 != pytest.approx(c0, rel=1e-6)
    mpc.x_ref = torch.zeros(mpc.state_dim, device=mpc.device)             # direct attribute assignment, as the reference's tests do
    assert mpc.objective(x_alias) == pytest.approx(c0, rel=1e-12)
    lam_old = mpc.dynamics.gpr_err[0].get_lambdas().copy()
    mpc.dynamics.gpr_err[0].set_lambdas(lam_old * 1.5)                     # hyper-parameter change -> new pack
    for gp in mpc.dynamics.gpr_err:
        gp.build_Ky_inv_mat()
    assert mpc.objective(x_alias) != pytest.approx(c0, rel=1e-6)


def test_cost_methods_known_answers(G, golden):
    z = golden("g5_cost.npz")
    mpc = G.RiskSensitiveMPC(1, 1, 2, 2, z["a_Q"], z["a_R"])
    c_np = mpc.cost(z["a_x"], z["a_u"], z["a_sig"], z["a_xref"], z["a_uref"])
    assert abs(c_np - z["a_cost_np"]) < 1e-9
    T = lambda a: torch.tensor(a, device=mpc.device).type(torch.float64)   # noqa: E731
    c_t = mpc.cost_torch(T(z["a_x"]), T(z["a_u"]), T(z["a_sig"]), T(z["a_xref"]), T(z["a_uref"]))
    assert abs(c_np - c_t.item()) < 1e-5                                    # test_mpc.py:104
    mpc = G.RiskSensitiveMPC(1.1, 2, 2, 2, z["a_Q"], z["a_R"], z["b_Rd"])
    mpc.last_traj = [0 for _ in range(4)]
    c_t = mpc.cost_torch([T(r) for r in z["b_x"]], T(z["b_u"]), [T(s) for s in z["b_sig"]], T(z["a_xref"]), T(z["a_uref"]))
    assert abs(c_t.item() - z["b_cost_torch"]) < 1e-6                       # test_mpc.py:243
    H = 5
    mpc = G.RiskSensitiveMPC(-1, H, 1, 1, 2 * np.identity(1), np.array([[0]]), np.array([[0]]))
    c_t = mpc.cost_torch(T(z["c_x"]).reshape(H + 1, 1), torch.zeros((H, 1), device=mpc.device).type(torch.float64),
                         T(z["c_sig"]).reshape(H + 1, 1, 1), torch.zeros(1, device=mpc.device), torch.zeros(1, device=mpc.device))
    assert abs(z["c_closed"] - c_t.item()) < 1e-7                           # test_mpc.py:274


def test_get_optimal_trajectory(G, golden):
    """Shape contract of the reference's smoke test (test_mpc.py:106-139) plus: the solve lowers the cost."""
    z = golden("g3_rollout_c1.npz")
    N, ds, da, H = (int(v) for v in z["dims"])
    empty = G.RiskSensitiveMPC(-1.0, H, ds, da, z["Q"], z["R"])
    assert np.all(empty.get_optimal_trajectory(np.zeros(ds)) == 0)          # no data: zeros (mpc.py:285-289)
    mpc = _mpc_from(G, z, -1.0)
    mpc.set_lb([-1.0] * da)
    mpc.set_ub([1.0] * da)
    traj = mpc.get_optimal_trajectory(z["x0"][0])
    assert traj.shape == (H, da) and np.all(np.abs(traj) <= 1.0 + 1e-9)
    assert mpc.objective(traj.reshape(-1)) <= mpc.objective(np.zeros(H * da)) + 1e-12
    assert mpc.solver_used in ("ipopt", "scipy-lbfgsb")


def test_batch_matches_loop_and_pack_refresh(G, golden):
    z = golden("g4_rollout_c2.npz")
    mpc = _mpc_from(G, z, -1.0)
    cost, grad = mpc.objective_batch(z["U"], z["x0"])
    for b in range(2):
        mpc.curr_state = torch.tensor(z["x0"][b]).to(mpc.device)
        np.testing.assert_allclose(mpc.objective(z["U"][b].reshape(-1)), cost[b], rtol=1e-9)
        np.testing.assert_allclose(mpc.gradient(z["U"][b].reshape(-1)), grad[b], rtol=1e-6, atol=1e-10)
    # appending one observation (Simulator.run does this every step, simulator.py:55) refreshes the pack
    p0 = mpc.dynamics.pack()
    assert mpc.dynamics.pack() is p0
    mpc.dynamics.append_train_data(np.zeros(3), np.zeros(1), np.array([0.1, -0.1, 0.05]))
    p1 = mpc.dynamics.pack()
    assert p1 is not p0 and p1.N == p0.N + 1
    c2, _ = mpc.objective_batch(z["U"], z["x0"])
    assert np.all(np.isfinite(c2)) and not np.allclose(c2, cost, rtol=1e-12)


def test_closed_loop_simulator(G):
    """Simulator.run semantics (reference src/simulator.py:37-60) on the dependency-free pendulum plant:
    solve, step, append one observation (pack refresh) per iteration."""
    rng = np.random.default_rng(3)
    plant = G.PendulumPlant(init_state=(0.3, 0.0))
    S = np.stack((rng.uniform(-1, 1, 40), rng.uniform(-2, 2, 40)), axis=1)
    A = rng.uniform(-2, 2, (40, 1))
    nxt = np.array([G.PendulumPlant(init_state=s).step(a)[0] for s, a in zip(S, A)])
    mpc = G.RiskSensitiveMPC(-1.0, 5, 2, 1, 0.5 * np.eye(2), 0.01 * np.eye(1))
    for g in mpc.dynamics.gpr_err:
        g.set_lambdas(np.array([1.0, 4.0, 4.0]))
        g.set_sigma_n(np.array(1e-2))
    mpc.dynamics.append_train_data(S, A, nxt)
    mpc.set_lb([-2.0]); mpc.set_ub([2.0])
    sim = G.Simulator(mpc, plant, num_iters=3)
    hist = sim.run()
    assert len(hist) == 3 and hist[0][1].shape == (1,)
    assert mpc.dynamics.gpr_err[0].num_train == 43 and mpc.dynamics.pack().N == 43
    assert all(np.isfinite(h[2]) for h in hist) and all(abs(h[1][0]) <= 2.0 + 1e-9 for h in hist)


def test_closed_loop_simulator_multistart(G):
    """get_optimal_trajectory(n_starts = 16) (multistart.py; the solve of src/mpc.py:269-330 with K starts advanced in lock-step):
    every solver iteration is ONE batched rollout of 16 candidate plans; the plan returned costs no more than the single-start
    plan from the same state on the same seeds (start 0 IS the single start's zero start), and Simulator.run works unchanged.
    Optimiser results are unpinned (no Ipopt in the image); the bound is against this build's own single-start solve."""
    rng = np.random.default_rng(3)
    S = np.stack((rng.uniform(-1, 1, 40), rng.uniform(-2, 2, 40)), axis=1)
    A = rng.uniform(-2, 2, (40, 1))
    nxt = np.array([G.PendulumPlant(init_state=s).step(a)[0] for s, a in zip(S, A)])

    def make():
        mpc = G.RiskSensitiveMPC(-1.0, 5, 2, 1, 0.5 * np.eye(2), 0.01 * np.eye(1))
        for g in mpc.dynamics.gpr_err:
            g.set_lambdas(np.array([1.0, 4.0, 4.0]))
            g.set_sigma_n(np.array(1e-2))
        mpc.dynamics.append_train_data(S, A, nxt)
        mpc.set_lb([-2.0]); mpc.set_ub([2.0])
        return mpc
    m1, m16 = make(), make()
    for state in ((0.3, 0.0), (-0.6, 1.2), (0.9, -1.5)):
        u1 = m1.get_optimal_trajectory(np.array(state))
        c1 = m1.objective(u1.reshape(-1))
        u16 = m16.get_optimal_trajectory(np.array(state), n_starts=16)
        info = m16.last_solve_info
        c16 = m16.objective(u16.reshape(-1))
        assert m16.solver_used == "lockstep-lbfgs x16" and info["starts"] == 16 and info["alive"][0]
        assert info["evaluations"] == info["ticks"] + 1 <= m16.multistart_options["max_ticks"] + 1
        np.testing.assert_allclose(c16, info["f"].min(), rtol=1e-9)          # the reported best IS the plan returned
        assert np.all(np.abs(u16) <= 2.0 + 1e-12)
        assert c16 <= c1 + 1e-6 * max(1.0, abs(c1)), (state, c16, c1)
        # the batched evaluation the search consumed equals the solver-callback entry, start by start
        cb, gb = m16.objective_batch(info["x"].reshape(16, 5, 1), curr_state=m16.curr_state)
        for k in (0, 7, 15):
            np.testing.assert_allclose(cb[k], m16.objective(info["x"][k]), rtol=1e-9)
    m16.n_starts = 16
    hist = G.Simulator(m16, G.PendulumPlant(init_state=(0.3, 0.0)), num_iters=3).run()
    assert len(hist) == 3 and m16.dynamics.gpr_err[0].num_train == 43 and all(abs(h[1][0]) <= 2.0 + 1e-9 for h in hist)


def test_integration_md_stub_runs_verbatim(G, golden, monkeypatch):
    """INTEGRATION.md section B documents the ctypes binding a maintainer of the reference would add.  Extract that code
    block from the file and execute it AS WRITTEN (only GPMPC_LIB points it at the in-tree library), with the mirror's
    Dynamics / RiskSensitiveMPC objects standing in for the reference's (same attributes), against the reference's own
    rollout outputs (g3: cost and gradient for every gamma, both start states)."""
    import os
    from gaussian_process_mpc_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sect = text[text.index("## B."):]
    start = sect.index("```python") + len("```python")
    code = sect[start:sect.index("```", start)]
    assert "def build_pack" in code and "def objective_and_gradient" in code
    monkeypatch.setenv("GPMPC_LIB", _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    z = golden("g3_rollout_c1.npz")
    mpc = _mpc_from(G, z, float(z["gammas"][0]))
    pack = ns["build_pack"](mpc.dynamics)
    for b in range(z["x0"].shape[0]):
        mpc.curr_state = torch.tensor(z["x0"][b], device=mpc.device).type(torch.float64)
        for gi, gamma in enumerate(z["gammas"]):
            mpc.gamma = float(gamma)
            c, g = ns["objective_and_gradient"](mpc, pack, z["U"][b].reshape(-1))
            np.testing.assert_allclose(c, z["costs"][gi, b], rtol=1e-6)
            np.testing.assert_allclose(g, z["grads"][gi, b], rtol=1e-4, atol=1e-7)


def test_readme_experiment_risk_averse_follows_the_data(G, golden):
    """The experiment behind the reference's README figures (src/experiments/pretrain_uncertainty.py:82-121; its own data
    files, hyper-parameters lambda = 0.5 / sigma_f = 1 / sigma_n = 1e-5, Q = 2 I, R = 0, H = 6, start (4, -4)), run through
    the mirror classes and the closed loop of Simulator.run (src/simulator.py:37-60) on the true dynamics s' = s + a.
    Behavioural properties, not shapes (README.md: risk-averse plans stay close to the training data, risk-neutral ones
    cut across the region without data): the optimiser lowers the cost; the first risk-averse input climbs the corridor
    at x = 4 while the risk-neutral one moves diagonally; predicted and realised paths of gamma = -1 stay closer to the
    training states than those of gamma = 1e-5; the loop makes progress towards the target."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from readme_uncertainty_experiment import build_mpc, distance_to_data
    data = golden("g9_closed_loop.npz")
    np.testing.assert_allclose(data["exp_next_states"], data["exp_states"] + data["exp_actions"], rtol=1e-14)      # s' = s + a

    class AdditivePlant(object):
        def __init__(self):
            self.state = np.array([4.0, -4.0])

        def reset(self):
            self.state = np.array([4.0, -4.0])
            return self.state.copy(), {}

        def step(self, a):
            self.state = self.state + np.asarray(a, dtype=float)
            return self.state.copy(), -float(self.state @ self.state), False, False, {}

    out = {}
    for gamma in (-1.0, 1e-5):
        mpc = build_mpc(gamma, data)
        s0 = np.array([4.0, -4.0])
        plan = mpc.get_optimal_trajectory(s0)
        assert plan.shape == (6, 2) and np.all(np.abs(plan) <= 1 + 1e-9)
        planned = mpc.objective(plan.reshape(-1))
        assert np.isfinite(planned) and planned < 0.6 * mpc.objective(np.zeros(12))                # the optimiser lowers the cost
        r = mpc.evaluate_batch(plan[None], s0)
        means = r["means"][0].cpu().numpy()
        hist = G.Simulator(mpc, AdditivePlant(), num_iters=4).run()
        path = np.array([h[0] for h in hist] + [hist[-1][0] + hist[-1][1]])
        assert mpc.dynamics.gpr_err[0].num_train == 404                                            # one observation per loop step
        out[gamma] = dict(plan=plan, d_pred=distance_to_data(means[1:], data["exp_states"]).mean(),
                          d_loop=distance_to_data(path, data["exp_states"]).mean(), path=path)
        # progress towards the target: 4 steps of |a| <= 1 along the corridor alone (risk-averse) shorten |s| by 1.66; the
        # variances carry percent-level round-off at sigma_n = 1e-5, so the exact path depends on the summation order
        assert np.linalg.norm(path[-1]) < np.linalg.norm(path[0]) - 1.2
    averse, neutral = out[-1.0], out[1e-5]
    assert averse["plan"][0, 1] > 0.8 and abs(averse["plan"][0, 0]) < 0.5        # up the corridor at x = 4 first
    assert neutral["plan"][0, 0] < -0.8 and neutral["plan"][0, 1] > 0.8          # diagonal, across the region without data
    assert averse["d_pred"] < 0.8 * neutral["d_pred"]
    assert averse["d_loop"] < 0.8 * neutral["d_loop"]


def test_incremental_append_matches_rebuild(G, golden):
    """gpmpc_kinv_append (Schur-complement append, O(N^2)) against the reference-style full rebuild (src/gpr.py:171)."""
    z = golden("g6_gp.npz")
    full = G.GaussianProcessRegression(3)
    inc = G.GaussianProcessRegression(3)
    for gp in (full, inc):
        gp.set_lambdas(z["lam"]); gp.set_sigma_f(1.4); gp.set_sigma_n(0.2)
        gp.append_train_data(z["X"][:50], z["y"][:50])
    for i in range(50, 64):
        full.append_train_data(z["X"][i], float(z["y"][i]))
        inc.append_train_data(z["X"][i], float(z["y"][i]), incremental=True)
    assert inc.num_train == 64 and inc.Ky_inv.shape == (64, 64)
    np.testing.assert_allclose(inc.Ky.cpu().numpy(), z["Ky"], rtol=1e-12, atol=1e-14)
    scale = np.abs(z["Ky_inv"]).max()
    np.testing.assert_allclose(inc.Ky_inv.cpu().numpy(), full.Ky_inv.cpu().numpy(), rtol=0, atol=1e-10 * scale)
    np.testing.assert_allclose(inc.Ky_inv.cpu().numpy(), z["Ky_inv"], rtol=0, atol=1e-9 * scale)
    f_inc, _ = inc.predict_latent_vars(z["Xp"])
    np.testing.assert_allclose(f_inc, z["f"], rtol=1e-8)


def test_incremental_append_drift_and_hyper_change(G):
    """260 single-observation appends through the O(N^2) path: the periodic full rebuild (every `rebuild_every` appends)
    bounds the accumulated round-off -- the final inverse agrees with a from-scratch build like a fresh inverse does --
    and a hyper-parameter edit between appends (the setters do not rebuild, src/gpr.py:53) makes the next append rebuild
    everything, as the reference's append always does (src/gpr.py:122), instead of mixing old and new hypers."""
    rng = np.random.default_rng(12)
    X = rng.uniform(-2, 2, (300, 3))
    y = np.sin(X).sum(axis=1) + 0.05 * rng.normal(size=300)
    inc = G.GaussianProcessRegression(3)
    inc.set_lambdas(np.array([1.5, 2.0, 0.8])); inc.set_sigma_f(np.array(1.2)); inc.set_sigma_n(np.array(1e-2))
    inc.append_train_data(X[:40], y[:40])
    rebuilds = 0
    for i in range(40, 300):
        v = inc._appends_since_rebuild
        inc.append_train_data(X[i], float(y[i]), incremental=True)
        rebuilds += int(inc._appends_since_rebuild < v)
    assert inc.num_train == 300 and rebuilds == 260 // (inc.rebuild_every + 1)
    ref = G.GaussianProcessRegression(3)
    ref.set_lambdas(np.array([1.5, 2.0, 0.8])); ref.set_sigma_f(np.array(1.2)); ref.set_sigma_n(np.array(1e-2))
    ref.append_train_data(X, y)
    np.testing.assert_allclose(inc.Ky.cpu().numpy(), ref.Ky.cpu().numpy(), rtol=1e-13, atol=1e-15)
    eye = torch.eye(300, dtype=torch.float64, device=inc.Ky.device)
    res_inc = float((inc.Ky @ inc.Ky_inv - eye).abs().max())
    res_ref = float((ref.Ky @ ref.Ky_inv - eye).abs().max())
    assert res_inc < 20 * max(res_ref, 1e-12), (res_inc, res_ref)          # cond(Ky) ~ 1e5: both residuals ~1e-11
    f_inc, _ = inc.predict_latent_vars(X[:20])
    f_ref, _ = ref.predict_latent_vars(X[:20])
    np.testing.assert_allclose(f_inc, f_ref, rtol=1e-7, atol=1e-9)
    # hyper-parameter change, then one incremental append: everything is rebuilt with the new values
    inc.set_lambdas(np.array([3.0, 3.0, 3.0]))
    inc.append_train_data(np.array([0.1, 0.2, 0.3]), 0.5, incremental=True)
    new = G.GaussianProcessRegression(3)
    new.set_lambdas(np.array([3.0, 3.0, 3.0])); new.set_sigma_f(np.array(1.2)); new.set_sigma_n(np.array(1e-2))
    new.append_train_data(np.vstack((X, [[0.1, 0.2, 0.3]])), np.append(y, 0.5))
    assert inc._appends_since_rebuild == 0
    assert torch.equal(inc.Ky, new.Ky) and torch.equal(inc.Ky_inv, new.Ky_inv)


def test_gps_that_shared_matrices_and_are_then_fed_separately(G):
    """Two GPs with identical hyper-parameters fed through Dynamics share ONE set of matrices (update_many / _adopt); the leader's
    incremental append writes into two REUSED buffer sets.  Feeding the leader ON ITS OWN twice afterwards (public
    append_train_data) must not change what the follower holds: its n-point inverse stays the n-point inverse
    (src/gpr.py:159-171 assigns fresh tensors on every update, so in the reference nothing is ever modified under a holder)."""
    rng = np.random.default_rng(5)
    dyn = G.Dynamics(2, 1)
    for gp in dyn.gpr_err:
        gp.set_lambdas(np.array([1.0, 2.0, 3.0])); gp.set_sigma_n(np.array(1e-2))
    S, A = rng.uniform(-1, 1, (30, 2)), rng.uniform(-1, 1, (30, 1))
    dyn.append_train_data(S, A, S + 0.1 * np.tanh(S))
    for _ in range(3):                                       # incremental: the GPs now hold views of the leader's ping-pong buffers
        s, a = rng.uniform(-1, 1, 2), rng.uniform(-1, 1, 1)
        dyn.append_train_data(s, a, s + 0.1 * np.tanh(s), incremental=True)
    lead, foll = dyn.gpr_err
    assert foll.Ky_inv.untyped_storage().data_ptr() == lead.Ky_inv.untyped_storage().data_ptr()      # shared
    held = {k: getattr(foll, k).clone() for k in ("Kf", "Ky", "Ky_inv")}
    for _ in range(3):                                       # the leader alone: every buffer set gets rewritten
        x = rng.uniform(-1, 1, 3)
        lead.append_train_data(x, float(np.sin(x).sum()), incremental=True)
    assert lead.num_train == 36 and foll.num_train == 33 and foll.Ky_inv.shape == (33, 33)
    for k, v in held.items():
        assert torch.equal(getattr(foll, k), v), k
    f, _ = foll.predict_latent_vars(np.concatenate((S[:5], A[:5]), axis=1))
    assert np.all(np.isfinite(f))


# ---- hyper-parameter training: src/gpr.py:173-251, 334-370 -------------------------------------------
def test_marginal_likelihood_value(G, golden):
    d = golden("g8_hyper.npz")
    gp = G.GaussianProcessRegression(3)
    gp.set_lambdas(d["ml_lam"]); gp.set_sigma_f(1.4); gp.set_sigma_n(0.2)
    gp.append_train_data(d["ml_X"], d["ml_y"])
    ml = gp.compute_marginal_likelihood()
    assert tuple(ml.shape) == (1, 1) and ml.is_cuda
    assert ml.item() == pytest.approx(float(d["ml_value"]), rel=1e-10)


def test_marginal_likelihood_gradient_against_oracle_autograd(G):
    """one-pass trace kernel (gpmpc_ml_grad) vs autograd through inv / det (the reference's backward)."""
    from oracle import gpmpc_oracle as O
    rng = np.random.default_rng(31)
    for N, D, nominal in [(37, 2, False), (130, 5, False), (64, 1, True), (200, 8, False)]:
        X = rng.uniform(-2, 2, (N, D)); y = np.sin(X).sum(axis=1) + 0.1 * rng.normal(size=N)
        ll, lf, ln = rng.uniform(-0.5, 1.0, D), 0.3, -1.2
        nom = (lambda x: 0.5 * x[:, 0:1]) if nominal else None
        gp = G.GaussianProcessRegression(D, nominal_model=nom)
        gp.set_lambdas(np.exp(ll)); gp.set_sigma_f(np.array(np.exp(lf))); gp.set_sigma_n(np.array(np.exp(ln)))   # float64 arrays: exact logs
        gp.append_train_data(X, y)
        tr = O.HyperTrainer(X, y, D, nominal=0.5 * X[:, 0] if nominal else None,
                            log_lambdas=gp.log_lambdas.detach().cpu().numpy(), log_sigma_f=gp.log_sigma_f.item(),
                            log_sigma_n=gp.log_sigma_n.item())
        ref = tr.step()                                   # likelihood + gradients at the initial point
        g = gp.marginal_likelihood_gradient()
        assert gp.compute_marginal_likelihood().item() == pytest.approx(ref["ml"], rel=1e-9)
        np.testing.assert_allclose(g["log_lambda"], ref["grad"]["log_lambdas"], rtol=1e-7, atol=1e-8)
        assert g["log_sigma_f"] == pytest.approx(ref["grad"]["log_sigma_f"], rel=1e-7, abs=1e-8)
        assert g["log_sigma_n"] == pytest.approx(ref["grad"]["log_sigma_n"], rel=2e-6, abs=1e-6)   # float32 noise path in the reference
        # dense API-compatibility tensors: 1/2 tr(B dK) reproduces the kernel's numbers
        dK = gp.kernel_matrix_gradient()
        a = gp.beta().reshape(-1, 1)
        Bm = a @ a.T - gp.Ky_inv
        dense = np.array([0.5 * (Bm * dK["lambda"][:, :, k].T).sum().item() for k in range(D)]) * gp.get_lambdas()
        np.testing.assert_allclose(dense, g["log_lambda"], rtol=1e-9, atol=1e-10)
        mg = gp.marginal_likelihood_grad(dK)
        assert mg["sigma_f"].item() * gp.get_sigma_f() == pytest.approx(g["log_sigma_f"], rel=1e-12)


@pytest.mark.parametrize("tag,x_dim,nominal", [("t1", 1, False), ("t2", 1, True), ("t3", 3, False)])
def test_update_hyperparams_matches_reference_trajectory(G, golden, tag, x_dim, nominal):
    """Adam iterates of update_hyperparams against the reference's own (src/gpr.py:334-370; fixture g8)."""
    d = golden("g8_hyper.npz")
    gp = G.GaussianProcessRegression(x_dim, nominal_model=(lambda x: x) if nominal else None)
    gp.append_train_data(d[tag + "_X"], d[tag + "_y"])
    K = len(d[tag + "_ml"])
    hist = gp.update_hyperparams(num_iters=K)
    assert len(hist) == K
    # The reference's sigma_n gradient passes through a float32 product (sigma_n**2 * float32 eye, src/gpr.py:170), i.e.
    # carries ~1e-7 relative noise that Adam feeds back into every iterate; the kernel's gradient is float64.  The
    # trajectories therefore agree to ~1e-7 in the log-parameters and, with |d ml / d theta| ~ 1e2, ~1e-6 in ml.
    for k, h in enumerate(hist):
        assert h["ml"] == pytest.approx(d[tag + "_ml"][k], rel=2e-6)
        np.testing.assert_allclose(h["grad"]["log_lambdas"], d[tag + "_g_log_lambdas"][k], rtol=2e-5, atol=1e-5)
        assert h["grad"]["log_sigma_n"] == pytest.approx(d[tag + "_g_log_sigma_n"][k], rel=2e-5, abs=1e-5)
        np.testing.assert_allclose(h["log_lambdas"], d[tag + "_log_lambdas"][k], rtol=1e-5, atol=1e-6)
        assert h["log_sigma_f"] == pytest.approx(d[tag + "_log_sigma_f"][k], rel=1e-5, abs=1e-6)
        assert h["log_sigma_n"] == pytest.approx(d[tag + "_log_sigma_n"][k], rel=1e-5, abs=1e-6)
    # the matrices were rebuilt at the final hyper-parameters, and the likelihood went up
    assert hist[-1]["ml"] > hist[0]["ml"]
    np.testing.assert_allclose(gp.get_lambdas(), np.exp(d[tag + "_log_lambdas"][-1]), rtol=1e-5)
    # a second call continues with the same Adam state (k calls of one iteration == k iterations)
    gp2 = G.GaussianProcessRegression(x_dim, nominal_model=(lambda x: x) if nominal else None)
    gp2.append_train_data(d[tag + "_X"], d[tag + "_y"])
    for _ in range(3):
        gp2.update_hyperparams(num_iters=1)
    np.testing.assert_allclose(gp2.log_lambdas.detach().cpu().numpy(), d[tag + "_log_lambdas"][2], rtol=1e-5, atol=1e-6)


def test_ml_grad_abi_errors(G):
    import ctypes
    from gaussian_process_mpc_amd._lib import lib, ptr, host_doubles
    L = lib()
    assert L.gpmpc_ml_grad_workspace_bytes(0, 3) == 0 and L.gpmpc_ml_grad_workspace_bytes(10, 9) == 0
    n, D = 16, 2
    X = torch.zeros((n, D), dtype=torch.float64, device="cuda"); K = torch.eye(n, dtype=torch.float64, device="cuda")
    a = torch.zeros(n, dtype=torch.float64, device="cuda"); out = torch.zeros(D + 3, dtype=torch.float64, device="cuda")
    nb = L.gpmpc_ml_grad_workspace_bytes(n, D)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _, lp = host_doubles(np.array([1.0, 2.0]))
    _, bad = host_doubles(np.array([1.0, -2.0]))
    wsp = ctypes.c_void_p(ws.data_ptr())
    assert L.gpmpc_ml_grad(n, D, ptr(X), ptr(K), ptr(a), ptr(a), lp, 1.0, 0.01, ptr(out), wsp, nb - 8, None) != 0   # workspace
    assert L.gpmpc_ml_grad(n, D, ptr(X), ptr(K), ptr(a), ptr(a), bad, 1.0, 0.01, ptr(out), wsp, nb, None) != 0     # lambda <= 0
    assert L.gpmpc_ml_grad(n, 9, ptr(X), ptr(K), ptr(a), ptr(a), lp, 1.0, 0.01, ptr(out), wsp, nb, None) != 0      # D > GPMPC_MAX_D
    assert L.gpmpc_ml_grad(n, D, ptr(X), ptr(K), ptr(a), ptr(a), lp, 1.0, 0.01, ptr(out), wsp, nb, None) == 0
    torch.cuda.synchronize()
    # alpha = 0, Ky_inv = I, X = 0: B = -I, Kf = sigma_f^2 -> d/dlog sigma_f = -n, d/dlog sigma_n = -n * noise, lambda terms 0
    np.testing.assert_allclose(out.cpu().numpy(), [0.0, 0.0, -n, -n * 0.01, 0.0], atol=1e-12)


@pytest.mark.parametrize("si", [0, 1])
def test_g10_readme_regime_through_the_library_and_the_classes(G, golden, si):
    """The reference's own regime (README experiment data, lambda = 0.5 for every GP, H = 6; sigma_n = 1e-3 / the
    experiments' 1e-5), reference-produced means / variances / cost / gradient of four candidate plans:
    (i) gpmpc_rollout on the fixture's Ky_inv -- the shared-lambda pack is detected; per quantity the HIP path meets
        means 1e-7, variances 1e-6, cost 1e-7, gradient 1e-5 (north star: 1e-5 / 1e-4) at BOTH noise levels: this regime is
        benign (variances 0.02-0.06 against sigma_f^2 = 1), unlike the dense synthetic sets of the accuracy sweep;
    (ii) the mirror classes with the experiment's setter calls (src/experiments/pretrain_uncertainty.py:100-105), Ky_inv
        rebuilt on the device: the float32 log of the setters is reproduced (effective lambda != 0.5) and the callbacks agree
        with the reference to the north-star tolerances; the NaN candidate (log det <= 0) passes through as NaN."""
    z = golden("g10_readme_regime.npz")
    N, ds, da, H = (int(v) for v in z["dims"])
    sn = float(z["sigma_ns"][si])
    kinv = np.stack([z[f"s{si}_Ky_inv"]] * ds)
    pack = G.GPPack(z["X"], z["Y"], kinv, z["lambdas"], z["sigma_f"])
    assert pack.shared_lambda
    nb = z["U"].shape[0]
    for gi, gamma in enumerate(z["gammas"]):
        cost = G.CostParams(float(gamma), z["Q"], z["R"])
        for r in (G.rollout(pack, np.tile(z["x0"], (nb, 1)), z["U"], cost),):
            np.testing.assert_allclose(r["means"].cpu().numpy(), z[f"s{si}_means"], rtol=1e-7, atol=1e-10)
            np.testing.assert_allclose(r["vars"].cpu().numpy(), z[f"s{si}_vars"], rtol=1e-6)
            np.testing.assert_allclose(r["cost"].cpu().numpy(), z[f"s{si}_costs"][gi], rtol=1e-7, equal_nan=True)
            fin = np.isfinite(z[f"s{si}_costs"][gi])
            np.testing.assert_allclose(r["grad"].cpu().numpy()[fin], z[f"s{si}_grads"][gi][fin], rtol=1e-5, atol=1e-8)
            assert np.isnan(r["cost"].cpu().numpy()[~fin]).all()
    # a large batch of the same candidates takes the shared-lambda pair kernel: same values
    reps = 600
    big = G.rollout(pack, np.tile(z["x0"], (nb * reps, 1)), np.tile(z["U"], (reps, 1, 1)), G.CostParams(1e-5, z["Q"], z["R"]))
    np.testing.assert_allclose(big["vars"][-nb:].cpu().numpy(), z[f"s{si}_vars"], rtol=1e-6)
    np.testing.assert_allclose(big["grad"][-nb:].cpu().numpy(), z[f"s{si}_grads"][1], rtol=1e-5, atol=1e-8)
    # (ii) the classes, as the experiment drives them
    mpc = G.RiskSensitiveMPC(-1.0, H, ds, da, z["Q"], z["R"], None)
    for i in range(ds):
        mpc.dynamics.gpr_err[i].set_sigma_n(sn)
        mpc.dynamics.gpr_err[i].set_lambdas([0.5, 0.5, 0.5, 0.5])
        mpc.dynamics.gpr_err[i].set_sigma_f(1.)
    mpc.dynamics.append_train_data(z["X"][:, :ds], z["X"][:, ds:], z["Y"])
    mpc.set_xref(np.array([0., 0.])); mpc.set_uref(np.array([0., 0.]))
    np.testing.assert_array_equal(np.stack([g.get_lambdas() for g in mpc.dynamics.gpr_err]), z["lambdas"])
    assert mpc.dynamics.pack().shared_lambda
    mpc.curr_state = torch.tensor(z["x0"], device=mpc.device).type(torch.float64)
    # the inverse is rebuilt here (rocSOLVER LU instead of the reference's CPU LAPACK LU): cond(Ky) ~ 1 / sigma_n^2
    vtol, gtol = (1e-6, 1e-5) if sn >= 1e-3 else (1e-4, 1e-3)
    for b in range(nb):
        means, covs = mpc.dynamics.forward_propagate_torch(H, mpc.curr_state, torch.tensor(z["U"][b]))
        np.testing.assert_allclose(torch.stack(means).cpu().numpy(), z[f"s{si}_means"][b], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(torch.stack([torch.diag(c) for c in covs]).cpu().numpy(), z[f"s{si}_vars"][b], rtol=vtol)
        for gi, gamma in enumerate(z["gammas"]):
            mpc.gamma = float(gamma)
            mpc._cache_key = None
            c = mpc.objective(z["U"][b].reshape(-1).copy())
            g = mpc.gradient(z["U"][b].reshape(-1).copy())
            want = z[f"s{si}_costs"][gi, b]
            if np.isnan(want):
                assert np.isnan(c)
            else:
                np.testing.assert_allclose(c, want, rtol=max(1e-6, 10 * vtol))
                np.testing.assert_allclose(g, z[f"s{si}_grads"][gi, b], rtol=max(1e-4, 10 * gtol), atol=1e-7)


def test_pack_is_refilled_in_place_while_the_padded_size_fits(G, golden):
    """The closed loop appends one observation per step (src/simulator.py:55): Dynamics.pack() refills the SAME device pack
    (gpmpc_pack_resize + gpmpc_pack_build: no allocation) while the padded training-set size is unchanged, and allocates a new
    one when it grows past a multiple of 64.  Values: identical to a pack built from scratch on the same data; the value
    cache behind objective / gradient notices the refill although the pack object is the same."""
    z = golden("g3_rollout_c1.npz")
    N, ds, da, H = (int(v) for v in z["dims"])          # N = 100: padded 128
    mpc = _mpc_from(G, z, -1.0)
    mpc.curr_state = torch.tensor(z["x0"][0]).to(mpc.device)
    x = z["U"][0].reshape(-1).copy()
    p0 = mpc.dynamics.pack()
    c0 = mpc.objective(x)
    rng = np.random.default_rng(3)
    for k in range(30):                                   # 100 -> 130 points: the padded size changes at 129
        s, a = rng.uniform(-1, 1, ds), rng.uniform(-1, 1, da)
        mpc.dynamics.append_train_data(s, a, s + 0.1 * np.tanh(s) + 0.1 * a.sum(), incremental=False)
        pk = mpc.dynamics.pack()
        n = mpc.dynamics.gpr_err[0].num_train
        assert (pk is p0) == (n <= 128), n
        c = mpc.objective(x)
        if k == 0:
            assert c != c0                                # same pack object, new contents: not served from the cache
        if n in (101, 128, 129, 130):
            fresh = G.GPPack(mpc.dynamics.gpr_err[0].X_train, torch.cat([g.y_train.reshape(-1, 1) for g in mpc.dynamics.gpr_err], dim=1),
                             torch.stack([g.Ky_inv for g in mpc.dynamics.gpr_err]), np.stack([g.get_lambdas() for g in mpc.dynamics.gpr_err]),
                             np.array([g.get_sigma_f() for g in mpc.dynamics.gpr_err]))
            r = G.rollout(fresh, z["x0"][0], z["U"][0], mpc._cost_params())
            assert c == r["cost"][0].item(), n


def test_side_stream_rebuild_catches_up_and_swaps(G):
    """async_rebuild: the periodic full rebuild of the incremental path runs on a side stream; when it has finished, the
    observations appended meanwhile are re-applied to its result and the matrices are swapped.  The swapped inverse must be
    what a rebuild followed by those appends gives (i.e. close to a fresh rebuild: the drift since the snapshot only), the
    counter must restart from the number of caught-up observations, and predictions must agree with a from-scratch GP."""
    rng = np.random.default_rng(11)
    D, n0, extra = 3, 120, 150
    X = rng.uniform(-2, 2, (n0 + extra, D))
    y = np.sin(X).sum(axis=1)
    inc, ref = G.GaussianProcessRegression(D), G.GaussianProcessRegression(D)
    for g in (inc, ref):
        g.set_lambdas(np.array([1.5, 2.0, 1.0])); g.set_sigma_n(1e-2); g.set_sigma_f(1.2)
    inc.rebuild_every, inc.async_rebuild = 16, True
    inc.append_train_data(X[:n0], y[:n0])
    swaps, full_on_step = 0, 0
    for k in range(n0, n0 + extra):
        before = inc._appends_since_rebuild
        inc.append_train_data(X[k], float(y[k]), incremental=True)
        if inc._appends_since_rebuild == 0:
            full_on_step += 1                               # a synchronous rebuild on the step: must not happen
        elif inc._appends_since_rebuild <= before:
            swaps += 1
            # restarted from the caught-up observations: as many as the host managed to append while the side stream rebuilt (16 at
            # the round-3 append cost; more since the data path got faster)
            assert 1 <= inc._appends_since_rebuild <= 64
    assert full_on_step == 0 and swaps >= extra // 40, (full_on_step, swaps)
    inc.finish_async_rebuild()
    ref.append_train_data(X, y)                             # one rebuild from scratch
    assert inc.num_train == ref.num_train == n0 + extra
    scale = float(ref.Ky_inv.abs().max())
    # the drift of the appends since the last snapshot only: as many as the host fitted into one side-stream rebuild (16 with the round-3
    # data path: within 1e-7 of the largest element; 2-3 times as many since round 4: 1.2e-7 seen)
    np.testing.assert_allclose(inc.Ky_inv.cpu().numpy(), ref.Ky_inv.cpu().numpy(), rtol=0, atol=5e-7 * scale)
    np.testing.assert_allclose(inc.Ky.cpu().numpy(), ref.Ky.cpu().numpy(), rtol=1e-12, atol=1e-14)
    Xp = rng.uniform(-2, 2, (5, D))
    fi, ci = inc.predict_latent_vars(Xp, covar=True)
    fr, cr = ref.predict_latent_vars(Xp, covar=True)
    np.testing.assert_allclose(fi, fr, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(ci, cr, rtol=1e-5, atol=1e-9)


def test_newton_refresh_of_the_incremental_inverse(G):
    """refresh = "newton": every `rebuild_every` appends the incrementally updated inverse is polished by Newton-Schulz steps
    instead of being rebuilt (two GEMMs per step).  Never a full rebuild on a step, the polished inverse within the tolerance of a
    from-scratch one, a residual that is not safely contractive falls back to the rebuild."""
    rng = np.random.default_rng(12)
    D, n0, extra = 3, 120, 150
    X = rng.uniform(-2, 2, (n0 + extra, D))
    y = np.sin(X).sum(axis=1)
    inc, ref = G.GaussianProcessRegression(D), G.GaussianProcessRegression(D)
    for g in (inc, ref):
        g.set_lambdas(np.array([1.5, 2.0, 1.0])); g.set_sigma_n(1e-2); g.set_sigma_f(1.2)
    inc.rebuild_every, inc.refresh = 16, "newton"
    inc.append_train_data(X[:n0], y[:n0])
    refreshed = 0
    for k in range(n0, n0 + extra):
        v = inc.version
        inc.append_train_data(X[k], float(y[k]), incremental=True)
        if inc._appends_since_rebuild == 0:
            refreshed += 1
            assert inc.version == v + 2 and 1 <= inc.newton_steps_last <= 6        # append + polish, not a rebuild
            fresh = torch.linalg.inv(inc.Ky)
            scale = float(fresh.abs().max())
            assert float((inc.Ky_inv - fresh).abs().max()) <= 1e-7 * scale
    assert refreshed == extra // 16
    ref.append_train_data(X, y)
    np.testing.assert_allclose(inc.Ky.cpu().numpy(), ref.Ky.cpu().numpy(), rtol=1e-12, atol=1e-14)
    Xp = rng.uniform(-2, 2, (5, D))
    fi, ci = inc.predict_latent_vars(Xp, covar=True)
    fr, cr = ref.predict_latent_vars(Xp, covar=True)
    np.testing.assert_allclose(fi, fr, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(ci, cr, rtol=1e-4, atol=1e-8)
    # an inverse that is far off (here: of other hyper-parameters' matrix) is rebuilt, not iterated on
    inc.Ky_inv = 0.1 * inc.Ky_inv
    inc._appends_since_rebuild = inc.rebuild_every
    inc._newton_refresh()
    np.testing.assert_allclose(inc.Ky_inv.cpu().numpy(), torch.linalg.inv(inc.Ky).cpu().numpy(), rtol=0, atol=1e-9 * scale)


def test_numpy_forward_propagate_signature(G, golden):
    """Dynamics.forward_propagate (the reference's numpy rollout, src/dynamics.py:62-124): numpy in, numpy (H+1, ds) /
    (H+1, ds, ds) out, values of the torch rollout (the reference's own rung-4 test holds the two to 1e-7)."""
    z = golden("g3_rollout_c1.npz")
    N, ds, da, H = (int(v) for v in z["dims"])
    mpc = _mpc_from(G, z, -1.0)
    means, covs = mpc.dynamics.forward_propagate(H, z["x0"][0], z["U"][0])
    assert isinstance(means, np.ndarray) and means.shape == (H + 1, ds) and covs.shape == (H + 1, ds, ds)
    np.testing.assert_allclose(means, z["means"][0], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.diagonal(covs, axis1=1, axis2=2), z["vars"][0], rtol=1e-5)
    assert np.all(covs[3] - np.diag(np.diag(covs[3])) == 0)


def test_update_Ky_inv_mat_block_inverse(G):
    """GaussianProcessRegression.update_Ky_inv_mat(k_new) (src/gpr.py:137-157; the reference's block-inverse check is
    src/test/test_gpr.py:563-609): appending the covariance column of a new point to Ky_inv gives the inverse of the bordered
    matrix [[Ky, k], [k^T, sigma_n^2 + sigma_f^2]]."""
    rng = np.random.default_rng(4)
    D, n = 2, 60
    gp = G.GaussianProcessRegression(D)
    gp.set_lambdas(np.array([1.0, 2.0])); gp.set_sigma_n(0.3); gp.set_sigma_f(1.1)
    X = rng.uniform(-2, 2, (n, D))
    gp.append_train_data(X, np.sin(X).sum(axis=1))
    xn = rng.uniform(-2, 2, D)
    k = gp.compute_pred_train_covariance(xn).reshape(n, 1)               # K(x_new, X)
    Ky = gp.Ky.cpu().numpy()
    kn = k.cpu().numpy()
    full = np.block([[Ky, kn], [kn.T, np.array([[gp.get_sigma_n() ** 2 + gp.get_sigma_f() ** 2]])]])
    gp.update_Ky_inv_mat(k)
    assert gp.Ky_inv.shape == (n + 1, n + 1)
    np.testing.assert_allclose(gp.Ky_inv.cpu().numpy() @ full, np.eye(n + 1), rtol=0, atol=1e-9)


def test_autotune_measures_and_keeps_the_fastest_plan(G):
    """gpmpc_pack_autotune: candidates are timed on the device, the winner is remembered for that call shape only, results under the
    tuned plan agree with the default plan's to rounding and are bit-reproducible; autotune_clear restores the thresholds' choice."""
    from gaussian_process_mpc_amd.synth import synth_problem
    from oracle import gpmpc_oracle as O
    pb = synth_problem(77, 300, 2, 1, 6, 64)
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    before, other_before = pack.plan(64, 6), pack.plan(32, 6)
    r0 = G.rollout(pack, pb["x0"], pb["U"], cost)
    res = pack.autotune(64, 6)
    timed = [c for c in res if c["ms"] > 0]
    assert len(timed) >= 3 and sum(c["winner"] for c in res) == 1
    win = next(c for c in res if c["winner"])
    assert win["ms"] <= min(c["ms"] for c in timed) * 1.0000001 or win is res[0]      # the fastest, or the default within 2 % of it
    assert res[0]["ms"] > 0 and res[0]["name"] == "default"
    r1 = G.rollout(pack, pb["x0"], pb["U"], cost)
    r2 = G.rollout(pack, pb["x0"], pb["U"], cost)
    for k in r1:
        assert torch.equal(r1[k], r2[k]), k
    np.testing.assert_allclose(r1["means"].cpu().numpy(), r0["means"].cpu().numpy(), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r1["vars"].cpu().numpy(), r0["vars"].cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(r1["grad"].cpu().numpy(), r0["grad"].cpu().numpy(), rtol=1e-4, atol=1e-8)
    assert pack.plan(32, 6) == other_before                           # other call shapes keep the thresholds' choice
    pack.autotune_clear()
    assert pack.plan(64, 6) == before
    r3 = G.rollout(pack, pb["x0"], pb["U"], cost)
    for k in r0:
        assert torch.equal(r0[k], r3[k]), k


def test_default_plan_is_within_reach_of_the_measured_best_between_generations(G):
    """Round 5 (review item 5): in the region the round-4 rule was more than 5 % off -- small training sets at batches between one and
    two generations of whole-horizon workgroups, B ~ 260...450 -- the plan chosen by the cost model (step.hip::plan_rollout) is held to the
    plan gpmpc_pack_autotune measures as fastest on this device: within 10 % on each of six shapes (5 % is the grid's own bar,
    profiles/r05/autotune_grid.txt; the extra margin is timing noise of a sub-millisecond measurement inside a test run), and at most
    two of the six beyond 5 %."""
    from gaussian_process_mpc_amd.synth import synth_problem
    from oracle import gpmpc_oracle as O
    shapes = [(300, 4, 1, 10, 320), (300, 4, 1, 10, 448), (200, 4, 1, 10, 320), (300, 2, 1, 10, 320), (512, 3, 1, 20, 224), (400, 3, 2, 15, 384)]
    ratios, packs = [], {}
    for N, ds, da, H, B in shapes:
        if (N, ds, da) not in packs:
            packs.clear(); torch.cuda.empty_cache()
            pb = synth_problem(3, N, ds, da, H, 8)
            kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
            packs[(N, ds, da)] = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
        pack = packs[(N, ds, da)]
        pack.autotune_clear()
        res = pack.autotune(B, H, graph=True)
        best = min(c["ms"] for c in res if c["ms"] > 0)
        assert res[0]["name"] == "default" and res[0]["ms"] > 0
        ratios.append(res[0]["ms"] / best)
        pack.autotune_clear()
    assert max(ratios) <= 1.10, list(zip(shapes, ratios))
    assert sum(r > 1.05 for r in ratios) <= 2, list(zip(shapes, ratios))


def test_gp_append_into_padded_buffers_and_strided_pack_build(G):
    """C ABI gpmpc_gp_append (one call: k vector, Schur step on Ky_inv, new row / column of Kf and Ky, written into buffers of another
    leading dimension) against torch on the n + 1 points, and gpmpc_pack_build_strided: a pack built from a strided view / ONE shared
    inverse equals the pack built from packed copies bit for bit."""
    import ctypes
    from gaussian_process_mpc_amd._lib import lib, ptr, stream_ptr, host_doubles, check
    dev = G.require_gpu()
    rng = np.random.default_rng(5)
    n, D, cap = 75, 3, 128
    X = torch.tensor(rng.uniform(-2, 2, (n + 1, D)), device=dev)
    lam, sf, noise = np.array([0.7, 1.3, 2.0]), 1.2, 1e-2
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2 / torch.tensor(lam, device=dev)).sum(-1)
    Kf_full = sf ** 2 * torch.exp(-0.5 * d2)
    Ky_full = Kf_full + noise * torch.eye(n + 1, dtype=torch.float64, device=dev)
    Kf, Ky = Kf_full[:n, :n].contiguous(), Ky_full[:n, :n].contiguous()
    Kinv = torch.linalg.inv(Ky)
    out = [torch.full((cap, cap), float("nan"), dtype=torch.float64, device=dev) for _ in range(3)]
    nb = lib().gpmpc_gp_append_workspace_bytes(n, D)
    ws = torch.empty(int(nb), dtype=torch.uint8, device=dev)
    _, lp = host_doubles(lam)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    check(lib().gpmpc_gp_append(n, D, ptr(X[:n].contiguous()), ptr(X[n:n + 1].contiguous()), lp, sf, noise, vp(Kf), vp(Ky), n, vp(Kinv), n,
                                vp(out[0]), vp(out[1]), vp(out[2]), cap, ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()), "gpmpc_gp_append")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[0][:n + 1, :n + 1].cpu().numpy(), Kf_full.cpu().numpy(), rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(out[1][:n + 1, :n + 1].cpu().numpy(), Ky_full.cpu().numpy(), rtol=1e-14, atol=1e-15)
    ref = torch.linalg.inv(Ky_full)
    np.testing.assert_allclose(out[2][:n + 1, :n + 1].cpu().numpy(), ref.cpu().numpy(), rtol=1e-7, atol=1e-9 * float(ref.abs().max()))
    assert torch.isnan(out[2][n + 1:, :]).all() and torch.isnan(out[2][:, n + 1:]).all()          # nothing beyond the (n + 1) block is touched
    # argument checks: aliasing and too small a leading dimension
    assert lib().gpmpc_gp_append(n, D, ptr(X[:n].contiguous()), ptr(X[n:n + 1].contiguous()), lp, sf, noise, vp(Kf), vp(Ky), n, vp(Kinv), n,
                                 vp(Kf), vp(out[1]), vp(out[2]), cap, ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()) == -1
    assert lib().gpmpc_gp_append(n, D, ptr(X[:n].contiguous()), ptr(X[n:n + 1].contiguous()), lp, sf, noise, vp(Kf), vp(Ky), n, vp(Kinv), n,
                                 vp(out[0]), vp(out[1]), vp(out[2]), n, ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()) == -1
    # strided pack build: a view of the padded buffer, shared by two GPs, against packed copies
    Y = torch.tensor(rng.normal(size=(n + 1, 2)), device=dev)
    lam2 = np.stack((lam, lam))
    view = out[2][:n + 1, :n + 1]
    p_view = G.GPPack(X, Y, view, lam2, np.array([sf, sf]))
    p_copy = G.GPPack(X, Y, torch.stack((view.contiguous(), view.contiguous())), lam2, np.array([sf, sf]))
    assert torch.equal(p_view.weights(), p_copy.weights()) and torch.equal(p_view.beta(), p_copy.beta())
