"""Pin the CPU oracle to the golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import gpmpc_oracle as O

T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
TIGHT = dict(rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("mode", ["faithful", "o2"])
def test_g1_single_step(golden, tag, mode):
    z = golden("g1_single_step.npz")
    u, S = T(z["u"]), T(z["S"])
    X, y = T(z[f"{tag}_X"]), T(z[f"{tag}_y"])
    sf1, sf2 = z[f"{tag}_sf"]
    m1, b1, l1 = O.mean_prop(T(z[f"{tag}_Kinv1"]), T(z["lam1"]), u, S, X, y, sf1)
    m2, b2, _ = O.mean_prop(T(z[f"{tag}_Kinv2"]), T(z["lam2"]), u, S, X, y, sf2)
    v1 = O.variance_prop(T(z[f"{tag}_Kinv1"]), T(z["lam1"]), u, S, X, m1, b1, sf1, mode)
    v2 = O.variance_prop(T(z[f"{tag}_Kinv2"]), T(z["lam2"]), u, S, X, m2, b2, sf2, mode)
    cv = O.covariance_prop(T(z["lam1"]), T(z["lam2"]), u, S, X, m1, m2, b1, b2, sf1, sf2)
    np.testing.assert_allclose([m1.item(), m2.item()], z[f"{tag}_mu"], **TIGHT)
    np.testing.assert_allclose(b1.numpy(), z[f"{tag}_beta1"], **TIGHT)
    np.testing.assert_allclose(l1.numpy(), z[f"{tag}_l1"], **TIGHT)
    vtol = TIGHT if mode == "faithful" else dict(rtol=1e-8)
    np.testing.assert_allclose([v1.item(), v2.item()], z[f"{tag}_var"], **vtol)
    np.testing.assert_allclose(cv.item(), z[f"{tag}_cov"], rtol=1e-10)


def test_g1_numpy_loop_rung(golden):
    """Rung 2 of the reference's ladder: loop formulas vs vectorised ones,
    with the reference's own absolute tolerances (test_uncertainty_prop.py:182-385)."""
    z = golden("g1_single_step.npz")
    X, y, u, S = z["a_X"], z["a_y"], z["u"], z["S"]
    Ky1 = np.linalg.inv(z["a_Kinv1"])
    Ky2 = np.linalg.inv(z["a_Kinv2"])
    mu, beta, l = O.mean_prop_loops(Ky1, z["lam1"], u, S, X, y)
    assert abs(mu - z["a_np_mu"]) < 1e-7 * max(1, abs(mu))
    assert abs(mu - z["a_mu"][0]) < 1e-7 * max(1, abs(mu))
    assert np.linalg.norm(l - z["a_l1"]) < 1e-5
    var = O.variance_prop_loops(Ky1, z["lam1"], u, S, X, y)
    assert abs(var - z["a_np_var"]) < 1e-5 * abs(var)
    assert abs(var - z["a_var"][0]) < 1e-5 * abs(var)


@pytest.mark.parametrize("k", range(6))
def test_g2_adversarial(golden, k):
    z = golden("g2_adversarial.npz")
    p = f"c{k}_"
    sf1, sf2, _ = z[p + "hyp"]
    X = T(z[p + "X"])
    u = T(z[p + "u"]).requires_grad_(True)
    S = T(z[p + "S"]).requires_grad_(True)
    m1, b1, _ = O.mean_prop(T(z[p + "Kinv1"]), T(z[p + "lam1"]), u, S, X, T(z[p + "y1"]), sf1)
    v1 = O.variance_prop(T(z[p + "Kinv1"]), T(z[p + "lam1"]), u, S, X, m1, b1, sf1)
    m2, b2, _ = O.mean_prop(T(z[p + "Kinv2"]), T(z[p + "lam2"]), u, S, X, T(z[p + "y2"]), sf2)
    v2 = O.variance_prop(T(z[p + "Kinv2"]), T(z[p + "lam2"]), u, S, X, m2, b2, sf2)
    np.testing.assert_allclose([m1.item(), m2.item()], z[p + "mu"], **TIGHT)
    np.testing.assert_allclose([v1.item(), v2.item()], z[p + "var"], **TIGHT)
    dm_du, dm_dS = torch.autograd.grad(m1, (u, S), retain_graph=True)
    dv_du, dv_dS = torch.autograd.grad(v1, (u, S), retain_graph=True)
    np.testing.assert_allclose(dm_du.numpy(), z[p + "dm_du"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dm_dS.numpy(), z[p + "dm_dS"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dv_du.numpy(), z[p + "dv_du"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dv_dS.numpy(), z[p + "dv_dS"], rtol=1e-8, atol=1e-10)
    cv = O.covariance_prop(T(z[p + "lam1"]), T(z[p + "lam2"]), u, S, X, m1, m2, b1, b2, sf1, sf2)
    np.testing.assert_allclose(cv.item(), z[p + "cov_torch"], rtol=1e-10)
    # unit sigma_f / shared y: bug-compatible == reference torch, corrected == reference numpy loop
    ud, Sd = T(z[p + "u"]), T(z[p + "S"])
    y = T(z[p + "y1"])
    mu1, bu1, _ = O.mean_prop(T(z[p + "unit_Kinv1"]), T(z[p + "lam1"]), ud, Sd, X, y)
    mu2, bu2, _ = O.mean_prop(T(z[p + "unit_Kinv2"]), T(z[p + "lam2"]), ud, Sd, X, y)
    bug = O.covariance_prop(T(z[p + "lam1"]), T(z[p + "lam2"]), ud, Sd, X, mu1, mu2, bu1, bu2)
    fix = O.covariance_prop(T(z[p + "lam1"]), T(z[p + "lam2"]), ud, Sd, X, mu1, mu2, bu1, bu2, bug_compatible=False)
    np.testing.assert_allclose(bug.item(), z[p + "unit_cov_torch"], rtol=1e-10)
    np.testing.assert_allclose(fix.item(), z[p + "unit_cov_numpy"], rtol=1e-8)
    if z["cases"][k][1] == 0:      # diagonal S: the reference's transposition is harmless
        np.testing.assert_allclose(bug.item(), fix.item(), rtol=1e-9)


def _bundle(z):
    return O.GPBundle(z["X"], z["Y"], z["lambdas"], z["sigma_f"], z["sigma_n"], Ky_inv=z["Ky_inv"])


@pytest.mark.parametrize("name", ["g3_rollout_c1.npz", "g4_rollout_c2.npz"])
@pytest.mark.parametrize("mode", ["faithful", "o2"])
def test_rollout_cost_gradient(golden, name, mode):
    z = golden(name)
    N, ds, da, H = z["dims"]
    gp = _bundle(z)
    x_ref = z["x_ref"] if "x_ref" in z else np.zeros(ds)
    u_ref = z["u_ref"] if "u_ref" in z else np.zeros(da)
    Rd = z["R_delta"] if "R_delta" in z else None
    last = z["last_traj"][:da] if "last_traj" in z else None
    for b in range(z["x0"].shape[0]):
        for gi, gamma in enumerate(z["gammas"]):
            r = O.objective_and_gradient(gp, int(H), z["x0"][b], z["U"][b], x_ref, u_ref, z["Q"], z["R"],
                                         float(gamma), Rd, last, mode)
            # The reference's own run-to-run reproducibility (BLAS thread count / reduction
            # order through the cancelling trace) is ~1e-10 on means after 10-20 steps.
            np.testing.assert_allclose(r["means"], z["means"][b], rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(r["vars"], z["vars"][b], rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(r["cost"], z["costs"][gi, b], rtol=1e-7)
            np.testing.assert_allclose(r["grad"], z["grads"][gi, b], rtol=1e-6, atol=1e-9)


def test_kinv_rebuild_matches_reference(golden):
    """Ky_inv rebuilt by the oracle (explicit inverse, gpr.py:171) equals the stored one."""
    z = golden("g3_rollout_c1.npz")
    gp = O.GPBundle(z["X"], z["Y"], z["lambdas"], z["sigma_f"], z["sigma_n"])
    scale = np.abs(z["Ky_inv"]).max()      # cond(Ky) ~ 1e6-1e7 at sigma_n = 1e-2
    np.testing.assert_allclose(gp.Ky_inv.numpy(), z["Ky_inv"], rtol=0, atol=1e-6 * scale)


def test_g5_cost_known_answers(golden):
    z = golden("g5_cost.npz")
    x, u, sig = T(z["a_x"]), T(z["a_u"]), T(z["a_sig"])
    c = O.cost([x[0], x[1]], u, [sig[0], sig[1]], T(z["a_xref"]), T(z["a_uref"]), z["a_Q"], z["a_R"], 1.0)
    assert abs(c.item() - z["a_cost_np"]) < 1e-6        # test_mpc.py:57
    assert abs(c.item() - z["a_cost_torch"]) < 1e-12
    x, u, sig = T(z["b_x"]), T(z["b_u"]), T(z["b_sig"])
    c = O.cost(list(x), u, list(sig), T(z["a_xref"]), T(z["a_uref"]), z["a_Q"], z["a_R"], 1.1,
               R_delta=z["b_Rd"], last_u=z["b_last"][:2])
    assert abs(c.item() - z["b_cost_torch"]) < 1e-12
    H = 5
    c = O.cost(list(T(z["c_x"]).reshape(H + 1, 1)), torch.zeros((H, 1), dtype=torch.float64),
               list(T(z["c_sig"]).reshape(H + 1, 1, 1)), torch.zeros(1, dtype=torch.float64),
               torch.zeros(1, dtype=torch.float64), 2 * np.eye(1), np.zeros((1, 1)), -1.0,
               R_delta=np.zeros((1, 1)), last_u=np.zeros(1))
    assert abs(c.item() - z["c_closed"]) < 1e-7          # test_mpc.py:274


def test_g6_gp(golden):
    z = golden("g6_gp.npz")
    # the fixture was built with set_sigma_f(1.4) / set_sigma_n(0.2): Python floats -> float32 log
    sf, sn = (float(O.effective_hyper(float(v))) for v in z["hyp"])
    assert sf != 1.4 and abs(sf - 1.4) < 1e-7
    Kf, Ky, Ki = O.kernel_matrices(z["X"], z["lam"], sf, sn)
    np.testing.assert_allclose(Kf.numpy(), z["Kf"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(Ky.numpy(), z["Ky"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(Ki.numpy(), z["Ky_inv"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(O.cross_kernel(z["Xp"], z["X"], T(z["lam"]), sf).numpy(), z["Ks"], rtol=1e-12)
    np.testing.assert_allclose(O.cross_kernel(z["Xp"][0], z["X"], T(z["lam"]), sf).numpy(), z["Ks_single"], rtol=1e-12)
    f, cov = O.predict(z["Xp"], z["X"], z["y"], z["Ky_inv"], z["lam"], sf, sn, covar=True)
    np.testing.assert_allclose(f, z["f"], rtol=1e-10)
    np.testing.assert_allclose(cov, z["cov_f"], rtol=1e-8, atol=1e-10)
    _, covy = O.predict(z["Xp"], z["X"], z["y"], z["Ky_inv"], z["lam"], sf, sn, covar=True, targets=True)
    np.testing.assert_allclose(covy, z["cov_y"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(covy - cov, sn ** 2 * np.eye(len(f)), atol=1e-12)   # test_gpr.py identity


def test_g7_quirks(golden):
    z = golden("g7_quirks.npz")
    assert O.INIT_STATE_VAR == z["init_var"].item() == 1e-3
    assert O.ACTION_NOISE_VAR == z["action_var"].item() == z["float32_1e3"].item()
    assert O.ACTION_NOISE_VAR != 1e-3


@pytest.mark.parametrize("gamma", [-1.0, 1e-5, 0.0])
def test_cport_matches_torch_oracle(gamma):
    """The plain-C / OpenMP port (oracle/cport: analytic adjoint, direct exponent, libm exp) against the torch oracle
    (autograd), which the tests above pin to the reference.  Independent evaluation order and an independent check of
    the closed-form gradient on the CPU."""
    from oracle import cport
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(7, 150, 3, 2, 4, 2)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    r = cport.rollout(pb, gp.Ky_inv.numpy(), gamma, nthreads=4)
    for b in range(2):
        o = O.objective_and_gradient(gp, 4, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], gamma, mode="o2")
        np.testing.assert_allclose(r["means"][b], o["means"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r["vars"][b], o["vars"], rtol=1e-6)
        np.testing.assert_allclose(r["cost"][b], o["cost"], rtol=1e-8)
        np.testing.assert_allclose(r["grad"][b], o["grad"], rtol=1e-6, atol=1e-9)


def test_cport_matches_reference_fixture(golden):
    """... and directly against the reference's own rollout / cost / gradient (g4 fixture; R_delta-free gamma rows of g3)."""
    from oracle import cport
    z = golden("g3_rollout_c1.npz")
    pb = {"X": z["X"], "Y": z["Y"], "lambdas": z["lambdas"], "sigma_f": z["sigma_f"], "ds": 2, "da": 2,
          "Q": z["Q"], "R": z["R"], "x_ref": np.zeros(2), "u_ref": np.zeros(2), "x0": z["x0"], "U": z["U"]}
    for gi, gamma in enumerate(z["gammas"]):
        r = cport.rollout(pb, z["Ky_inv"], float(gamma), nthreads=4)
        np.testing.assert_allclose(r["means"], z["means"], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(r["vars"], z["vars"], rtol=1e-6)
        np.testing.assert_allclose(r["cost"], z["costs"][gi], rtol=1e-7)
        np.testing.assert_allclose(r["grad"], z["grads"][gi], rtol=1e-6, atol=1e-9)


def test_extended_precision_yardstick_matches_reference_fixture(golden):
    """The x87 extended-precision forward rollout (oracle/cport/gpmpc_cpu_ld.c; the yardstick of the noise-level accuracy
    sweep) against the reference's own rollout (g3 fixture) and against the fp64 C port on a seeded problem."""
    from oracle import cport
    from gaussian_process_mpc_amd.synth import synth_problem
    z = golden("g3_rollout_c1.npz")
    pb = {"X": z["X"], "Y": z["Y"], "lambdas": z["lambdas"], "sigma_f": z["sigma_f"], "ds": 2, "da": 2,
          "x0": z["x0"], "U": z["U"]}
    e = cport.rollout_extended(pb, z["Ky_inv"], nthreads=4)
    np.testing.assert_allclose(e["means"], z["means"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(e["vars"], z["vars"], rtol=1e-6)
    pb = synth_problem(19, 150, 3, 1, 4, 2)
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    e, c = cport.rollout_extended(pb, kinv, nthreads=4), cport.rollout(pb, kinv, -1.0, nthreads=4)
    np.testing.assert_allclose(c["means"], e["means"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(c["vars"], e["vars"], rtol=1e-6)
    e1 = cport.rollout_extended(pb, kinv, nthreads=1)                 # fixed-order reductions: independent of the thread count
    assert np.array_equal(e1["vars"], e["vars"]) and np.array_equal(e1["means"], e["means"])
    c1 = cport.rollout(pb, kinv, -1.0, nthreads=1)
    assert np.array_equal(c1["vars"], c["vars"]) and np.array_equal(c1["grad"], c["grad"])


@pytest.mark.parametrize("ds,da", [(5, 1), (6, 1), (4, 2), (5, 2), (1, 2)])
def test_cport_matches_torch_oracle_input_dims(ds, da):
    """The C port at the input dimensions D = 6, 7 (BASELINE config 4 is ds = 6, da = 1) the GPU instance tests
    (tests/test_gpu_instances.py) use it for: values and analytic adjoint against the torch oracle's autograd."""
    from oracle import cport
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(17, 70, ds, da, 3, 1)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    r = cport.rollout(pb, gp.Ky_inv.numpy(), -1.0, nthreads=4)
    o = O.objective_and_gradient(gp, 3, pb["x0"][0], pb["U"][0], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0, mode="o2")
    np.testing.assert_allclose(r["means"][0], o["means"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(r["vars"][0], o["vars"], rtol=1e-6)
    np.testing.assert_allclose(r["cost"][0], o["cost"], rtol=1e-8)
    np.testing.assert_allclose(r["grad"][0], o["grad"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("k", range(6))
def test_cport_fullcov_single_step_matches_reference_fixture(golden, k):
    """The full-covariance C port (oracle/cport/gpmpc_cpu_fullcov.c), one step at a given (u, S), against the
    reference's own outputs (g2): mean_prop_torch / variance_prop_torch values and the reference's numpy double loop
    covariance_prop (the consistent cross term; the reference's torch version is index-transposed for full S)."""
    from oracle import cport
    z = golden("g2_adversarial.npz")
    p = f"c{k}_"
    sf1, sf2, _ = z[p + "hyp"]
    lam = np.stack([z[p + "lam1"], z[p + "lam2"]])
    m, c = cport.moment_match_fullcov(z[p + "X"], np.stack([z[p + "Kinv1"], z[p + "Kinv2"]]),
                                      np.stack([z[p + "y1"], z[p + "y2"]], axis=1), lam, [sf1, sf2], z[p + "u"], z[p + "S"], nthreads=2)
    np.testing.assert_allclose(m, z[p + "mu"], rtol=1e-11)
    np.testing.assert_allclose(np.diag(c), z[p + "var"], rtol=1e-9)
    _, cu = cport.moment_match_fullcov(z[p + "X"], np.stack([z[p + "unit_Kinv1"], z[p + "unit_Kinv2"]]),
                                       np.stack([z[p + "y1"], z[p + "y1"]], axis=1), lam, [1.0, 1.0], z[p + "u"], z[p + "S"], nthreads=2)
    np.testing.assert_allclose(cu[0, 1], z[p + "unit_cov_numpy"], rtol=1e-9)
    assert cu[0, 1] == cu[1, 0]
    if z["cases"][k][1] == 0:      # diagonal S: the reference's torch form agrees too
        np.testing.assert_allclose(cu[0, 1], z[p + "unit_cov_torch"], rtol=1e-9)


def test_cport_fullcov_g1_test_suite_geometry(golden):
    """... and on the reference test-suite geometry (g1: full S, proportional lambdas, where the reference's torch and
    numpy covariances agree): mean, variance, covariance."""
    from oracle import cport
    z = golden("g1_single_step.npz")
    lam = np.stack([z["lam1"], z["lam2"]])
    for tag in ("a", "b", "c"):
        y = z[tag + "_y"]
        m, c = cport.moment_match_fullcov(z[tag + "_X"], np.stack([z[tag + "_Kinv1"], z[tag + "_Kinv2"]]), np.stack([y, y], axis=1),
                                          lam, z[tag + "_sf"], z["u"], z["S"], nthreads=2)
        np.testing.assert_allclose(m, z[tag + "_mu"], rtol=1e-10)
        np.testing.assert_allclose(np.diag(c), z[tag + "_var"], rtol=1e-7)
        np.testing.assert_allclose(c[0, 1], z[tag + "_cov"], rtol=1e-7)


@pytest.mark.parametrize("ds,da,gamma", [(3, 2, -1.0), (2, 1, 1e-5), (4, 1, 0.0), (5, 2, 0.7)])
def test_cport_fullcov_matches_torch_oracle(ds, da, gamma):
    """Whole full-covariance rollouts of the C port against the torch extension oracle (forward_propagate_fullcov /
    objective_and_gradient_fullcov): means, covariances, cost, and the complex-step directional derivatives of the cost
    against autograd.  This is what licenses the port as the full-size checker of BASELINE config 5 on the GPU box."""
    from oracle import cport
    from gaussian_process_mpc_amd.synth import synth_problem
    H, B = 3, 2
    pb = synth_problem(19, 80, ds, da, H, B)
    pb["Q"] = pb["Q"] + 0.02 * (np.ones((ds, ds)) - np.eye(ds))
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    dirs = np.random.default_rng(3).normal(size=(B, 2, H, da))
    r = cport.rollout_fullcov(pb, gp.Ky_inv.numpy(), gamma, dirs=dirs, nthreads=4)
    for b in range(B):
        o = O.objective_and_gradient_fullcov(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], gamma)
        np.testing.assert_allclose(r["means"][b], o["means"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(r["covs"][b], o["covs"], rtol=1e-6, atol=1e-9 * np.abs(o["covs"]).max())
        np.testing.assert_allclose(r["cost"][b], o["cost"], rtol=1e-9)
        for d in range(2):
            np.testing.assert_allclose(r["ddir"][b, d], float((o["grad"] * dirs[b, d]).sum()), rtol=1e-7, atol=1e-10)


def test_g8_marginal_likelihood(golden):
    """compute_marginal_likelihood at set hyper-parameters (src/gpr.py:240-251)."""
    d = golden("g8_hyper.npz")
    lam = torch.as_tensor(O.effective_hyper(d["ml_lam"]))
    sf, sn = (float(O.effective_hyper(float(v))) for v in d["ml_hyp"])
    ml = O.marginal_likelihood(d["ml_X"], d["ml_y"], lam, sf, sn).item()
    assert ml == pytest.approx(float(d["ml_value"]), rel=1e-12)


@pytest.mark.parametrize("tag,x_dim,nominal", [("t1", 1, False), ("t2", 1, True), ("t3", 3, False)])
def test_g8_update_hyperparams(golden, tag, x_dim, nominal):
    """update_hyperparams trajectories (src/gpr.py:334-370): likelihood, autograd gradients and the Adam iterates."""
    d = golden("g8_hyper.npz")
    X, y = d[tag + "_X"], d[tag + "_y"]
    tr = O.HyperTrainer(X, y, x_dim, nominal=X[:, 0] if nominal else None)
    for k in range(len(d[tag + "_ml"])):
        h = tr.step()
        assert h["ml"] == pytest.approx(d[tag + "_ml"][k], rel=1e-10)
        np.testing.assert_allclose(h["grad"]["log_lambdas"], d[tag + "_g_log_lambdas"][k], rtol=1e-7, atol=1e-9)
        assert h["grad"]["log_sigma_f"] == pytest.approx(d[tag + "_g_log_sigma_f"][k], rel=1e-7, abs=1e-9)
        assert h["grad"]["log_sigma_n"] == pytest.approx(d[tag + "_g_log_sigma_n"][k], rel=1e-6, abs=1e-7)
        np.testing.assert_allclose(h["log_lambdas"], d[tag + "_log_lambdas"][k], rtol=1e-8, atol=1e-10)
        assert h["log_sigma_f"] == pytest.approx(d[tag + "_log_sigma_f"][k], rel=1e-8, abs=1e-10)
        assert h["log_sigma_n"] == pytest.approx(d[tag + "_log_sigma_n"][k], rel=1e-8, abs=1e-10)


def test_g10_readme_regime(golden):
    """The reference's OWN regime (README experiment data, one lambda for every GP, sigma_n = 1e-3 and the experiments'
    1e-5): oracle (both modes) and C port against the reference's rollout / cost / gradient, NaN candidate included
    (gamma = -1 with Q = 2 I: a plan that leaves the data has log det <= 0; passed through, src/mpc.py:183).
    The fixture's lambdas are what the rollout READS (exp of the float32 log the setter stores, src/gpr.py:51-60), not 0.5."""
    from oracle import cport
    z = golden("g10_readme_regime.npz")
    N, ds, da, H = (int(v) for v in z["dims"])
    assert np.all(z["lambdas_set"] == 0.5) and np.all(z["lambdas"] != 0.5) and np.allclose(z["lambdas"], 0.5, rtol=1e-7)
    for si, sn in enumerate(z["sigma_ns"]):
        kinv = np.stack([z[f"s{si}_Ky_inv"]] * ds)
        gp = O.GPBundle(z["X"], z["Y"], z["lambdas"], z["sigma_f"], np.full(ds, sn), Ky_inv=kinv)
        for mode in ("faithful", "o2"):
            for gi, gamma in enumerate(z["gammas"]):
                for b in range(z["U"].shape[0]):
                    r = O.objective_and_gradient(gp, H, z["x0"], z["U"][b], np.zeros(ds), np.zeros(da), z["Q"], z["R"],
                                                 float(gamma), None, None, mode)
                    np.testing.assert_allclose(r["means"], z[f"s{si}_means"][b], rtol=1e-8, atol=1e-11)
                    np.testing.assert_allclose(r["vars"], z[f"s{si}_vars"][b], rtol=1e-7)
                    np.testing.assert_allclose(r["cost"], z[f"s{si}_costs"][gi, b], rtol=1e-8, equal_nan=True)
                    np.testing.assert_allclose(r["grad"], z[f"s{si}_grads"][gi, b], rtol=1e-6, atol=1e-9, equal_nan=True)
        pb = {"X": z["X"], "Y": z["Y"], "lambdas": z["lambdas"], "sigma_f": z["sigma_f"], "ds": ds, "da": da, "Q": z["Q"],
              "R": z["R"], "x_ref": np.zeros(ds), "u_ref": np.zeros(da), "x0": np.tile(z["x0"], (z["U"].shape[0], 1)), "U": z["U"]}
        for gi, gamma in enumerate(z["gammas"]):
            c = cport.rollout(pb, kinv, float(gamma), nthreads=4)
            np.testing.assert_allclose(c["means"], z[f"s{si}_means"], rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(c["vars"], z[f"s{si}_vars"], rtol=1e-7)
            np.testing.assert_allclose(c["cost"], z[f"s{si}_costs"][gi], rtol=1e-8, equal_nan=True)
            fin = np.isfinite(z[f"s{si}_costs"][gi])
            np.testing.assert_allclose(c["grad"][fin], z[f"s{si}_grads"][gi][fin], rtol=1e-6, atol=1e-9)
        # this regime is benign: the reference's fp64 variances sit within 1e-7 of the extended-precision evaluation at both
        # noise levels (predictive variances of 0.02-0.06 against sigma_f^2 = 1: mild cancellation)
        e = cport.rollout_extended(pb, kinv, nthreads=4)
        np.testing.assert_allclose(z[f"s{si}_vars"], e["vars"], rtol=1e-7)


def test_g11_reference_scalars_at_full_size(golden):
    """One hop from the reference at N = 2048: mean / variance of every GP for two single-step queries, produced by the
    reference itself on the C3 training set (re-derived here from the seed; the inverse is rebuilt on the CPU as the
    reference does, src/gpr.py:171) -- against the oracle's O(N^2) mode for all of them and its faithful mode (N^3 trace)
    for one."""
    from gaussian_process_mpc_amd.synth import synth_problem
    z = golden("g11_fullsize_pin.npz")
    N, ds, da = (int(v) for v in z["dims"])
    pb = synth_problem(int(z["seed"][0]), N, ds, da, 20, 8)
    torch.set_num_threads(8)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    T = lambda a: torch.as_tensor(a, dtype=torch.float64)     # noqa: E731
    for q in range(2):
        for a in range(ds):
            m, beta, _ = O.mean_prop(gp.Ky_inv[a], gp.lambdas[a], T(z["u"][q]), T(z["S"][q]), gp.X, gp.Y[:, a])
            if q == 0:
                # the reference's setters store float32 logs (src/gpr.py:51-88): its effective lambda / sigma_n differ from the
                # set values by ~6e-8 relative, which is what is left here
                np.testing.assert_allclose(float(beta.abs().sum()), z["beta_abs_sum"][a], rtol=1e-6)
            v = O.variance_prop(gp.Ky_inv[a], gp.lambdas[a], T(z["u"][q]), T(z["S"][q]), gp.X, m, beta,
                                mode="faithful" if (q, a) == (1, 2) else "o2")
            np.testing.assert_allclose(m.item(), z["mean"][q, a], rtol=1e-7)
            np.testing.assert_allclose(v.item(), z["var"][q, a], rtol=1e-5)


def test_g12_reference_rollout_at_full_size(golden):
    """Oracle (O(N^2) mode) against the reference's own objective / gradient / trajectory at N = 2048, H = 2 (g12)."""
    from gaussian_process_mpc_amd.synth import synth_problem
    z = golden("g12_fullsize_rollout.npz")
    N, ds, da, H = (int(v) for v in z["dims"])
    pb = synth_problem(int(z["seed"][0]), N, ds, da, 20, 8)
    torch.set_num_threads(8)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    b = int(z["traj"][0])
    r = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b, :H], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], float(z["gamma"][0]),
                                 mode="o2")
    np.testing.assert_allclose(r["means"], z["means"][0], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r["vars"], z["vars"][0], rtol=1e-5)
    np.testing.assert_allclose(r["cost"], z["cost"][0], rtol=1e-7)
    np.testing.assert_allclose(r["grad"], z["grad"][0], rtol=1e-4, atol=1e-8)
