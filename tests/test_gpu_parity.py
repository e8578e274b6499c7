"""GPU parity tests: the HIP path (through the C ABI) against the committed golden vectors
(produced by the reference itself) and against the CPU oracle on seeded inputs.

Tolerances (north star): state means 1e-5 relative, variances 1e-4 relative, fp64.
The asserted tolerances below are tighter where the arithmetic allows it.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEAN_RTOL = 1e-5      # north-star bound; observed ~1e-10
VAR_RTOL = 1e-4       # north-star bound; observed ~1e-7


@pytest.fixture(scope="module")
def G():
    import gaussian_process_mpc_amd as g
    g.require_gpu()
    return g


def _pack_from(G, z):
    return G.GPPack(z["X"], z["Y"], z["Ky_inv"], z["lambdas"], z["sigma_f"])


def _cost_from(G, z, gamma):
    ds, da = int(z["dims"][1]), int(z["dims"][2])
    return G.CostParams(gamma, z["Q"], z["R"],
                        R_delta=z["R_delta"] if "R_delta" in z else None,
                        x_ref=z["x_ref"] if "x_ref" in z else np.zeros(ds),
                        u_ref=z["u_ref"] if "u_ref" in z else np.zeros(da),
                        last_u=z["last_traj"][:da] if "last_traj" in z else None)


def test_pack_constants(G, golden):
    """beta and the folded weight matrix against their definitions (numpy, from the fixture)."""
    z = golden("g3_rollout_c1.npz")
    pack = _pack_from(G, z)
    N, ds = z["X"].shape[0], z["Y"].shape[1]
    beta = pack.beta().cpu().numpy()
    W = pack.weights().cpu().numpy()
    for a in range(ds):
        b_ref = z["Ky_inv"][a] @ z["Y"][:, a]
        np.testing.assert_allclose(beta[a], b_ref, rtol=1e-9, atol=1e-9 * np.abs(b_ref).max())
        d = z["X"][:, None, :] - z["X"][None, :, :]
        lam_part = np.exp(-0.25 * np.sum(d * d / z["lambdas"][a], axis=2))
        Wsym = 0.5 * (z["Ky_inv"][a] + z["Ky_inv"][a].T) - np.outer(b_ref, b_ref)
        M = Wsym * lam_part * z["sigma_f"][a] ** 4
        ref = np.triu(M, 1) * 2 + np.diag(np.diag(M))          # element (i<=j)
        got = W[a].T[:N, :N]                                    # stored at [j][i]
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
        assert np.all(W[a][:, N:] == 0) and np.all(W[a][N:, :] == 0)
        assert np.all(np.tril(got, -1) == 0)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g1_single_step(G, golden, tag):
    z = golden("g1_single_step.npz")
    X, y = z[f"{tag}_X"], z[f"{tag}_y"]
    Ki = np.stack([z[f"{tag}_Kinv1"], z[f"{tag}_Kinv2"]])
    pack = G.GPPack(X, np.stack([y, y], axis=1), Ki, np.stack([z["lam1"], z["lam2"]]), z[f"{tag}_sf"])
    r = G.moment_match(pack, z["u"], z["S"], want_cov=True, bug_compatible=True)
    np.testing.assert_allclose(r["mean"][0].cpu().numpy(), z[f"{tag}_mu"], rtol=1e-9)
    np.testing.assert_allclose(r["var"][0].cpu().numpy(), z[f"{tag}_var"], rtol=1e-6)
    cov = r["cov"][0].cpu().numpy()
    np.testing.assert_allclose(cov[0, 1], z[f"{tag}_cov"], rtol=1e-7)
    np.testing.assert_allclose(np.diag(cov), z[f"{tag}_var"], rtol=1e-6)
    if f"{tag}_np_cov" in z:
        # reference's own rung-2 tolerances (abs 1e-7 mean, 1e-5 var / cov) against its numpy loops
        assert abs(r["mean"][0, 0].item() - z[f"{tag}_np_mu"]) < 1e-7 * max(1.0, abs(z[f"{tag}_np_mu"]))
        assert abs(r["var"][0, 0].item() - z[f"{tag}_np_var"]) < 1e-5 * abs(z[f"{tag}_np_var"])


@pytest.mark.parametrize("k", range(6))
def test_g2_adversarial(G, golden, k):
    z = golden("g2_adversarial.npz")
    p = f"c{k}_"
    sf1, sf2, _ = z[p + "hyp"]
    pack = G.GPPack(z[p + "X"], np.stack([z[p + "y1"], z[p + "y2"]], axis=1),
                    np.stack([z[p + "Kinv1"], z[p + "Kinv2"]]), np.stack([z[p + "lam1"], z[p + "lam2"]]),
                    np.array([sf1, sf2]))
    r = G.moment_match(pack, z[p + "u"], z[p + "S"], want_cov=True, want_grad=True, bug_compatible=True)
    np.testing.assert_allclose(r["mean"][0].cpu().numpy(), z[p + "mu"], rtol=1e-9)
    np.testing.assert_allclose(r["var"][0].cpu().numpy(), z[p + "var"], rtol=1e-6)
    np.testing.assert_allclose(r["cov"][0, 0, 1].item(), z[p + "cov_torch"], rtol=1e-6, atol=1e-9)
    sym = lambda A: 0.5 * (A + A.T)                                 # noqa: E731
    np.testing.assert_allclose(r["dmean_du"][0, 0].cpu().numpy(), z[p + "dm_du"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(r["dmean_dS"][0, 0].cpu().numpy(), sym(z[p + "dm_dS"]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(r["dvar_du"][0, 0].cpu().numpy(), z[p + "dv_du"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r["dvar_dS"][0, 0].cpu().numpy(), sym(z[p + "dv_dS"]), rtol=1e-5, atol=1e-7)
    # unit sigma_f, shared y: consistent form == the reference's numpy double loop
    packu = G.GPPack(z[p + "X"], np.stack([z[p + "y1"], z[p + "y1"]], axis=1),
                     np.stack([z[p + "unit_Kinv1"], z[p + "unit_Kinv2"]]),
                     np.stack([z[p + "lam1"], z[p + "lam2"]]), np.ones(2))
    fix = G.moment_match(packu, z[p + "u"], z[p + "S"], want_cov=True)["cov"][0].cpu().numpy()
    bug = G.moment_match(packu, z[p + "u"], z[p + "S"], want_cov=True, bug_compatible=True)["cov"][0].cpu().numpy()
    np.testing.assert_allclose(fix[0, 1], z[p + "unit_cov_numpy"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(fix[1, 0], fix[0, 1], rtol=0, atol=0)
    np.testing.assert_allclose(bug[0, 1], z[p + "unit_cov_torch"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", ["g3_rollout_c1.npz", "g4_rollout_c2.npz"])
def test_rollout_golden(G, golden, name):
    """Rollout + cost + gradient against the reference's own outputs, per trajectory and batched."""
    z = golden(name)
    pack = _pack_from(G, z)
    B = z["x0"].shape[0]
    for gi, gamma in enumerate(z["gammas"]):
        cost = _cost_from(G, z, float(gamma))
        rb = G.rollout(pack, z["x0"], z["U"], cost)                 # batched
        for b in range(B):
            r1 = G.rollout(pack, z["x0"][b], z["U"][b], cost)       # single trajectory, as the reference
            for r, idx in ((r1, 0), (rb, b)):
                np.testing.assert_allclose(r["means"][idx].cpu().numpy(), z["means"][b], rtol=MEAN_RTOL, atol=1e-9)
                np.testing.assert_allclose(r["vars"][idx].cpu().numpy(), z["vars"][b], rtol=VAR_RTOL, atol=1e-12)
                np.testing.assert_allclose(r["cost"][idx].item(), z["costs"][gi, b], rtol=1e-6)
                np.testing.assert_allclose(r["grad"][idx].cpu().numpy(), z["grads"][gi, b], rtol=1e-4, atol=1e-7)
            # the batched and the single-trajectory paths use different tilings: agree to rounding
            np.testing.assert_allclose(rb["means"][b].cpu().numpy(), r1["means"][0].cpu().numpy(), rtol=1e-9, atol=1e-12)
    # tighter than the north-star bound: what the fp64 path actually achieves
    np.testing.assert_allclose(rb["means"].cpu().numpy(), z["means"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(rb["vars"].cpu().numpy(), z["vars"], rtol=1e-5, atol=1e-12)


def test_rollout_objective_only_matches(G, golden):
    z = golden("g4_rollout_c2.npz")
    pack = _pack_from(G, z)
    cost = _cost_from(G, z, -1.0)
    a = G.rollout(pack, z["x0"], z["U"], cost, want_grad=True)
    b = G.rollout(pack, z["x0"], z["U"], cost, want_grad=False)
    np.testing.assert_allclose(b["cost"].cpu().numpy(), a["cost"].cpu().numpy(), rtol=1e-9)
    np.testing.assert_allclose(b["vars"].cpu().numpy(), a["vars"].cpu().numpy(), rtol=1e-7)   # different kernels (with / without moments)


def test_rollout_reproducible(G, golden):
    z = golden("g3_rollout_c1.npz")
    pack = _pack_from(G, z)
    cost = _cost_from(G, z, -1.0)
    a = G.rollout(pack, z["x0"], z["U"], cost)
    b = G.rollout(pack, z["x0"], z["U"], cost)
    for k in a:
        assert torch.equal(a[k], b[k]), k       # fixed-order reductions: bit-identical


def test_g5_cost_known_answers(G, golden):
    from gaussian_process_mpc_amd.rollout import cost_full
    z = golden("g5_cost.npz")
    c = G.CostParams(1.0, z["a_Q"], z["a_R"], x_ref=z["a_xref"], u_ref=z["a_uref"])
    got = cost_full(c, z["a_x"], z["a_sig"], z["a_u"]).item()
    assert abs(got - z["a_cost_np"]) < 1e-6 and abs(got - z["a_cost_torch"]) < 1e-9
    c = G.CostParams(1.1, z["a_Q"], z["a_R"], R_delta=z["b_Rd"], x_ref=z["a_xref"], u_ref=z["a_uref"], last_u=z["b_last"][:2])
    got = cost_full(c, z["b_x"], z["b_sig"], z["b_u"]).item()
    assert abs(got - z["b_cost_torch"]) < 1e-9
    H = 5
    c = G.CostParams(-1.0, 2 * np.eye(1), np.zeros((1, 1)), R_delta=np.zeros((1, 1)))
    got = cost_full(c, z["c_x"].reshape(H + 1, 1), z["c_sig"].reshape(H + 1, 1, 1), np.zeros((H, 1))).item()
    assert abs(got - z["c_closed"]) < 1e-7


def test_risk_neutral_limit(G, golden):
    """gamma = 0 (analytic limit, an extension: the reference divides by gamma) is the limit of small gamma."""
    z = golden("g3_rollout_c1.npz")
    pack = _pack_from(G, z)
    r0 = G.rollout(pack, z["x0"], z["U"], _cost_from(G, z, 0.0))
    r1 = G.rollout(pack, z["x0"], z["U"], _cost_from(G, z, 1e-7))
    np.testing.assert_allclose(r0["cost"].cpu().numpy(), r1["cost"].cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(r0["grad"].cpu().numpy(), r1["grad"].cpu().numpy(), rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("N,ds,da,H,B", [(96, 2, 1, 4, 3), (200, 3, 2, 5, 5), (333, 4, 1, 3, 9), (130, 1, 1, 6, 2)])
def test_rollout_vs_oracle_seeded(G, N, ds, da, H, B):
    """Seeded problems (ragged N, several dims, batch sizes that do not fill a trajectory group)
    against the CPU oracle, including the gradient (autograd in the oracle)."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(7, N, ds, da, H, B)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    for gamma in (-1.0, 1e-5):
        r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(gamma, pb["Q"], pb["R"]))
        for b in range(B):
            o = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"],
                                         gamma, mode="o2")
            np.testing.assert_allclose(r["means"][b].cpu().numpy(), o["means"], rtol=MEAN_RTOL, atol=1e-9)
            np.testing.assert_allclose(r["vars"][b].cpu().numpy(), o["vars"], rtol=VAR_RTOL, atol=1e-12)
            np.testing.assert_allclose(r["cost"][b].item(), o["cost"], rtol=1e-6)
            np.testing.assert_allclose(r["grad"][b].cpu().numpy(), o["grad"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("N,ds,da,H", [(200, 3, 1, 3), (300, 4, 1, 2), (130, 2, 2, 3)])
def test_large_batch_shape_vs_oracle(G, N, ds, da, H):
    """The large-batch shape of the rollout (256x256 tiles, scalar-broadcast kernel, two trajectories per wave, the
    horizon-step-1 variant) is selected by the amount of work: drive it with a thousand-plus trajectories of a small
    problem and hold a handful of them to the CPU oracle (values and gradient)."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    tiles = {1: 1, 2: 3}[(N + 255) // 256]
    B = 5600 // (tiles * ds) + 77         # ceil(B / 2) * (tiles * ds) >= 2800 workgroups selects the shape; odd batch on purpose
    B += 1 - B % 2
    pb = synth_problem(11, N, ds, da, H, B)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(-1.0, pb["Q"], pb["R"]))
    assert all(torch.isfinite(v).all() for v in r.values())
    for b in (0, 1, B // 2, B - 2, B - 1):
        o = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0,
                                     mode="o2")
        np.testing.assert_allclose(r["means"][b].cpu().numpy(), o["means"], rtol=MEAN_RTOL, atol=1e-9)
        np.testing.assert_allclose(r["vars"][b].cpu().numpy(), o["vars"], rtol=VAR_RTOL, atol=1e-12)
        np.testing.assert_allclose(r["cost"][b].item(), o["cost"], rtol=1e-6)
        np.testing.assert_allclose(r["grad"][b].cpu().numpy(), o["grad"], rtol=1e-4, atol=1e-7)
    # the same trajectories through the small-batch kernels
    r5 = G.rollout(pack, pb["x0"][:5], pb["U"][:5], G.CostParams(-1.0, pb["Q"], pb["R"]))
    np.testing.assert_allclose(r5["cost"].cpu().numpy(), r["cost"][:5].cpu().numpy(), rtol=1e-8)
    np.testing.assert_allclose(r5["grad"].cpu().numpy(), r["grad"][:5].cpu().numpy(), rtol=1e-5, atol=1e-9)


def test_long_horizon_uses_the_streaming_tail(G):
    """H = 100 at ds = 4: the step Jacobians no longer fit in LDS together, so the tail kernel falls back to one
    Jacobian per iteration (prefetched); hold it to the oracle like the short horizons."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    N, ds, da, H, B = 40, 4, 1, 100, 2
    pb = synth_problem(13, N, ds, da, H, B)
    pb["U"] = 0.3 * pb["U"]                                   # keep the long rollout inside the data
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(1e-5, pb["Q"], pb["R"]))
    for b in range(B):
        o = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], 1e-5,
                                     mode="o2")
        np.testing.assert_allclose(r["means"][b].cpu().numpy(), o["means"], rtol=MEAN_RTOL, atol=1e-9)
        np.testing.assert_allclose(r["vars"][b].cpu().numpy(), o["vars"], rtol=VAR_RTOL, atol=1e-12)
        np.testing.assert_allclose(r["cost"][b].item(), o["cost"], rtol=1e-6)
        np.testing.assert_allclose(r["grad"][b].cpu().numpy(), o["grad"], rtol=1e-4, atol=1e-7)


def test_reduced_precision_sweep_modes_document_the_failure(G):
    """BASELINE config 3 asks for an fp64-vs-fp32 tolerance sweep: the fp32 modes (objective only) must run, and must
    be far outside the tolerance that fp64 meets (the variance is a cancelling N^2 sum)."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    N, ds, da, H, B = 512, 3, 1, 2, 2
    pb = synth_problem(15, N, ds, da, H, B, sigma_n=0.1)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    ref = np.stack([torch.stack([torch.diagonal(c) for c in O.forward_propagate(gp, H, pb["x0"][b], torch.as_tensor(pb["U"][b]),
                                                                                mode="o2")[1]]).numpy() for b in range(B)])
    err = {}
    for prec in ("fp64", "fp32acc", "fp32"):
        v = G.rollout(pack, pb["x0"], pb["U"], cost, want_grad=False, precision=prec)["vars"].cpu().numpy()
        assert np.isfinite(v).all()
        err[prec] = np.max(np.abs(v[:, 1:] - ref[:, 1:]) / np.abs(ref[:, 1:]))
    assert err["fp64"] < 1e-7
    assert err["fp32acc"] > 1e3 * err["fp64"] and err["fp32"] > 1e3 * err["fp64"]
    with pytest.raises(ValueError):
        G.rollout(pack, pb["x0"], pb["U"], cost, want_grad=True, precision="fp32")


def test_gradient_finite_difference(G):
    """Analytic adjoint against central differences of the HIP objective itself."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(11, 150, 2, 1, 5, 1)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    U = pb["U"][0]
    g = G.rollout(pack, pb["x0"], U, cost)["grad"][0].cpu().numpy()
    eps = 1e-5
    for j in range(U.shape[0]):
        Up, Um = U.copy(), U.copy()
        Up[j, 0] += eps
        Um[j, 0] -= eps
        fd = (G.rollout(pack, pb["x0"], Up, cost, want_grad=False)["cost"].item() -
              G.rollout(pack, pb["x0"], Um, cost, want_grad=False)["cost"].item()) / (2 * eps)
        assert abs(fd - g[j, 0]) < 1e-5 * max(1.0, abs(fd)), (j, fd, g[j, 0])


@pytest.mark.parametrize("k", range(6))
def test_cross_covariance_pair_kernel_and_jacobians(G, golden, k):
    """Consistent-form cross-covariance through the pair kernel (Gaussian-product weights) equals the direct N^2
    kernel and the reference's numpy loop; its analytic Jacobians equal autograd of the corrected oracle."""
    from oracle import gpmpc_oracle as O
    z = golden("g2_adversarial.npz")
    p = f"c{k}_"
    packu = G.GPPack(z[p + "X"], np.stack([z[p + "y1"], z[p + "y1"]], axis=1),
                     np.stack([z[p + "unit_Kinv1"], z[p + "unit_Kinv2"]]),
                     np.stack([z[p + "lam1"], z[p + "lam2"]]), np.ones(2))
    direct = G.moment_match(packu, z[p + "u"], z[p + "S"], want_cov=True)["cov"][0].cpu().numpy()
    packu.enable_fullcov()
    r = G.moment_match(packu, z[p + "u"], z[p + "S"], want_cov=True, want_grad=True)
    cov = r["cov"][0].cpu().numpy()
    np.testing.assert_allclose(cov[0, 1], z[p + "unit_cov_numpy"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(cov, direct, rtol=1e-7, atol=1e-10)
    assert cov[0, 1] == cov[1, 0]
    # Jacobians of Cov[f1, f2] against autograd through the oracle's corrected covariance_prop
    T = lambda a: torch.as_tensor(a, dtype=torch.float64)          # noqa: E731
    u = T(z[p + "u"]).requires_grad_(True)
    S = T(z[p + "S"]).requires_grad_(True)
    X, y = T(z[p + "X"]), T(z[p + "y1"])
    m1, b1, _ = O.mean_prop(T(z[p + "unit_Kinv1"]), T(z[p + "lam1"]), u, S, X, y)
    m2, b2, _ = O.mean_prop(T(z[p + "unit_Kinv2"]), T(z[p + "lam2"]), u, S, X, y)
    c = O.covariance_prop(T(z[p + "lam1"]), T(z[p + "lam2"]), u, S, X, m1, m2, b1, b2, bug_compatible=False)
    dc_du, dc_dS = torch.autograd.grad(c, (u, S))
    sym = lambda A: 0.5 * (A + A.T)                                 # noqa: E731
    np.testing.assert_allclose(r["dcov_du"][0, 0, 1].cpu().numpy(), dc_du.numpy(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(r["dcov_dS"][0, 0, 1].cpu().numpy(), sym(dc_dS.numpy()), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(r["dcov_du"][0, 0, 0].cpu().numpy(), r["dvar_du"][0, 0].cpu().numpy(), rtol=0, atol=0)


@pytest.mark.parametrize("N,ds,da,H,B,gamma", [(90, 2, 1, 4, 3, -1.0), (150, 3, 1, 3, 2, 1e-5), (130, 3, 2, 3, 2, -1.0),
                                               (70, 1, 1, 4, 2, -1.0), (100, 2, 1, 3, 2, 0.0)])
def test_rollout_fullcov_vs_oracle(G, N, ds, da, H, B, gamma):
    """Full-covariance rollout (config 5 semantics) + cost + analytic gradient against the extension oracle
    (reference single-step functions composed; autograd)."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(5, N, ds, da, H, B)
    pb["Q"] = pb["Q"] + 0.02 * (np.ones((ds, ds)) - np.eye(ds))          # non-diagonal Q couples the covariances
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    r = G.rollout_fullcov(pack, pb["x0"], pb["U"], G.CostParams(gamma, pb["Q"], pb["R"]))
    for b in range(B):
        o = O.objective_and_gradient_fullcov(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], gamma)
        np.testing.assert_allclose(r["means"][b].cpu().numpy(), o["means"], rtol=1e-5, atol=1e-9)
        scale = np.abs(o["covs"]).max(axis=(1, 2), keepdims=True)
        np.testing.assert_allclose(r["covs"][b].cpu().numpy(), o["covs"], rtol=1e-4, atol=1e-6 * scale.max())
        np.testing.assert_allclose(r["cost"][b].item(), o["cost"], rtol=1e-6)
        np.testing.assert_allclose(r["grad"][b].cpu().numpy(), o["grad"], rtol=1e-4, atol=1e-7)
    if ds > 1:
        offdiag = r["covs"][:, 1:].cpu().numpy()[:, :, 0, 1]
        assert np.abs(offdiag).max() > 0                                 # the off-diagonal terms are really there
    # diagonal-covariance rollout differs (it drops them) but shares the first step
    rd = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(gamma, pb["Q"], pb["R"]))
    np.testing.assert_allclose(rd["means"][:, 1].cpu().numpy(), r["means"][:, 1].cpu().numpy(), rtol=1e-9)
    np.testing.assert_allclose(rd["vars"][:, 1].cpu().numpy(),
                               torch.diagonal(r["covs"][:, 1], dim1=1, dim2=2).cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("N,ds,da,H,B", [(1, 2, 1, 2, 1), (2, 1, 1, 3, 2), (63, 2, 1, 2, 2), (64, 3, 1, 1, 3),
                                         (65, 7, 1, 2, 2), (70, 6, 2, 2, 3), (257, 2, 2, 2, 600),
                                         (70, 6, 2, 2, 400), (65, 7, 1, 2, 320)])
def test_rollout_edge_shapes(G, N, ds, da, H, B):
    """Edge cases: a single training point (the reference special-cases 0-D y, uncertainty_prop.py:324), N around the
    64-row padding boundary, the largest supported input dimension D = 8, H = 1, and batches large enough to take the
    256-row tiles / scalar-broadcast kernel at small N (ragged last tile; D = 8 with one and two action dimensions)."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(21, N, ds, da, H, B, sigma_n=0.05)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    assert pack.Np % 64 == 0 and pack.Np >= N
    r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(-1.0, pb["Q"], pb["R"]))
    for b in sorted(set([0, B // 2, B - 1])):
        o = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0)
        np.testing.assert_allclose(r["means"][b].cpu().numpy(), o["means"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(r["vars"][b].cpu().numpy(), o["vars"], rtol=1e-4, atol=1e-12)
        np.testing.assert_allclose(r["cost"][b].item(), o["cost"], rtol=1e-6)
        np.testing.assert_allclose(r["grad"][b].cpu().numpy(), o["grad"], rtol=1e-4, atol=1e-7)


def test_abi_error_codes_on_device(G, golden):
    import ctypes
    from gaussian_process_mpc_amd import _lib
    z = golden("g3_rollout_c1.npz")
    lib = G.lib()
    h = ctypes.c_void_p()
    assert lib.gpmpc_pack_create(ctypes.byref(h), 100, 2, 2) == 0
    x0 = torch.zeros((1, 2), dtype=torch.float64, device="cuda")
    U = torch.zeros((1, 10, 2), dtype=torch.float64, device="cuda")
    out = torch.zeros(64, dtype=torch.float64, device="cuda")
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    c = _cost_from(G, z, -1.0)
    args = (h, 1, 10, _lib.ptr(x0), _lib.ptr(U), ctypes.byref(c.c), 1, None, None, _lib.ptr(out), _lib.ptr(out))
    assert lib.gpmpc_rollout(*args, ctypes.c_void_p(ws.data_ptr()), ws.numel(), None) == -5      # pack not built
    assert lib.gpmpc_pack_destroy(h) == 0
    pack = _pack_from(G, z)
    need = lib.gpmpc_rollout_workspace_bytes(pack.handle, 1, 10, 1)
    assert need > 1024
    args = (pack.handle,) + args[1:]
    assert lib.gpmpc_rollout(*args, ctypes.c_void_p(ws.data_ptr()), 1024, None) == -4              # workspace too small
    assert lib.gpmpc_rollout(*args[:10], None, ctypes.c_void_p(ws.data_ptr()), ws.numel(), None) == -1   # grad wanted, no buffer
    assert lib.gpmpc_rollout_fullcov(pack.handle, 1, 10, _lib.ptr(x0), _lib.ptr(U), ctypes.byref(c.c), 0, _lib.ptr(out),
                                     _lib.ptr(out), _lib.ptr(out), None, ctypes.c_void_p(ws.data_ptr()), ws.numel(), None) == -5
    # round-3 entry points: differentiable propagation, cost derivatives, pack re-use
    jac = torch.zeros(10 * 4 * 6, dtype=torch.float64, device="cuda")
    need_j = lib.gpmpc_rollout_jac_workspace_bytes(pack.handle, 1, 10)
    assert need_j > need
    jargs = (pack.handle, 1, 10, _lib.ptr(x0), _lib.ptr(U), _lib.ptr(out), _lib.ptr(out), _lib.ptr(jac))
    assert lib.gpmpc_rollout_jac(*jargs, ctypes.c_void_p(ws.data_ptr()), need_j - 8, None) == -4     # workspace too small
    assert lib.gpmpc_rollout_jac(*jargs[:7], None, ctypes.c_void_p(ws.data_ptr()), ws.numel(), None) == -1   # no Jacobian buffer
    assert lib.gpmpc_rollout_jac(*jargs, ctypes.c_void_p(ws.data_ptr()), ws.numel(), None) == 0
    assert lib.gpmpc_rollout_vjp(1, 10, 2, 2, _lib.ptr(jac), None, None, _lib.ptr(out), None, None) == 0       # zero seeds are allowed
    assert lib.gpmpc_rollout_vjp(1, 10, 2, 2, None, None, None, _lib.ptr(out), None, None) == -1
    assert lib.gpmpc_cost_grad(1, 10, 2, 2, ctypes.byref(c.c), _lib.ptr(out), _lib.ptr(out), _lib.ptr(out), _lib.ptr(out), _lib.ptr(out),
                               None, None, None) == -1                                                     # one of three derivative outputs
    assert lib.gpmpc_pack_resize(pack.handle, 129) == -1          # padded size would change (100 -> 128, 129 -> 192)
    assert lib.gpmpc_pack_resize(pack.handle, 101) == 0
    assert lib.gpmpc_pack_shared_lambda(pack.handle) == -5        # resized, not rebuilt yet
    assert lib.gpmpc_rollout(*args, ctypes.c_void_p(ws.data_ptr()), ws.numel(), None) == -5
    torch.cuda.synchronize()


def test_nan_passthrough(G, golden):
    """A negative input variance makes the reference produce NaN (sqrt / log of a negative number); the HIP path must
    pass NaN through, not clamp (keep NaN semantics)."""
    z = golden("g3_rollout_c1.npz")
    pack = _pack_from(G, z)
    D = 4
    S = np.diag([-50.0, 1e-3, 1e-3, 1e-3])
    r = G.moment_match(pack, np.zeros(D), S)
    assert torch.isnan(r["var"]).any() or torch.isnan(r["mean"]).any()


@pytest.mark.parametrize("chunks", ["1", "auto", "8"])
def test_row_chunked_head_kernel_vs_cport(G, monkeypatch, chunks):
    """Small batch of a large training set (N = 2049, ds = 5: 765 work items, the many-items reduction): the head kernel split
    over row chunks (GPMPC_HEAD_CHUNKS: off / chosen per call / 8) against the C port.  Guards the bit-consistency the
    scheme depends on: every workgroup of a trajectory must derive identical input variances, or the row-side transform of
    a unit and the column rows written by another chunk's workgroup disagree by 1e-9 and the N^2 sum turns that into 1e-2."""
    from oracle import cport, gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    if chunks == "auto":
        monkeypatch.delenv("GPMPC_HEAD_CHUNKS", raising=False)
    else:
        monkeypatch.setenv("GPMPC_HEAD_CHUNKS", chunks)
    monkeypatch.setenv("GPMPC_FUSED", "0")
    pb = synth_problem(99, 2049, 5, 1, 3, 4)
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])          # reads the overrides
    r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(-1.0, pb["Q"], pb["R"]))
    c = cport.rollout(pb, kinv, -1.0, nthreads=16)
    np.testing.assert_allclose(r["means"].cpu().numpy(), c["means"], rtol=MEAN_RTOL, atol=1e-9)
    np.testing.assert_allclose(r["vars"].cpu().numpy(), c["vars"], rtol=VAR_RTOL)
    np.testing.assert_allclose(r["cost"].cpu().numpy(), c["cost"], rtol=1e-6)
    np.testing.assert_allclose(r["grad"].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7)


def test_concurrent_streams_share_a_pack(G):
    """include/gpmpc.h: calls are re-entrant across streams as long as the workspaces differ.  Two streams drive the same
    pack with different batches, interleaved and overlapping on the device (with the per-kernel timing accumulators
    switched on, which are shared state); every result equals the one the call gives on its own, bit for bit."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd._lib import lib
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(31, 700, 3, 1, 6, 96)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    jobs = [(slice(0, 64), True), (slice(64, 96), True), (slice(10, 13), False), (slice(0, 96), True)]   # several kernel shapes
    x0, U = torch.as_tensor(pb["x0"], device=pack.device), torch.as_tensor(pb["U"], device=pack.device)
    alone = [G.rollout(pack, x0[s], U[s], cost, want_grad=g) for s, g in jobs]
    torch.cuda.synchronize()
    lib().gpmpc_timing_enable(1)
    try:
        streams = [torch.cuda.Stream(device=pack.device) for _ in range(2)]
        got = [[None] * len(jobs) for _ in streams]
        for rep in range(3):
            for j, (s, g) in enumerate(jobs):
                for k, st in enumerate(streams):
                    jj = (j + k) % len(jobs)                      # the two streams run DIFFERENT jobs at the same time
                    with torch.cuda.stream(st):
                        got[k][jj] = G.rollout(pack, x0[jobs[jj][0]], U[jobs[jj][0]], cost, want_grad=jobs[jj][1])
        torch.cuda.synchronize()
    finally:
        lib().gpmpc_timing_enable(0)
        lib().gpmpc_pair_kernel_time(None, None, 1)              # drain and drop the recorded events
    for k in range(2):
        for j in range(len(jobs)):
            for name in alone[j]:
                assert torch.equal(alone[j][name], got[k][j][name]), (k, j, name)


def test_graph_replay_matches_eager(G, golden):
    """GPMPC_USE_GRAPH: the captured launch sequence gives bit-identical results, survives new input values,
    a changed cost (re-capture) and interleaving with eager calls."""
    z = golden("g4_rollout_c2.npz")
    pack = _pack_from(G, z)
    cost = _cost_from(G, z, -1.0)
    for rep in range(3):
        for b in range(2):
            e = G.rollout(pack, z["x0"][b], z["U"][b] * (1 + 0.1 * rep), cost)
            g = G.rollout(pack, z["x0"][b], z["U"][b] * (1 + 0.1 * rep), cost, graph=True)
            for k in e:
                assert torch.equal(e[k], g[k]), (rep, b, k)
    cost2 = _cost_from(G, z, 1e-5)
    e = G.rollout(pack, z["x0"][0], z["U"][0], cost2)
    g = G.rollout(pack, z["x0"][0], z["U"][0], cost2, graph=True)
    assert torch.equal(e["cost"], g["cost"]) and torch.equal(e["grad"], g["grad"])
    # a caller alternating two shapes (objective-only / objective+gradient, as a line search does) replays both: the
    # pack keeps several captured graphs, nothing is re-captured after the first round
    from gaussian_process_mpc_amd._lib import lib
    pack2 = _pack_from(G, z)
    for rep in range(4):
        a = G.rollout(pack2, z["x0"][0], z["U"][0], cost, graph=True, want_traj=False)
        b = G.rollout(pack2, z["x0"][0], z["U"][0], cost, graph=True, want_traj=False, want_grad=False)
        assert abs(a["cost"][0].item() - b["cost"][0].item()) <= 1e-9 * abs(a["cost"][0].item())
        if rep == 0:
            first = lib().gpmpc_pack_graph_captures(pack2.handle)
    assert first == 2 and lib().gpmpc_pack_graph_captures(pack2.handle) == 2


def test_objective_gradient_callback_entry(G, golden):
    """gpmpc_objective_gradient (host in / host out, one captured graph per pack: upload + rollout + download) returns
    exactly the bits of gpmpc_rollout for the same candidate, across new inputs, a new start state, a changed cost
    (re-capture), a changed horizon (re-allocation) and the objective-only form; and the reference's own values (g4)."""
    z = golden("g4_rollout_c2.npz")
    pack = _pack_from(G, z)
    H = z["U"].shape[1]
    for gamma in (-1.0, 1e-5):
        cost = _cost_from(G, z, gamma)
        for rep in range(3):
            for b in range(2):
                U = z["U"][b] * (1 + 0.1 * rep)
                e = G.rollout(pack, z["x0"][b], U, cost, want_traj=False)
                cg = pack.objective_gradient(z["x0"][b], U, cost)
                assert cg.shape == (1 + H * U.shape[1],)
                assert cg[0] == e["cost"][0].item() and np.array_equal(cg[1:], e["grad"][0].cpu().numpy().reshape(-1)), (gamma, rep, b)
    gi = list(z["gammas"]).index(-1.0) if -1.0 in list(z["gammas"]) else 0
    cg = pack.objective_gradient(z["x0"][0], z["U"][0], _cost_from(G, z, float(z["gammas"][gi])))
    np.testing.assert_allclose(cg[0], z["costs"][gi, 0], rtol=1e-6)
    np.testing.assert_allclose(cg[1:].reshape(z["grads"][gi, 0].shape), z["grads"][gi, 0], rtol=1e-4, atol=1e-7)
    cost = _cost_from(G, z, -1.0)
    short = pack.objective_gradient(z["x0"][1], z["U"][1][:5], cost)                   # shorter horizon: re-capture
    e = G.rollout(pack, z["x0"][1], z["U"][1][:5], cost, want_traj=False)
    assert short[0] == e["cost"][0].item() and np.array_equal(short[1:], e["grad"][0].cpu().numpy().reshape(-1))
    only = pack.objective_gradient(z["x0"][1], z["U"][1], cost, want_grad=False)
    assert only.shape == (1,)
    np.testing.assert_allclose(only[0], G.rollout(pack, z["x0"][1], z["U"][1], cost, want_grad=False)["cost"][0].item(), rtol=1e-12)


def test_callback_entry_with_timing_enabled_falls_back_to_uncaptured_launches(G, golden):
    """gpmpc_timing_enable(1) and the solver callback together: per-kernel events cannot be recorded inside a captured
    graph, so the entry enqueues the same work uncaptured -- same bits, no error -- and the fused step kernel's launches are
    accounted under their own timing class (2), not as pair-kernel time."""
    import ctypes
    from gaussian_process_mpc_amd._lib import lib
    z = golden("g4_rollout_c2.npz")
    pack = _pack_from(G, z)
    cost = _cost_from(G, z, -1.0)
    H = z["U"].shape[1]
    ref = pack.objective_gradient(z["x0"][0], z["U"][0], cost)
    ms, nl = ctypes.c_double(), ctypes.c_longlong()
    lib().gpmpc_timing_enable(1)
    try:
        lib().gpmpc_pair_kernel_time(ctypes.byref(ms), ctypes.byref(nl), 1)
        got = pack.objective_gradient(z["x0"][0], z["U"][0], cost)
        assert np.array_equal(got, ref)
        lib().gpmpc_pair_kernel_time_class(2, ctypes.byref(ms), ctypes.byref(nl))
        assert nl.value == H and ms.value > 0.0                      # one fused launch per horizon step
        lib().gpmpc_pair_kernel_time_class(0, ctypes.byref(ms), ctypes.byref(nl))
        assert nl.value == 0
    finally:
        lib().gpmpc_timing_enable(0)
        lib().gpmpc_pair_kernel_time(ctypes.byref(ms), ctypes.byref(nl), 1)
    assert np.array_equal(pack.objective_gradient(z["x0"][0], z["U"][0], cost), ref)      # and captured again afterwards


def test_scalar_broadcast_kernels_match_staged(G, golden, monkeypatch):
    """The scalar-broadcast pair kernels (pair_kernel_sb.h / pair_kernel_sbf.h: expanded exponent, row-grouped moments)
    take over once the grid fills the chip; force both forms on the same large batch and compare them with each other
    and, for a few entries, with the reference's fixtures."""
    z = golden("g2_adversarial.npz")
    p = "c3_"                                                       # D = 4, full S
    sf1, sf2, _ = z[p + "hyp"]
    pack = G.GPPack(z[p + "X"], np.stack([z[p + "y1"], z[p + "y2"]], axis=1),
                    np.stack([z[p + "Kinv1"], z[p + "Kinv2"]]), np.stack([z[p + "lam1"], z[p + "lam2"]]),
                    np.array([sf1, sf2])).enable_fullcov()
    rng = np.random.default_rng(8)
    nq, D = 900, 4
    u = z[p + "u"] + 0.3 * rng.normal(size=(nq, D))
    Aq = rng.normal(size=(nq, D, D))
    S = 0.03 * Aq @ np.swapaxes(Aq, 1, 2) + 0.01 * np.eye(D)
    u[0], S[0] = z[p + "u"], z[p + "S"]
    monkeypatch.setenv("GPMPC_PAIR_SB", "0")
    a = G.moment_match(pack.reload_tuning(), u, S, want_cov=True, want_grad=True)    # the overrides are read per pack, not per call
    monkeypatch.setenv("GPMPC_PAIR_SB", "1")
    b = G.moment_match(pack.reload_tuning(), u, S, want_cov=True, want_grad=True)
    for k in a:
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        np.testing.assert_allclose(y, x, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(x).max()), err_msg=k)
    np.testing.assert_allclose(b["var"][0].cpu().numpy(), z[p + "var"], rtol=1e-6)
    sym = lambda M: 0.5 * (M + M.T)                                 # noqa: E731
    np.testing.assert_allclose(b["dvar_dS"][0, 0].cpu().numpy(), sym(z[p + "dv_dS"]), rtol=1e-5, atol=1e-7)
    # rollouts: diagonal (pair_kernel_sb.h) and full covariance (pair_kernel_sbf.h with NS2 = state_dim)
    from gaussian_process_mpc_amd.synth import synth_problem
    from oracle import gpmpc_oracle as O
    pb = synth_problem(31, 100, 2, 1, 3, 700)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pk = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"] + 0.02 * (1 - np.eye(2)), pb["R"])
    monkeypatch.setenv("GPMPC_PERSIST", "0")                 # (a batch of 700 would otherwise take the whole-horizon kernel either way)
    for fn in (G.rollout, G.rollout_fullcov):
        monkeypatch.setenv("GPMPC_PAIR_SB", "0")
        a = fn(pk.reload_tuning(), pb["x0"], pb["U"], cost)
        monkeypatch.setenv("GPMPC_PAIR_SB", "1")
        b = fn(pk.reload_tuning(), pb["x0"], pb["U"], cost)
        for k in a:
            x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
            np.testing.assert_allclose(y, x, rtol=1e-6, atol=1e-10, err_msg=f"{fn.__name__}:{k}")
        assert not all(torch.equal(a[k], b[k]) for k in a)          # two different kernels really ran
    o = O.objective_and_gradient_fullcov(gp, 3, pb["x0"][5], pb["U"][5], pb["x_ref"], pb["u_ref"], cost_Q := pb["Q"] + 0.02 * (1 - np.eye(2)), pb["R"], -1.0)
    np.testing.assert_allclose(b["covs"][5].cpu().numpy(), o["covs"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(b["grad"][5].cpu().numpy(), o["grad"], rtol=1e-4, atol=1e-7)
