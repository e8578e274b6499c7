"""GPU parity at BASELINE.json's full sizes.

A full reference/oracle run with gradients at these sizes needs 46 GiB (C3) or cannot run at all (C4), so the
full-size checks are (i) the CPU oracle on a bounded slice of the same workload (first horizon step(s), a few
trajectories) and (ii) size-independent properties of the whole H-step batch: agreement of the batched and the
single-trajectory code paths (different tilings), bit reproducibility, invariance under a permutation of the
training set, and a directional finite-difference check of the analytic gradient.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gaussian_process_mpc_amd as g
    g.require_gpu()
    return g


@pytest.fixture(scope="module")
def c3(G):
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    cfg = CONFIGS["C3"]
    pb = synth_problem(3, cfg["N"], cfg["ds"], cfg["da"], cfg["H"], 8)
    torch.set_num_threads(16)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])     # CPU inverse, as the reference
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    return pb, gp, pack


def test_c3_first_steps_against_oracle(G, c3):
    """N=2048, ds=4, da=1: two horizon steps, objective AND gradient, against the O(N^2) oracle with autograd."""
    from oracle import gpmpc_oracle as O
    pb, gp, pack = c3
    H = 2
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    r = G.rollout(pack, pb["x0"][:2], pb["U"][:2, :H], cost)
    for b in range(2):
        o = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b, :H], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"],
                                     -1.0, mode="o2")
        np.testing.assert_allclose(r["means"][b].cpu().numpy(), o["means"], rtol=1e-5, atol=1e-9)     # north star
        np.testing.assert_allclose(r["vars"][b].cpu().numpy(), o["vars"], rtol=1e-4, atol=1e-12)      # north star
        np.testing.assert_allclose(r["cost"][b].item(), o["cost"], rtol=1e-6)
        np.testing.assert_allclose(r["grad"][b].cpu().numpy(), o["grad"], rtol=1e-4, atol=1e-7)


def test_c3_single_step_moment_match_against_faithful_oracle(G, c3):
    """One (u, diagonal S) query at N=2048 against the FAITHFUL oracle (N^3 trace, reference op order)."""
    from oracle import gpmpc_oracle as O
    pb, gp, pack = c3
    T = lambda a: torch.as_tensor(a, dtype=torch.float64)     # noqa: E731
    u = np.concatenate((pb["x0"][0], pb["U"][0, 0]))
    S = np.diag(np.concatenate((np.full(4, 1e-3), [O.ACTION_NOISE_VAR])))
    r = G.moment_match(pack, u, S)
    for a in range(4):
        m, beta, _ = O.mean_prop(gp.Ky_inv[a], gp.lambdas[a], T(u), T(S), gp.X, gp.Y[:, a])
        v = O.variance_prop(gp.Ky_inv[a], gp.lambdas[a], T(u), T(S), gp.X, m, beta, mode="faithful")
        np.testing.assert_allclose(r["mean"][0, a].item(), m.item(), rtol=1e-5)
        np.testing.assert_allclose(r["var"][0, a].item(), v.item(), rtol=1e-4)


def test_c3_full_horizon_properties(G, c3):
    pb, gp, pack = c3
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    rb = G.rollout(pack, pb["x0"], pb["U"], cost)                     # B = 8, H = 20: 256-row tiles
    assert all(torch.isfinite(v).all() for v in rb.values())
    assert float(rb["vars"].min()) > 0
    r1 = G.rollout(pack, pb["x0"][3], pb["U"][3], cost)               # B = 1: one-wave 64x64 tiles
    np.testing.assert_allclose(r1["means"][0].cpu().numpy(), rb["means"][3].cpu().numpy(), rtol=1e-8, atol=1e-11)
    # two different kernels / tilings: the cancelling N^2 sum (sum|terms| / |var| ~ 1e9-1e10) leaves ~1e-6 of noise
    np.testing.assert_allclose(r1["vars"][0].cpu().numpy(), rb["vars"][3].cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(r1["cost"][0].item(), rb["cost"][3].item(), rtol=1e-8)
    np.testing.assert_allclose(r1["grad"][0].cpu().numpy(), rb["grad"][3].cpu().numpy(), rtol=1e-4, atol=1e-8)
    again = G.rollout(pack, pb["x0"], pb["U"], cost)
    for k in rb:
        assert torch.equal(rb[k], again[k]), k                        # fixed-order reductions
    # directional derivative by central differences of the HIP objective
    rng = np.random.default_rng(5)
    d = rng.normal(size=pb["U"][0].shape)
    d /= np.linalg.norm(d)
    # the cost carries ~1e-7 of cancellation noise at N = 2048 (sum_|terms| / |var| ~ 1e9): a wide stencil
    eps = 2e-3
    cp = G.rollout(pack, pb["x0"][0], pb["U"][0] + eps * d, cost, want_grad=False)["cost"].item()
    cm = G.rollout(pack, pb["x0"][0], pb["U"][0] - eps * d, cost, want_grad=False)["cost"].item()
    g0 = float((rb["grad"][0].cpu().numpy() * d).sum())
    assert abs((cp - cm) / (2 * eps) - g0) < 2e-4 * max(1.0, abs(g0)), ((cp - cm) / (2 * eps), g0)


def test_c3_big_tiling_odd_batch(G, c3):
    """The large-batch shape (256x256 tiles, two trajectories per wave) with an odd batch: the padded partner of the
    last trajectory must not leak, every trajectory is independent of its partner, and the shape agrees with the
    small-batch kernels on the same trajectories."""
    from gaussian_process_mpc_amd.synth import synth_problem
    pb, gp, pack = c3
    big = synth_problem(3, pb["N"], pb["ds"], pb["da"], pb["H"], 25)
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    r25 = G.rollout(pack, big["x0"], big["U"], cost)
    r24 = G.rollout(pack, big["x0"][:24], big["U"][:24], cost)
    assert all(torch.isfinite(v).all() for v in r25.values())
    for k in r24:
        assert torch.equal(r25[k][:24], r24[k]), k                     # partner-independent, bit for bit
    r8 = G.rollout(pack, big["x0"][17:25], big["U"][17:25], cost)         # 256x64 tiles, one trajectory per wave
    # different tilings sum the cancelling N^2 terms in different orders: ~1e-6 of noise in the variances, which the
    # next steps' means inherit at ~1e-8
    np.testing.assert_allclose(r8["means"].cpu().numpy(), r25["means"][17:25].cpu().numpy(), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r8["vars"].cpu().numpy(), r25["vars"][17:25].cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(r8["cost"].cpu().numpy(), r25["cost"][17:25].cpu().numpy(), rtol=1e-8)
    np.testing.assert_allclose(r8["grad"].cpu().numpy(), r25["grad"][17:25].cpu().numpy(), rtol=1e-4, atol=1e-8)


def test_c3_training_set_permutation_invariance(G, c3):
    """The result is a sum over pairs: re-ordering the training points only re-orders the tiles."""
    pb, gp, pack = c3
    rng = np.random.default_rng(9)
    perm = rng.permutation(pb["N"])
    Ki = gp.Ky_inv.numpy()[:, perm][:, :, perm]
    pack_p = G.GPPack(pb["X"][perm], pb["Y"][perm], Ki, pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    a = G.rollout(pack, pb["x0"][:2], pb["U"][:2, :5], cost)
    b = G.rollout(pack_p, pb["x0"][:2], pb["U"][:2, :5], cost)
    np.testing.assert_allclose(b["means"].cpu().numpy(), a["means"].cpu().numpy(), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(b["vars"].cpu().numpy(), a["vars"].cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(b["grad"].cpu().numpy(), a["grad"].cpu().numpy(), rtol=1e-4, atol=1e-8)


def test_c3_bench_shape_against_cport(G, c3):
    """The exact bench.py workload (B = 256 trajectories: 256x256 tiles, two trajectories per wave, dispatch interleave 4,
    the horizon-step-1 variant and the 19 full launches): first, middle and last trajectory of the batch, whole horizon,
    held DIRECTLY to the C port -- not through agreement between batch sizes."""
    from oracle import cport
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    pb, gp, pack = c3
    cfg = CONFIGS["C3"]
    big = synth_problem(3, cfg["N"], cfg["ds"], cfg["da"], cfg["H"], cfg["B"])     # bench.py's own batch
    assert np.array_equal(big["X"], pb["X"])
    r = G.rollout(pack, big["x0"], big["U"], G.CostParams(cfg["gamma"], big["Q"], big["R"]))
    assert all(torch.isfinite(v).all() for v in r.values())
    pick = [0, 127, 255]
    c = cport.rollout(big, gp.Ky_inv.numpy(), cfg["gamma"], x0=big["x0"][pick], U=big["U"][pick], nthreads=16)
    np.testing.assert_allclose(r["means"][pick].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9)       # north star
    np.testing.assert_allclose(r["vars"][pick].cpu().numpy(), c["vars"], rtol=1e-4)                    # north star
    np.testing.assert_allclose(r["cost"][pick].cpu().numpy(), c["cost"], rtol=1e-6)
    np.testing.assert_allclose(r["grad"][pick].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7)


def test_c5_full_covariance_at_full_size_against_cport(G, c3):
    """BASELINE config 5 at its size (N = 2048, ds = 4, H = 20, gamma = -1, FULL covariance): two trajectories of a
    B = 64 batch (the pair_kernel_sbf.h shape of bench.py --config C5: 256x256 tiles, 4 variance + 6 cross units) against
    the full-covariance C port over the whole horizon: means 1e-5, covariances 1e-4, cost, and the analytic gradient
    against complex-step directional derivatives."""
    from oracle import cport
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    pb, gp, pack = c3
    cfg = CONFIGS["C5"]
    big = synth_problem(5, cfg["N"], cfg["ds"], cfg["da"], cfg["H"], 64)
    kinv5 = None
    if not np.array_equal(big["X"], pb["X"]):                 # config 5 draws its own training set (seed 1005)
        from oracle import gpmpc_oracle as O
        gp5 = O.GPBundle(big["X"], big["Y"], big["lambdas"], big["sigma_f"], big["sigma_n"])
        kinv5 = gp5.Ky_inv.numpy()
        pack5 = G.GPPack(big["X"], big["Y"], kinv5, big["lambdas"], big["sigma_f"])
    else:
        kinv5, pack5 = gp.Ky_inv.numpy(), pack
    r = G.rollout_fullcov(pack5, big["x0"], big["U"], G.CostParams(cfg["gamma"], big["Q"], big["R"]))
    assert all(torch.isfinite(v).all() for v in r.values())
    pick = [0, 63]
    dirs = np.random.default_rng(55).normal(size=(2, 2, cfg["H"], cfg["da"]))
    c = cport.rollout_fullcov(big, kinv5, cfg["gamma"], x0=big["x0"][pick], U=big["U"][pick], dirs=dirs, nthreads=16)
    np.testing.assert_allclose(r["means"][pick].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9)       # north star
    np.testing.assert_allclose(r["covs"][pick].cpu().numpy(), c["covs"], rtol=1e-4, atol=1e-6 * np.abs(c["covs"]).max())
    np.testing.assert_allclose(torch.diagonal(r["covs"][pick], dim1=2, dim2=3).cpu().numpy(),
                               np.diagonal(c["covs"], axis1=2, axis2=3), rtol=1e-4)                    # north star (variances)
    np.testing.assert_allclose(r["cost"][pick].cpu().numpy(), c["cost"], rtol=1e-6)
    g = r["grad"][pick].cpu().numpy()
    for k in range(2):
        for d in range(2):
            np.testing.assert_allclose(float((g[k] * dirs[k, d]).sum()), c["ddir"][k, d], rtol=1e-4, atol=1e-7)
    assert np.abs(r["covs"][:, 1:, 0, 1].cpu().numpy()).max() > 0
    # the same two trajectories as a batch of TWO and the first one ALONE: the two-launch small-batch form (fullcov.hip::k_fc_head +
    # the pair kernel on 256x64 tiles, two columns per iteration), directly against the same C-port values
    cost5 = G.CostParams(cfg["gamma"], big["Q"], big["R"])
    assert pack5.plan_fullcov(64, cfg["H"])["form"] == "four_launch"
    for sel in ([0, 63], [0]):
        plan = pack5.plan_fullcov(len(sel), cfg["H"])
        assert plan["form"] == "two_launch" and plan["tiling"] == "256x64" and plan["columns_per_iteration"] == 2, plan
        rs = G.rollout_fullcov(pack5, big["x0"][sel], big["U"][sel], cost5)
        k = len(sel)
        np.testing.assert_allclose(rs["means"].cpu().numpy(), c["means"][:k], rtol=1e-5, atol=1e-9, err_msg=str(sel))
        np.testing.assert_allclose(rs["covs"].cpu().numpy(), c["covs"][:k], rtol=1e-4, atol=1e-6 * np.abs(c["covs"]).max(), err_msg=str(sel))
        np.testing.assert_allclose(rs["cost"].cpu().numpy(), c["cost"][:k], rtol=1e-6, err_msg=str(sel))
        gs = rs["grad"].cpu().numpy()
        for kk in range(k):
            for d in range(2):
                np.testing.assert_allclose(float((gs[kk] * dirs[kk, d]).sum()), c["ddir"][kk, d], rtol=1e-4, atol=1e-7, err_msg=str(sel))
        np.testing.assert_allclose(gs, g[:k], rtol=1e-5, atol=1e-8 * np.abs(g).max())          # and the forms agree on the whole gradient
    del pack5
    torch.cuda.empty_cache()


def test_c4_full_horizon_against_cport(G):
    """BASELINE config 4 (N = 4096, ds = 6, da = 1, H = 30): the per-GPU bench batch (B = 128: 256x256 tiles, the D = 7 /
    one-trajectory-per-wave kernel, its horizon-step-1 variant once and the full variant 29 times) with three of its
    trajectories held to the C port over the WHOLE horizon."""
    from oracle import cport, gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    cfg = CONFIGS["C4"]
    pb = synth_problem(4, cfg["N"], cfg["ds"], cfg["da"], cfg["H"], 128)
    torch.set_num_threads(16)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    kinv = gp.Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(cfg["gamma"], pb["Q"], pb["R"]))
    assert all(torch.isfinite(v).all() for v in r.values())
    pick = [0, 64, 127]
    c = cport.rollout(pb, kinv, cfg["gamma"], x0=pb["x0"][pick], U=pb["U"][pick], nthreads=16)
    np.testing.assert_allclose(r["means"][pick].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9)       # north star
    np.testing.assert_allclose(r["vars"][pick].cpu().numpy(), c["vars"], rtol=1e-4)                    # north star
    np.testing.assert_allclose(r["cost"][pick].cpu().numpy(), c["cost"], rtol=1e-6)
    np.testing.assert_allclose(r["grad"][pick].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7)
    # the SAME problem at B = 1 and B = 2 (what one rank's solver callbacks would run: the one-launch-per-step form on 256x64 tiles at
    # D = 7), whole horizon, eager and as a captured graph, held directly to the C port as well
    forms = set()
    for B in (1, 2):
        forms.add(pack.plan(B, cfg["H"])["kernel"])
        for graph in (False, True):
            rs = G.rollout(pack, pb["x0"][pick[:B]], pb["U"][pick[:B]], G.CostParams(cfg["gamma"], pb["Q"], pb["R"]), graph=graph)
            np.testing.assert_allclose(rs["means"].cpu().numpy(), c["means"][:B], rtol=1e-5, atol=1e-9, err_msg=f"B={B}")
            np.testing.assert_allclose(rs["vars"].cpu().numpy(), c["vars"][:B], rtol=1e-4, err_msg=f"B={B}")
            np.testing.assert_allclose(rs["cost"].cpu().numpy(), c["cost"][:B], rtol=1e-6, err_msg=f"B={B}")
            np.testing.assert_allclose(rs["grad"].cpu().numpy(), c["grad"][:B], rtol=1e-4, atol=1e-7, err_msg=f"B={B}")
    assert forms and all(k for k in forms)
    del pack
    torch.cuda.empty_cache()


@pytest.mark.parametrize("N,ds,da,H", [(2500, 6, 1, 4), (3011, 5, 1, 3), (2700, 5, 2, 3), (4096, 6, 2, 2)])
def test_one_trajectory_on_balanced_runs_against_cport(G, N, ds, da, H):
    """Work list 7 (pack.hip::build_worklist_runs; step_fused.h, Q = 256): ONE trajectory of a training set whose 256x64 tiles would take
    several workgroup generations runs on balanced runs of up to 256 columns.  Sizes that are no multiple of 64 / 256 (clipped and ragged
    last runs, waves that enter a run at their diagonal block), D = 6 ... 8, objective + gradient and objective only, eager and as a graph,
    held to the C port; and to the same library on the plain 256x64 list."""
    import os
    from oracle import cport, gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(11, N, ds, da, H, 1)
    torch.set_num_threads(16)
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    plan = pack.plan(1, H)
    assert plan["tiling"] == "256x256" and plan["launches_per_step"] == 1 and ",256," in plan["kernel"].replace(" ", ""), plan
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    c = cport.rollout(pb, kinv, -1.0, x0=pb["x0"], U=pb["U"], nthreads=16)
    out = []
    for graph in (False, True):
        r = G.rollout(pack, pb["x0"], pb["U"], cost, graph=graph)
        np.testing.assert_allclose(r["means"].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9)      # north star
        np.testing.assert_allclose(r["vars"].cpu().numpy(), c["vars"], rtol=1e-4)                   # north star
        np.testing.assert_allclose(r["cost"].cpu().numpy(), c["cost"], rtol=1e-6)
        np.testing.assert_allclose(r["grad"].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7)
        out.append(r)
    for k in ("means", "vars", "cost", "grad"):
        assert torch.equal(out[0][k], out[1][k]), k                                                 # eager == replayed graph, bit for bit
    rf = G.rollout(pack, pb["x0"], pb["U"], cost, want_grad=False)
    np.testing.assert_allclose(rf["vars"].cpu().numpy(), c["vars"], rtol=1e-4)
    np.testing.assert_allclose(rf["cost"].cpu().numpy(), c["cost"], rtol=1e-6)
    os.environ["GPMPC_NO_RUNS"] = "1"                          # (read when a pack is created)
    try:
        plain = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    finally:
        del os.environ["GPMPC_NO_RUNS"]
    assert plain.plan(1, H)["tiling"] == "256x64", plain.plan(1, H)
    rp = G.rollout(plain, pb["x0"], pb["U"], cost)
    # (two summation orders of the same terms: the variances are differences of sums ~1e2 times larger, 1e-8 relative is their rounding)
    np.testing.assert_allclose(out[0]["vars"].cpu().numpy(), rp["vars"].cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(out[0]["grad"].cpu().numpy(), rp["grad"].cpu().numpy(), rtol=1e-5, atol=1e-10)
    del pack, plain
    torch.cuda.empty_cache()


def test_n4096_ds4_single_and_pair_of_trajectories_against_cport(G):
    """N = 4096 with FOUR state dimensions (D = 5 instances: the one-launch-per-step form was extended to N <= ~4300 for them),
    B = 1 and B = 2, whole horizon H = 20, directly against the C port."""
    from oracle import cport, gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(7, 4096, 4, 1, 20, 2)
    torch.set_num_threads(16)
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    c = cport.rollout(pb, kinv, -1.0, x0=pb["x0"], U=pb["U"], nthreads=16)
    for B in (1, 2):
        r = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost)
        np.testing.assert_allclose(r["means"].cpu().numpy(), c["means"][:B], rtol=1e-5, atol=1e-9, err_msg=f"B={B}")      # north star
        np.testing.assert_allclose(r["vars"].cpu().numpy(), c["vars"][:B], rtol=1e-4, err_msg=f"B={B}")                   # north star
        np.testing.assert_allclose(r["cost"].cpu().numpy(), c["cost"][:B], rtol=1e-6, err_msg=f"B={B}")
        np.testing.assert_allclose(r["grad"].cpu().numpy(), c["grad"][:B], rtol=1e-4, atol=1e-7, err_msg=f"B={B}")
    del pack
    torch.cuda.empty_cache()


def test_c4_first_step_against_oracle(G):
    """N=4096, ds=6, da=1 (config 4): first horizon step of two trajectories against the O(N^2) oracle."""
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    cfg = CONFIGS["C4"]
    pb = synth_problem(4, cfg["N"], cfg["ds"], cfg["da"], 3, 2)
    torch.set_num_threads(16)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    r = G.rollout(pack, pb["x0"], pb["U"][:, :1], G.CostParams(-1.0, pb["Q"], pb["R"]))
    for b in range(2):
        o = O.objective_and_gradient(gp, 1, pb["x0"][b], pb["U"][b, :1], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"],
                                     -1.0, mode="o2")
        np.testing.assert_allclose(r["means"][b].cpu().numpy(), o["means"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(r["vars"][b].cpu().numpy(), o["vars"], rtol=1e-4, atol=1e-12)
        np.testing.assert_allclose(r["grad"][b].cpu().numpy(), o["grad"], rtol=1e-4, atol=1e-7)
    del pack
    torch.cuda.empty_cache()


def test_c3_full_horizon_against_cport(G, c3):
    """The WHOLE C3 rollout (N=2048, H=20) with cost and gradient for two trajectories against the plain-C / OpenMP CPU
    port (direct exponent, libm exp, analytic adjoint; pinned to the torch oracle and to the reference's fixtures by
    tests/test_oracle_golden.py) -- the full-size, full-horizon parity the torch oracle cannot afford (46 GiB)."""
    from oracle import cport
    pb, gp, pack = c3
    r = G.rollout(pack, pb["x0"][:2], pb["U"][:2], G.CostParams(-1.0, pb["Q"], pb["R"]))
    c = cport.rollout(pb, gp.Ky_inv.numpy(), -1.0, x0=pb["x0"][:2], U=pb["U"][:2], nthreads=16)
    np.testing.assert_allclose(r["means"].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9)       # north star
    np.testing.assert_allclose(r["vars"].cpu().numpy(), c["vars"], rtol=1e-4)                    # north star
    np.testing.assert_allclose(r["cost"].cpu().numpy(), c["cost"], rtol=1e-6)
    np.testing.assert_allclose(r["grad"].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7)


def test_c3_small_and_mid_batches_directly_against_cport(G, c3):
    """N = 2048, ds = 4, H = 20 at B = 1, 2, 4, 8, 16 and 24 -- the narrow-tile and the 256x64 one-launch forms, their two-branch split,
    head + pair kernel on 256x64 and the 256x128 two-trajectory tiling -- EVERY trajectory of every batch held directly to the C port
    over the whole horizon (B = 1 used to be tied to the B = 8 batch only), eager and graph replay."""
    from oracle import cport
    from gaussian_process_mpc_amd.synth import synth_problem
    pb, gp, pack = c3
    big = synth_problem(3, pb["N"], pb["ds"], pb["da"], pb["H"], 24)
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    c = cport.rollout(big, gp.Ky_inv.numpy(), -1.0, x0=big["x0"], U=big["U"], nthreads=16)
    kernels = set()
    for B in (1, 2, 4, 8, 16, 24):
        pl = pack.plan(B, pb["H"])
        kernels.add((pl["kernel"], pl["tiling"]))
        for graph in ((False, True) if B <= 8 else (False,)):
            r = G.rollout(pack, big["x0"][:B], big["U"][:B], cost, graph=graph)
            # north star 1e-5 relative; a state mean that passes through ~1e-4 carries the ~6e-9 absolute noise every mean of this problem
            # has (the variances' 1e-6 summation-order noise, propagated): absolute floor 1e-8 of the O(1) state scale
            np.testing.assert_allclose(r["means"].cpu().numpy(), c["means"][:B], rtol=1e-5, atol=1e-8, err_msg=f"B={B} {pl}")
            np.testing.assert_allclose(r["vars"].cpu().numpy(), c["vars"][:B], rtol=1e-4, err_msg=f"B={B} {pl}")                # north star
            np.testing.assert_allclose(r["cost"].cpu().numpy(), c["cost"][:B], rtol=1e-6, err_msg=f"B={B} {pl}")
            np.testing.assert_allclose(r["grad"].cpu().numpy(), c["grad"][:B], rtol=1e-4, atol=1e-7, err_msg=f"B={B} {pl}")
    assert len(kernels) >= 3, kernels          # the batch sizes really reach different kernel shapes


def test_beyond_baseline_sizes_n8192_against_cport(G):
    """Twice the largest BASELINE training set (N = 8192, ds = 2, da = 1; M is 1 GiB, 528 tiles of 256x256 per GP): the
    256x64 shape (B = 3) and the 256x256 / two-trajectories-per-wave shape (B = 6) against the C port, values and gradient.
    (The C port reduces its row sums in a fixed order in extended precision: with thread-local running sums of the 33 M
    cancelling terms it was the CHECKER that drifted, by up to 1.5e-3 of the variance depending on the thread count.)"""
    from oracle import cport, gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(21, 8192, 2, 1, 2, 6)
    torch.set_num_threads(16)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    kinv = gp.Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    c = cport.rollout(pb, kinv, -1.0, x0=pb["x0"][:2], U=pb["U"][:2], nthreads=16)
    for B in (3, 6):
        r = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost)
        assert all(torch.isfinite(v).all() for v in r.values())
        np.testing.assert_allclose(r["means"][:2].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9, err_msg=f"B={B}")
        np.testing.assert_allclose(r["vars"][:2].cpu().numpy(), c["vars"], rtol=1e-4, err_msg=f"B={B}")
        np.testing.assert_allclose(r["cost"][:2].cpu().numpy(), c["cost"], rtol=1e-6, err_msg=f"B={B}")
        np.testing.assert_allclose(r["grad"][:2].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7, err_msg=f"B={B}")
    del pack
    torch.cuda.empty_cache()


@pytest.mark.parametrize("cid", ["C1", "C2"])
def test_small_configs_exact_workload_against_cport(G, cid):
    """BASELINE configs 1 and 2 as bench.py runs them -- B = 1, the whole horizon, their own gamma, through the one-launch-
    per-step path, as a captured graph AND through the solver-callback entry gpmpc_objective_gradient -- against the C port
    (config 2 was only held to the reference at N = 128 through g4 before)."""
    from oracle import cport, gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    cfg = CONFIGS[cid]
    pb = synth_problem(int(cid[1]), cfg["N"], cfg["ds"], cfg["da"], cfg["H"], 1)
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(cfg["gamma"], pb["Q"], pb["R"])
    c = cport.rollout(pb, kinv, cfg["gamma"], nthreads=8)
    for graph in (False, True):
        r = G.rollout(pack, pb["x0"], pb["U"], cost, graph=graph)
        np.testing.assert_allclose(r["means"].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9)       # north star
        np.testing.assert_allclose(r["vars"].cpu().numpy(), c["vars"], rtol=1e-4)                    # north star
        np.testing.assert_allclose(r["cost"].cpu().numpy(), c["cost"], rtol=1e-6)
        np.testing.assert_allclose(r["grad"].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7)
    cg = pack.objective_gradient(pb["x0"][0], pb["U"][0], cost)
    assert cg[0] == r["cost"][0].item() and np.array_equal(cg[1:], r["grad"][0].cpu().numpy().reshape(-1))


def test_g11_reference_scalars_at_n2048(G, c3, golden):
    """One hop from the reference at full size: gpmpc_moment_match on the C3 pack against mean / variance values the
    REFERENCE produced at N = 2048 (tests/golden/g11_fullsize_pin.npz; inputs re-derived from the seed, inverse rebuilt
    on the CPU as src/gpr.py:171)."""
    pb, gp, pack = c3
    z = golden("g11_fullsize_pin.npz")
    r = G.moment_match(pack, z["u"], z["S"])
    np.testing.assert_allclose(r["mean"].cpu().numpy(), z["mean"], rtol=1e-7)          # north star 1e-5
    np.testing.assert_allclose(r["var"].cpu().numpy(), z["var"], rtol=1e-5)            # north star 1e-4


def test_g12_reference_objective_and_gradient_at_n2048(G, c3, golden):
    """The WHOLE path one hop from the reference at full size: objective, gradient and trajectory of two plans that the
    REFERENCE evaluated at N = 2048, ds = 4 (H = 2; tests/golden/g12_fullsize_rollout.npz, inputs re-derived from the seed),
    through gpmpc_rollout, through the solver-callback entry, and through the autograd boundary."""
    pb, gp, pack = c3
    z = golden("g12_fullsize_rollout.npz")
    H = int(z["dims"][3])
    cost = G.CostParams(float(z["gamma"][0]), pb["Q"], pb["R"])
    tr = [int(b) for b in z["traj"]]
    r = G.rollout(pack, pb["x0"][tr], pb["U"][tr, :H], cost)
    np.testing.assert_allclose(r["means"].cpu().numpy(), z["means"], rtol=1e-7, atol=1e-10)        # north star 1e-5
    np.testing.assert_allclose(r["vars"].cpu().numpy(), z["vars"], rtol=1e-5)                       # north star 1e-4
    np.testing.assert_allclose(r["cost"].cpu().numpy(), z["cost"], rtol=1e-7)
    np.testing.assert_allclose(r["grad"].cpu().numpy(), z["grad"], rtol=1e-4, atol=1e-8)
    cg = pack.objective_gradient(pb["x0"][tr[0]], pb["U"][tr[0], :H], cost)
    np.testing.assert_allclose(cg[0], z["cost"][0], rtol=1e-7)
    np.testing.assert_allclose(cg[1:].reshape(H, -1), z["grad"][0], rtol=1e-4, atol=1e-8)
    from gaussian_process_mpc_amd.autograd import CostFunction, RolloutFunction
    U = torch.tensor(pb["U"][tr, :H], device=pack.device, requires_grad=True)
    m, v = RolloutFunction.apply(torch.tensor(pb["x0"][tr], device=pack.device), U, pack)
    CostFunction.apply(m, torch.diag_embed(v), U, cost).sum().backward()
    np.testing.assert_allclose(U.grad.cpu().numpy(), z["grad"], rtol=1e-4, atol=1e-8)
