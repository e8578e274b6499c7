"""Every pair-kernel template instance the dispatchers can select for D <= 7 is held to a CPU checker.

The rollout picks its kernel by the amount of work (csrc/step.hip::plan_rollout, csrc/moment.hip::plan_mom):

    whole horizon in one launch (traj_persist.h)          B >= ~0.7 trajectories per CU, N <= 512: one workgroup per trajectory
    one launch per step, 64-row tiles (step_fused.h)      small batches (the staged pair_kernel.h with GPMPC_FUSED=0 / full S)
    one launch per step, 256x64 / 32 / 16 tiles           from ~150 (N >= 512; else ~400) to ~4700 (7000: <= 200 tiles per trajectory)
      (step_fused.h Q = 0 / 32 / 16)                      tile workgroups: scalar-broadcast column loop inside the fused step kernel
    scalar broadcast, 256x64 tiles   (pair_kernel_sb.h)   beyond, one trajectory per wave
    scalar broadcast, 256x128 tiles  (pair_kernel_sb.h)   >= 1600 workgroups of two trajectories per wave, N > 512
    scalar broadcast, 256x256 tiles  (pair_kernel_sb.h)   >= 2800 workgroups of two trajectories per wave (D <= 5), >= 1500 of one
    scalar broadcast, full S         (pair_kernel_sbf.h)  full-covariance rollout / moment matching on large batches

and each is instantiated per (D, state_dim, GRAD, FIRST).  The cases below drive every (state_dim, action_dim) with
D <= 7 and action_dim in {1, 2} through all shapes by the batch size and compare sampled trajectories with the plain-C
ports under oracle/cport (pinned to the torch oracle and to the reference's fixtures by tests/test_oracle_golden.py):
means 1e-5, variances / covariances 1e-4 (BASELINE north star), cost 1e-6, gradient 1e-4.
Reference: src/dynamics.py:126-191, src/tools/uncertainty_prop.py:296-465, src/mpc.py:156-255.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIMS = [(1, 1), (1, 2), (2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 2), (5, 1), (5, 2), (6, 1), (6, 2), (7, 1)]      # up to D = 8 = GPMPC_MAX_D


@pytest.fixture(scope="module")
def G():
    import gaussian_process_mpc_amd as g
    g.require_gpu()
    return g


def _problem(seed, N, ds, da, H, B):
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(seed, N, ds, da, H, B)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    return pb, gp.Ky_inv.numpy()


@pytest.mark.parametrize("ds,da", DIMS)
def test_diag_rollout_every_shape_vs_cport(G, ds, da):
    """Diagonal-covariance rollout: staged / 256x64 / 256x256 kernels (GRAD and objective-only, horizon step 1 and
    later steps) at every input dimension, against the C port's values and analytic adjoint."""
    from oracle import cport
    N, H = 150, 3                                   # Np = 192: 1 row tile, 3 column chunks of the 256x64 work list
    b_big = 5600 // ds + 3                          # ceil(B / 2) * ds >= 2800 workgroups (D <= 5; B * ds >= 1500 above) -> 256x256 tiles
    b_big += 1 - b_big % 2                          # odd on purpose (the last wave of the two-trajectory shape is half empty)
    b_mid = 2048 // (3 * ds) + 2                    # B * 3 ds >= 1700 -> 256x64 tiles
    pb, kinv = _problem(40 + 8 * ds + da, N, ds, da, H, b_big)
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    import os
    narrow = {"GPMPC_FUSED_SB": "1", "GPMPC_PAIR_SB": "1"}
    for B, env in ((3, None), (b_mid, None), (b_mid, {"GPMPC_FUSED_SB": "1"}), (5, dict(narrow, GPMPC_TILING="5")),
                   (4, dict(narrow, GPMPC_TILING="6")), (7, {"GPMPC_PERSIST": "16"}), (6, {"GPMPC_PERSIST": "8"}), (b_big, None)):
        # b_mid runs twice: as head kernel + pair_kernel_sb.h on the 256x64 tiles (the plan for a training set of less than one
        # 256-row tile, and for more than ~4700 tile workgroups) and, forced, as one launch per step on the same tiles
        # (step_fused.h, Q = 0: the plan from ~400 to ~4700 tile workgroups of a larger training set); the same form on 256x32 and
        # 256x16 tiles (Q = 32 / 16: the plan for a handful of trajectories of a large training set) is forced on small batches;
        # GPMPC_PERSIST forces the whole-horizon kernel (traj_persist.h: one workgroup of 16 / 8 waves per trajectory, the plan for
        # batches of about one trajectory per CU and more of a training set of up to 512 points)
        try:
            os.environ.update(env or {})
            pack.reload_tuning()
            if env and "GPMPC_PERSIST" in env:
                assert pack.plan(B, H)["form"] == "persist"
            r = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost)
            f = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost, want_grad=False)       # the GRAD = false instances
        finally:
            for k in env or {}:
                os.environ.pop(k, None)
            pack.reload_tuning()
        assert all(torch.isfinite(v).all() for v in r.values())
        if env and "GPMPC_PERSIST" in env:
            # the SAME arithmetic in another summation order: against the step-per-launch plan the means agree to ~1e-12 and the variances
            # to ~1e-9 at this size -- far inside the north-star tolerances, and tight enough to see a lost mantissa word (a spill-path
            # miscompile of the heaviest instance, D = 8 with 94 spilled VGPRs, returned means 1e-6 off: csrc/traj_persist.h)
            try:
                os.environ["GPMPC_PERSIST"] = "0"
                pack.reload_tuning()
                r_step = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost)
            finally:
                os.environ.pop("GPMPC_PERSIST", None)
                pack.reload_tuning()
            np.testing.assert_allclose(r["means"].cpu().numpy(), r_step["means"].cpu().numpy(), rtol=1e-9, atol=1e-12, err_msg=f"persist B={B}")
            np.testing.assert_allclose(r["vars"].cpu().numpy(), r_step["vars"].cpu().numpy(), rtol=1e-6, atol=1e-14, err_msg=f"persist B={B}")
            np.testing.assert_allclose(r["grad"].cpu().numpy(), r_step["grad"].cpu().numpy(), rtol=1e-5, atol=1e-10, err_msg=f"persist B={B}")
        pick = sorted({0, 1, B // 2, B - 1})
        c = cport.rollout(pb, kinv, -1.0, x0=pb["x0"][pick], U=pb["U"][pick], nthreads=8)
        np.testing.assert_allclose(r["means"][pick].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9, err_msg=f"B={B}")
        np.testing.assert_allclose(r["vars"][pick].cpu().numpy(), c["vars"], rtol=1e-4, atol=1e-12, err_msg=f"B={B}")
        np.testing.assert_allclose(r["cost"][pick].cpu().numpy(), c["cost"], rtol=1e-6, err_msg=f"B={B}")
        np.testing.assert_allclose(r["grad"][pick].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7, err_msg=f"B={B}")
        np.testing.assert_allclose(f["cost"].cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-9)
        np.testing.assert_allclose(f["vars"].cpu().numpy(), r["vars"].cpu().numpy(), rtol=1e-7)


@pytest.mark.parametrize("ds,da,N", [(2, 1, 110), (2, 2, 200), (3, 1, 110), (3, 2, 130), (4, 1, 110), (4, 2, 300), (4, 1, 449), (5, 1, 110)])
def test_fullcov_rollout_with_one_lambda_vs_cport(G, ds, da, N, monkeypatch):
    """Full-covariance rollout of a pack whose GPs share their length-scales (the reference's experiments): from two to four state
    dimensions the cross units of the two-launch form run through pair_kernel_sbfx.h (one pass over the pairs for ALL of them, no weight
    stream, beta_a,i beta_b,j applied to shared sums), the variance units through pair_kernel_sbf.h.  Held to the C port (which knows
    nothing about shared length-scales) and to the same library with the sharing switched off; ds = 5 keeps the per-unit kernels."""
    from oracle import cport, gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    H = 3
    bmax = 700 if N == 449 else 5
    pb = synth_problem(80 + 8 * ds + da, N, ds, da, H, bmax, shared_lambda=True)
    pb["Q"] = pb["Q"] + 0.02 * (np.ones((ds, ds)) - np.eye(ds))
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    rng = np.random.default_rng(ds * 10 + da)
    pack.enable_fullcov()                            # (rollout_fullcov does it on first use; the plan is asked for before that here)
    # the plan takes the shared form from B Np^2 pairs >= 3.5e7 (fullcov.hip::plan_fc2): forced on for the small shapes of this test
    # (GPMPC_FC_SHARED=1), and taken by the default plan at N = 449 (Np = 512), ds = 4, B = 24
    forced = {"GPMPC_FC_SHARED": "1"}
    cases = [(1, forced), (3, forced), (5, dict(forced, GPMPC_FC_TILING="4")), (2, dict(forced, GPMPC_FC_TILING="0"))]
    if N == 449:                                     # (B = 24, 36 tiles of 64 columns each: 864 -> 64-column tiles by the default plan; the forced small batches: 16-column tiles)
        cases += [(24, None), (700, None)]            # (B = 700: 25 200 tiles of 64 columns -> up to 256 columns per wave, listed by the pack)
        assert pack.plan_fullcov(3, H)["shared_cross_units"] == 0
    for B, env in cases:
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        pack.reload_tuning()
        plan = pack.plan_fullcov(B, H)
        assert plan["form"] == "two_launch" and plan["shared_cross_units"] == (1 if ds <= 4 else 0), plan
        if ds <= 4:
            assert plan["cross_tile_columns"] == {24: 64, 700: 256}.get(B, 16), plan
        r = G.rollout_fullcov(pack, pb["x0"][:B], pb["U"][:B], cost)
        f = G.rollout_fullcov(pack, pb["x0"][:B], pb["U"][:B], cost, want_grad=False)
        for k in (env or {}):
            monkeypatch.delenv(k)
        pack.reload_tuning()
        assert all(torch.isfinite(v).all() for v in r.values())
        pick = [0, B - 1]
        dirs = rng.normal(size=(2, 2, H, da))
        c = cport.rollout_fullcov(pb, kinv, -1.0, x0=pb["x0"][pick], U=pb["U"][pick], dirs=dirs, nthreads=8)
        np.testing.assert_allclose(r["means"][pick].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9, err_msg=f"B={B}")
        np.testing.assert_allclose(r["covs"][pick].cpu().numpy(), c["covs"], rtol=1e-4, atol=1e-6 * np.abs(c["covs"]).max(), err_msg=f"B={B}")
        np.testing.assert_allclose(r["cost"][pick].cpu().numpy(), c["cost"], rtol=1e-6, err_msg=f"B={B}")
        g = r["grad"][pick].cpu().numpy()
        for k in range(2):
            for d in range(2):
                np.testing.assert_allclose(float((g[k] * dirs[k, d]).sum()), c["ddir"][k, d], rtol=1e-4, atol=1e-7, err_msg=f"B={B} {k} {d}")
        np.testing.assert_allclose(f["cost"].cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-9)
        np.testing.assert_allclose(f["covs"].cpu().numpy(), r["covs"].cpu().numpy(), rtol=1e-9, atol=1e-14)
    # the same pack with the sharing off and forced on: the forms agree far inside the tolerance against the C port
    monkeypatch.setenv("GPMPC_FC_SHARED", "0")
    pack.reload_tuning()
    assert pack.plan_fullcov(3, H)["shared_cross_units"] == 0
    u = G.rollout_fullcov(pack, pb["x0"][:3], pb["U"][:3], cost)
    monkeypatch.setenv("GPMPC_FC_SHARED", "1")
    pack.reload_tuning()
    r = G.rollout_fullcov(pack, pb["x0"][:3], pb["U"][:3], cost)
    monkeypatch.delenv("GPMPC_FC_SHARED")
    pack.reload_tuning()
    # (two cuts of cancelling sums: half the tolerance the covariances are held to against the C port above)
    for key in ("means", "covs", "cost", "grad"):
        np.testing.assert_allclose(r[key].cpu().numpy(), u[key].cpu().numpy(), rtol=5e-5, atol=1e-7 * float(u[key].abs().max()), err_msg=key)


@pytest.mark.parametrize("ds,da", [(2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 2), (5, 1), (5, 2), (6, 1), (6, 2)])
def test_fullcov_rollout_every_shape_vs_cport(G, ds, da, monkeypatch):
    """Full-covariance rollout (config 5 semantics): the staged kernel (small batch) and pair_kernel_sbf.h (large batch)
    at every input dimension up to 7, against the C port: means, covariances, cost, and the analytic gradient held to
    complex-step directional derivatives of the C port."""
    from oracle import cport
    N, H = 110, 3
    units = ds + ds * (ds - 1) // 2
    b_big = 2 * (1536 // units) + 5                 # ceil(B / 2) * units >= 1536 (Np = 192: not a multiple of 256) -> pair_kernel_sbf.h
    pb, kinv = _problem(60 + 8 * ds + da, N, ds, da, H, b_big)
    pb["Q"] = pb["Q"] + 0.02 * (np.ones((ds, ds)) - np.eye(ds))           # couples the off-diagonal covariances into the cost
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    rng = np.random.default_rng(ds * 10 + da)
    # (B, overrides): the default plan at a small and a large batch (two launches per step on 256x64 tiles | four on 256x256 tiles +
    # pair_kernel_sbf.h), the two-launch form forced on each of its tilings, and the four-launch form forced at the small batch
    cases = [(3, None), (b_big, None)]
    if da <= 2:
        cases += [(3, {"GPMPC_FC_FORM": "1", "GPMPC_FC_TILING": "4"}), (2, {"GPMPC_FC_FORM": "1", "GPMPC_FC_TILING": "0"}),
                  (1, {"GPMPC_FC_FORM": "1", "GPMPC_FC_TILING": "2"}), (3, {"GPMPC_FC_FORM": "0"}), (b_big, {"GPMPC_FC_FORM": "0"})]
    ref3 = None
    for B, env in cases:
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        pack.reload_tuning()
        r = G.rollout_fullcov(pack, pb["x0"][:B], pb["U"][:B], cost)
        for k in (env or {}):
            monkeypatch.delenv(k)
        pack.reload_tuning()
        assert all(torch.isfinite(v).all() for v in r.values())
        if B == 3:                                  # the forms agree with each other far inside the tolerance against the C port
            if ref3 is None:
                ref3 = r
            else:
                for key in ("means", "covs", "cost", "grad"):
                    np.testing.assert_allclose(r[key].cpu().numpy(), ref3[key].cpu().numpy(), rtol=1e-5,          # (the covariances are
                                               atol=1e-8 * float(ref3[key].abs().max()), err_msg=f"{key} {env}")   # cancelling sums, cut differently)
        pick = [0, B - 1]
        dirs = rng.normal(size=(2, 2, H, da))
        c = cport.rollout_fullcov(pb, kinv, -1.0, x0=pb["x0"][pick], U=pb["U"][pick], dirs=dirs, nthreads=8)
        np.testing.assert_allclose(r["means"][pick].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9, err_msg=f"B={B}")
        np.testing.assert_allclose(r["covs"][pick].cpu().numpy(), c["covs"], rtol=1e-4, atol=1e-6 * np.abs(c["covs"]).max(),
                                   err_msg=f"B={B}")
        np.testing.assert_allclose(r["cost"][pick].cpu().numpy(), c["cost"], rtol=1e-6, err_msg=f"B={B}")
        g = r["grad"][pick].cpu().numpy()
        for k in range(2):
            for d in range(2):
                np.testing.assert_allclose(float((g[k] * dirs[k, d]).sum()), c["ddir"][k, d], rtol=1e-4, atol=1e-7,
                                           err_msg=f"B={B} trajectory {pick[k]} direction {d}")
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        pack.reload_tuning()
        f = G.rollout_fullcov(pack, pb["x0"][:B], pb["U"][:B], cost, want_grad=False)
        for k in (env or {}):
            monkeypatch.delenv(k)
        pack.reload_tuning()
        np.testing.assert_allclose(f["cost"].cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-9)


@pytest.mark.parametrize("D,ds", [(6, 2), (7, 2), (6, 3)])
def test_moment_match_full_S_large_D_vs_cport(G, D, ds):
    """gpmpc_moment_match at D = 6, 7 with a FULL input covariance (the pair kernels' full-moment instances that the
    rollouts of the cases above do not reach: NS2 = D), small and large query batches, against the C port's single step."""
    from oracle import cport
    da = D - ds
    pb, kinv = _problem(90 + D + ds, 90, ds, da, 1, 1)
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"]).enable_fullcov()
    rng = np.random.default_rng(D)
    units = ds + ds * (ds - 1) // 2
    for nq in (2, 2 * (1024 // units) + 3):
        u = 0.5 * rng.normal(size=(nq, D))
        A = rng.normal(size=(nq, D, D))
        S = 0.02 * A @ np.swapaxes(A, 1, 2) + 0.01 * np.eye(D)
        r = G.moment_match(pack, u, S, want_cov=True, want_grad=True)
        assert all(torch.isfinite(v).all() for v in r.values())
        for q in (0, nq - 1):
            m, c = cport.moment_match_fullcov(pb["X"], kinv, pb["Y"], pb["lambdas"], pb["sigma_f"], u[q], S[q], nthreads=4)
            np.testing.assert_allclose(r["mean"][q].cpu().numpy(), m, rtol=1e-5, atol=1e-9)
            np.testing.assert_allclose(r["cov"][q].cpu().numpy(), c, rtol=1e-4, atol=1e-6 * np.abs(c).max())
            np.testing.assert_allclose(r["var"][q].cpu().numpy(), np.diag(c), rtol=1e-4)
        # Jacobians: central differences of the HIP values themselves along one random direction of (u, S)
        q = nq - 1
        du = rng.normal(size=D)
        dS = rng.normal(size=(D, D))
        dS = 0.5 * (dS + dS.T)
        eps = 1e-4
        rp = G.moment_match(pack, u[q] + eps * du, S[q] + eps * dS)
        rm = G.moment_match(pack, u[q] - eps * du, S[q] - eps * dS)
        for key, ju, jS in (("mean", "dmean_du", "dmean_dS"), ("var", "dvar_du", "dvar_dS")):
            fd = (rp[key][0] - rm[key][0]).cpu().numpy() / (2 * eps)
            an = r[ju][q].cpu().numpy() @ du + np.einsum("akl,kl->a", r[jS][q].cpu().numpy(), dS)
            np.testing.assert_allclose(an, fd, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(fd).max()), err_msg=key)


@pytest.mark.parametrize("ds,da", [(1, 1), (2, 2), (4, 1), (6, 1), (5, 2)])
def test_two_kernel_small_batch_form_vs_cport(G, ds, da, monkeypatch):
    """GPMPC_FUSED=0 keeps the round-1 form for small batches (head kernel + staged pair_kernel.h per step; also what a
    64-row work list with more than 4096 items falls back to): its diagonal-S gradient instances (one and two trajectories
    per workgroup) against the C port, and against the default one-launch-per-step path on the same inputs."""
    from oracle import cport
    N, H = 150, 3
    pb, kinv = _problem(120 + 8 * ds + da, N, ds, da, H, 5)
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    fused = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    monkeypatch.setenv("GPMPC_FUSED", "0")
    two = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])            # tuning is read at pack creation
    monkeypatch.delenv("GPMPC_FUSED")
    c = cport.rollout(pb, kinv, -1.0, nthreads=8)
    for B in (1, 2, 5):
        r = G.rollout(two, pb["x0"][:B], pb["U"][:B], cost)
        f = G.rollout(fused, pb["x0"][:B], pb["U"][:B], cost)
        np.testing.assert_allclose(r["means"].cpu().numpy(), c["means"][:B], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(r["vars"].cpu().numpy(), c["vars"][:B], rtol=1e-4, atol=1e-12)
        np.testing.assert_allclose(r["cost"].cpu().numpy(), c["cost"][:B], rtol=1e-6)
        np.testing.assert_allclose(r["grad"].cpu().numpy(), c["grad"][:B], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(f["cost"].cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-8)
        assert not torch.equal(f["vars"], r["vars"]) or N < 64        # two different kernels really ran
        o = G.rollout(two, pb["x0"][:B], pb["U"][:B], cost, want_grad=False)
        np.testing.assert_allclose(o["cost"].cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-9)
