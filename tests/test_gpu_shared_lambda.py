"""Shared length-scales: the regime of every experiment of the reference (one lambda for all GPs of the bundle,
src/experiments/pretrain_uncertainty.py:100-105, pretrain_pendulum.py:54-55, pretrain_cts_cartpole.py:42-43).

The pack detects bit-identical lambda rows and the rollout switches to pair_kernel_sbs.h (exponent and exp once per pair
for a group of GPs).  Held here to the plain-C port (oracle/cport: means 1e-5, variances 1e-4, cost 1e-6, gradient
1e-4) at the pendulum and cart-pole shapes, at every (state_dim, action_dim) the dispatcher serves, at C3 sizes, and to
the distinct-lambda kernels on the same pack (GPMPC_SHARED=0).
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


def _problem(seed, N, ds, da, H, B, shared=True, lam_range=(2.0, 6.0)):
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    pb = synth_problem(seed, N, ds, da, H, B, shared_lambda=shared, lam_range=lam_range)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    return pb, gp.Ky_inv.numpy()


def _check_vs_cport(r, pb, kinv, pick, tag):
    from oracle import cport
    c = cport.rollout(pb, kinv, -1.0, x0=pb["x0"][pick], U=pb["U"][pick], nthreads=8)
    np.testing.assert_allclose(r["means"][pick].cpu().numpy(), c["means"], rtol=1e-5, atol=1e-9, err_msg=tag)
    np.testing.assert_allclose(r["vars"][pick].cpu().numpy(), c["vars"], rtol=1e-4, atol=1e-12, err_msg=tag)
    np.testing.assert_allclose(r["cost"][pick].cpu().numpy(), c["cost"], rtol=1e-6, err_msg=tag)
    np.testing.assert_allclose(r["grad"][pick].cpu().numpy(), c["grad"], rtol=1e-4, atol=1e-7, err_msg=tag)


def test_detection_is_bit_exact(G):
    pb, kinv = _problem(7, 100, 3, 1, 2, 2)
    assert G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"]).shared_lambda
    lam = pb["lambdas"].copy()
    lam[2, 1] = np.nextafter(lam[2, 1], 10.0)                       # one ulp off: not shared
    assert not G.GPPack(pb["X"], pb["Y"], kinv, lam, pb["sigma_f"]).shared_lambda
    pb2, kinv2 = _problem(7, 100, 3, 1, 2, 2, shared=False)
    assert not G.GPPack(pb2["X"], pb2["Y"], kinv2, pb2["lambdas"], pb2["sigma_f"]).shared_lambda
    # sigma_f may differ between the GPs: it only scales the folded weights
    sf = np.array([1.0, 0.7, 1.3])
    assert G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], sf).shared_lambda


@pytest.mark.parametrize("ds,da", [(2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 2), (5, 1), (5, 2), (6, 1), (6, 2), (7, 1)])
def test_every_shape_vs_cport_and_vs_distinct_kernels(G, ds, da, monkeypatch):
    """256x64 and 256x256 shared work lists, GRAD / objective-only / horizon-step-1 instances of every (ds, da), incl. the
    partial last GP group (ds = 5: groups of 3 + 2)."""
    N, H = 150, 3
    b_mid = 2048 // (3 * ds) + 2                       # the scalar-broadcast path starts here (B * 3 ds >= 1700): 256x64 list
    b_big = max(5600 // ds + 3, 1600)                  # >= 1500 shared workgroups: 256x256 list
    pb, kinv = _problem(140 + 8 * ds + da, N, ds, da, H, b_big)
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    assert pack.shared_lambda
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    res = {}
    for B in (b_mid, b_big):
        r = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost)
        assert all(torch.isfinite(v).all() for v in r.values())
        _check_vs_cport(r, pb, kinv, sorted({0, 1, B // 2, B - 1}), f"ds={ds} da={da} B={B}")
        f = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost, want_grad=False)       # the GRAD = false instances
        np.testing.assert_allclose(f["cost"].cpu().numpy(), r["cost"].cpu().numpy(), rtol=1e-9)
        np.testing.assert_allclose(f["vars"].cpu().numpy(), r["vars"].cpu().numpy(), rtol=1e-7)
        again = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost)
        for k in r:
            assert torch.equal(r[k], again[k]), k                                    # fixed-order reductions
        res[B] = r
    # the one-launch-per-step form with a group of GPs per tile workgroup (step_fused.h, NG > 1: the plan from ~2200 to ~7000 tile
    # workgroups of a training set of at least one 256-row tile), forced onto the mid-size batch: against the C port and against the
    # two-launch shared-lambda kernel above (same tiles, same order inside a tile)
    monkeypatch.setenv("GPMPC_FUSED_SB", "1")
    pack.reload_tuning()
    o = G.rollout(pack, pb["x0"][:b_mid], pb["U"][:b_mid], cost)
    _check_vs_cport(o, pb, kinv, sorted({0, 1, b_mid // 2, b_mid - 1}), f"one-launch form ds={ds} da={da}")
    np.testing.assert_allclose(res[b_mid]["vars"].cpu().numpy(), o["vars"].cpu().numpy(), rtol=1e-6, atol=1e-14)
    np.testing.assert_allclose(res[b_mid]["grad"].cpu().numpy(), o["grad"].cpu().numpy(), rtol=1e-5, atol=1e-9)
    of = G.rollout(pack, pb["x0"][:b_mid], pb["U"][:b_mid], cost, want_grad=False)
    np.testing.assert_allclose(of["cost"].cpu().numpy(), o["cost"].cpu().numpy(), rtol=1e-9)
    monkeypatch.delenv("GPMPC_FUSED_SB")
    pack.reload_tuning()
    # the whole-horizon kernel over units of TWO (or all three / four) GPs (traj_persist.h, NG = 2 ... 4: the plan for ~one trajectory per CU and more of a training
    # set of up to 512 points; instantiated up to D = 6 -- beyond, and for a single GP, the distinct-lambda instance runs), forced onto a
    # small batch, 16 and 8 waves per workgroup: against the C port and, tightly, against the step-per-launch result
    for pw in ("16", "8"):
        monkeypatch.setenv("GPMPC_PERSIST", pw)
        pack.reload_tuning()
        pl = pack.plan(9, H)
        # GPs per unit: all of them where that instance exists (ds = 4 at D = 5, ds = 3 at D <= 5; 16-wave workgroups), else two
        ng = 1 if not 3 <= ds + da <= 6 else ((ds if pw == "16" and ((ds == 4 and ds + da == 5) or (ds == 3 and ds + da <= 5)) else 2))
        assert pl["form"] == "persist" and (f",{ng}>" in pl["kernel"]), pl
        w = G.rollout(pack, pb["x0"][:9], pb["U"][:9], cost)
        _check_vs_cport(w, pb, kinv, [0, 4, 8], f"whole-horizon form ds={ds} da={da} waves={pw}")
        np.testing.assert_allclose(w["means"].cpu().numpy(), res[b_mid]["means"][:9].cpu().numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(w["vars"].cpu().numpy(), res[b_mid]["vars"][:9].cpu().numpy(), rtol=1e-6, atol=1e-14)
        np.testing.assert_allclose(w["grad"].cpu().numpy(), res[b_mid]["grad"][:9].cpu().numpy(), rtol=1e-5, atol=1e-9)
        wf = G.rollout(pack, pb["x0"][:9], pb["U"][:9], cost, want_grad=False)
        np.testing.assert_allclose(wf["cost"].cpu().numpy(), w["cost"].cpu().numpy(), rtol=1e-9)
    monkeypatch.delenv("GPMPC_PERSIST")
    pack.reload_tuning()
    # the distinct-lambda kernels on the same pack: the same sums, the exponent rounded the same way -> agreement far inside
    # the tolerance (the tilings and the summation order of the partial sums are the same)
    monkeypatch.setenv("GPMPC_SHARED", "0")
    pack.reload_tuning()
    for B in (b_mid, b_big):
        d = G.rollout(pack, pb["x0"][:B], pb["U"][:B], cost)
        np.testing.assert_allclose(res[B]["means"].cpu().numpy(), d["means"].cpu().numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res[B]["vars"].cpu().numpy(), d["vars"].cpu().numpy(), rtol=1e-6, atol=1e-14)
        np.testing.assert_allclose(res[B]["grad"].cpu().numpy(), d["grad"].cpu().numpy(), rtol=1e-5, atol=1e-9)
    monkeypatch.delenv("GPMPC_SHARED")
    pack.reload_tuning()


@pytest.mark.parametrize("name,N,ds,da,H,B", [("pendulum", 400, 2, 1, 10, 700), ("cartpole", 600, 4, 1, 10, 160),
                                              ("pendulum-ragged", 333, 2, 1, 6, 301)])
def test_reference_experiment_shapes_vs_cport(G, name, N, ds, da, H, B):
    """Pendulum (ds = 2, da = 1) and cart-pole (ds = 4, da = 1) shapes with the experiments' single lambda for all GPs."""
    pb, kinv = _problem(len(name), N, ds, da, H, B, lam_range=(0.5, 0.5 + 1e-9))
    pb["lambdas"][:] = 0.5                                              # pretrain_uncertainty.py:102: lambda = 0.5 everywhere
    from oracle import gpmpc_oracle as O
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    assert pack.shared_lambda
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    r = G.rollout(pack, pb["x0"], pb["U"], cost)
    assert all(torch.isfinite(v).all() for v in r.values())
    _check_vs_cport(r, pb, kinv, [0, B // 3, B - 1], name)


def test_c3_sizes_shared_lambda_vs_cport(G):
    """N = 2048, ds = 4, da = 1, H = 20 with one lambda for all GPs: B = 64 batch (256x256 shared list), three trajectories
    over the whole horizon against the C port; autograd-boundary forward (full step-1 Jacobians) on the same pack."""
    from gaussian_process_mpc_amd.synth import CONFIGS
    cfg = CONFIGS["C3"]
    torch.set_num_threads(16)
    pb, kinv = _problem(3, cfg["N"], cfg["ds"], cfg["da"], cfg["H"], 64)
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    assert pack.shared_lambda
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    r = G.rollout(pack, pb["x0"], pb["U"], cost)
    assert all(torch.isfinite(v).all() for v in r.values())
    _check_vs_cport(r, pb, kinv, [0, 31, 63], "C3 shared")
    r8 = G.rollout(pack, pb["x0"][:8], pb["U"][:8], cost)                 # 256x64 shared list
    np.testing.assert_allclose(r8["means"].cpu().numpy(), r["means"][:8].cpu().numpy(), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r8["vars"].cpu().numpy(), r["vars"][:8].cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(r8["grad"].cpu().numpy(), r["grad"][:8].cpu().numpy(), rtol=1e-4, atol=1e-8)
    from gaussian_process_mpc_amd.autograd import RolloutFunction
    U = torch.tensor(pb["U"][:8], device=pack.device, requires_grad=True)
    m, v = RolloutFunction.apply(torch.tensor(pb["x0"][:8], device=pack.device), U, pack)
    np.testing.assert_allclose(m.detach().cpu().numpy(), r8["means"].cpu().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(v.detach().cpu().numpy(), r8["vars"].cpu().numpy(), rtol=1e-7)


def test_c4_sizes_shared_lambda_vs_cport(G):
    """N = 4096, ds = 6, da = 1 with one lambda for all GPs: groups of three GPs per workgroup (the NG = 3 instances at D = 7),
    B = 24 on the 256x256 shared list, two trajectories over H = 6 steps against the C port."""
    torch.set_num_threads(16)
    pb, kinv = _problem(4, 4096, 6, 1, 6, 24)
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    assert pack.shared_lambda
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    r = G.rollout(pack, pb["x0"], pb["U"], cost)
    assert all(torch.isfinite(v).all() for v in r.values())
    _check_vs_cport(r, pb, kinv, [0, 23], "C4 sizes shared")


def test_captured_graphs_follow_a_change_of_the_lambda_sharing(G):
    """A pack refilled in place (GPPack.rebuild) keeps its captured launch sequences -- unless the refill changes whether the
    lambdas are shared: a graph captured with the shared-lambda kernel must not be replayed on distinct lambdas (and the other
    way round the replay would merely be slow).  Graph-replayed values after each refill == eager values of a fresh pack."""
    pb, kinv = _problem(9, 700, 3, 1, 4, 40, shared=True)
    pbd, kinvd = _problem(9, 700, 3, 1, 4, 40, shared=False)
    cost = G.CostParams(-1.0, pb["Q"], pb["R"])
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    X, Y = torch.as_tensor(pb["X"], device=pack.device), torch.as_tensor(pb["Y"], device=pack.device)
    for step, (lam, kv, want_shared) in enumerate([(pb["lambdas"], kinv, True), (pbd["lambdas"], kinvd, False), (pb["lambdas"], kinv, True)]):
        if step > 0:
            assert pack.rebuild(X, Y, torch.as_tensor(kv, device=pack.device), lam, pb["sigma_f"])
        assert pack.shared_lambda == want_shared
        g1 = G.rollout(pack, pb["x0"], pb["U"], cost, graph=True)
        g2 = G.rollout(pack, pb["x0"], pb["U"], cost, graph=True)           # the replay
        fresh = G.rollout(G.GPPack(pb["X"], pb["Y"], kv, lam, pb["sigma_f"]), pb["x0"], pb["U"], cost)
        for k in ("cost", "grad", "means", "vars"):
            assert torch.equal(g2[k], fresh[k]) and torch.equal(g1[k], fresh[k]), (step, k)
