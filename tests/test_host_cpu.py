"""CPU-only tests: the C-ABI library loads and exports everything include/gpmpc.h declares, host
logic (marshalling, sharding, collectives over gloo, synthetic generator), and the rule that the
product never routes through the oracle or a CPU fallback.  No GPU compute calls."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian_process_mpc_amd")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import gaussian_process_mpc_amd as g
    return g


def test_library_exports_every_declared_symbol(built):
    from gaussian_process_mpc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gpmpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gpmpc_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert getattr(h, name) is not None
    assert b"gfx950" in built.lib().gpmpc_version()


def test_cost_params_struct_layout_matches_header(built, tmp_path):
    from gaussian_process_mpc_amd._lib import CostParamsC
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gpmpc.h"\nint main(){printf("%zu %zu %zu %zu",'
                   'sizeof(gpmpc_cost_params), offsetof(gpmpc_cost_params,R), offsetof(gpmpc_cost_params,x_ref),'
                   'offsetof(gpmpc_cost_params,has_R_delta));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, off_R, off_xref, off_flag = (int(v) for v in subprocess.check_output([str(exe)]).split())
    assert size == ctypes.sizeof(CostParamsC)
    assert off_R == CostParamsC.R.offset and off_xref == CostParamsC.x_ref.offset
    assert off_flag == CostParamsC.has_R_delta.offset


def test_no_gpu_means_loud_failure_not_fallback(built):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = built
    with pytest.raises(g.GpmpcError):
        g.require_gpu()
    with pytest.raises(g.GpmpcError):
        g.GaussianProcessRegression(3)
    with pytest.raises(g.GpmpcError):
        g.RiskSensitiveMPC(-1.0, 5, 2, 1, np.eye(2), np.eye(1))
    with pytest.raises(g.GpmpcError):
        g.GPPack(np.zeros((4, 3)), np.zeros((4, 2)), np.zeros((2, 4, 4)), np.ones((2, 3)), np.ones(2))


def test_abi_argument_validation_without_device(built):
    lib = built.lib()
    h = ctypes.c_void_p()
    assert lib.gpmpc_pack_create(ctypes.byref(h), 0, 2, 1) == -1          # n_train < 1
    assert lib.gpmpc_pack_create(ctypes.byref(h), 10, 9, 1) == -1         # state_dim > GPMPC_MAX_DS
    assert lib.gpmpc_pack_create(ctypes.byref(h), 10, 4, 5) == -1         # D > GPMPC_MAX_D
    assert lib.gpmpc_pack_create(None, 10, 2, 1) == -1
    assert lib.gpmpc_pack_destroy(None) == 0
    assert lib.gpmpc_rollout(None, 1, 1, None, None, None, 0, None, None, None, None, None, 0, None) == -1
    assert lib.gpmpc_moment_match(None, 1, None, None, 0, *([None] * 10), None, 0, None) == -1
    assert lib.gpmpc_rollout_workspace_bytes(None, 1, 1, 0) == 0
    assert lib.gpmpc_predict_workspace_bytes(100, 3, 7) >= 2 * 7 * 100 * 8
    assert lib.gpmpc_matvec(0, 1, None, None, None, None) == -1
    assert lib.gpmpc_device_count() >= 0


def test_xcd_aware_dispatch_order_is_a_bijection_and_keeps_a_tile_on_its_xcd(built):
    """step_fused.h re-derives (trajectory, grid column) from the linear workgroup id (gpmpc_internal.h::gpmpc_xcd_remap, the same inline
    function on host and device).  Every (trajectory, column) must be visited exactly once -- a hole or a double visit is a missing or a
    twice-written partial sum --, and tile k of EVERY trajectory must sit on XCD k mod 8 (linear id mod 8) as long as k is within the last
    multiple of 8: that is what the order is for."""
    lib = built.lib()
    b, c = ctypes.c_int(), ctypes.c_int()
    for nt, ds, B in [(576, 4, 8), (577, 4, 3), (7, 2, 5), (8, 2, 2), (24, 4, 64), (163, 6, 2), (1, 1, 4), (40, 3, 33)]:
        gx = nt + 2 * ds
        seen = set()
        for L in range(gx * B):
            assert lib.gpmpc_debug_xcd_order(L, gx, nt, B, ctypes.byref(b), ctypes.byref(c)) == 0
            assert 0 <= b.value < B and 0 <= c.value < gx
            seen.add((b.value, c.value))
            if c.value < (nt & ~7):
                assert L % 8 == c.value % 8, (nt, B, L, b.value, c.value)
            if c.value >= nt:                                       # role workgroups come after every tile workgroup
                assert L >= nt * B
        assert len(seen) == gx * B, (nt, ds, B)
    assert lib.gpmpc_debug_xcd_order(10, 8, 9, 2, ctypes.byref(b), ctypes.byref(c)) == -1       # more tile columns than columns
    assert lib.gpmpc_debug_xcd_order(16, 8, 4, 2, ctypes.byref(b), ctypes.byref(c)) == -1       # id beyond the grid


def test_balanced_run_list_covers_every_column_once_within_one_generation(built):
    """pack.hip::build_worklist_runs (host part): for ONE trajectory of a large training set every 256-row tile row of every GP is cut into
    runs of at most 256 columns, multiples of 8, contiguous from the tile row's first column to the padded size, and the whole list fits the
    workgroup slots it was built for; small training sets get no list."""
    lib = built.lib()
    n = ctypes.c_int()
    for Np, ds, slots in [(4096, 6, 1024 - 12), (2560, 6, 1012), (3072, 5, 1014), (2752, 5, 1014), (4096, 7, 1010), (4032, 6, 1012)]:
        assert lib.gpmpc_debug_run_list(Np, ds, slots, None, 0, ctypes.byref(n)) == 0
        k = n.value
        assert 0 < k <= slots, (Np, ds, slots, k)
        items = (ctypes.c_int * (4 * k))()
        assert lib.gpmpc_debug_run_list(Np, ds, slots, items, k, ctypes.byref(n)) == 0 and n.value == k
        it = np.array(items[:]).reshape(k, 4)
        assert (np.diff(it[:, 0]) >= 0).all() and set(it[:, 0]) == set(range(ds))            # unit-contiguous, every GP present
        for u in range(ds):
            for i0 in range(0, Np, 256):
                runs = it[(it[:, 0] == u) & (it[:, 1] == i0)]
                assert len(runs) >= 1 and runs[0, 2] == i0 and runs[-1, 3] == Np
                assert (runs[1:, 2] == runs[:-1, 3]).all()                                    # contiguous, no overlap
                lens = runs[:, 3] - runs[:, 2]
                assert (lens > 0).all() and (lens <= 256).all() and (lens % 8 == 0).all()
        lens = it[:, 3] - it[:, 2]
        assert lens.max() - np.median(lens) <= 16                                             # balanced: equal runs up to the 8-column grain
    assert lib.gpmpc_debug_run_list(2048, 4, 1016, None, 0, ctypes.byref(n)) == 0 and n.value == 0      # about one generation already
    assert lib.gpmpc_debug_run_list(320, 4, 1016, None, 0, ctypes.byref(n)) == 0 and n.value == 0
    assert lib.gpmpc_debug_run_list(100, 4, 1016, None, 0, ctypes.byref(n)) == -1                       # not a padded size
    assert lib.gpmpc_debug_run_list(8192, 8, 1008, None, 0, ctypes.byref(n)) == 0 and n.value == 0      # more than one generation even at 256 columns


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), f
                assert "oracle" not in txt.lower() or f == "__init__.py" or "the oracle is pinned" in txt or \
                    all("import" not in ln for ln in txt.splitlines() if "oracle" in ln.lower()), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    hits = [m.start() for m in pat.finditer(bench)]
    start = bench.index("def cpu_baseline")
    end = bench.index("def main")
    assert 1 <= len(hits) <= 2 and all(start < h < end for h in hits)      # torch oracle + its C port, both inside cpu_baseline


def test_cost_params_marshalling(built):
    g = built
    Q = np.array([[2.0, 0.5], [0.25, 3.0]])
    R = np.array([[1.5]])
    c = g.CostParams(-1.0, Q, R, R_delta=np.array([[0.7]]), x_ref=[0.1, 0.2], u_ref=[0.3], last_u=np.array([9.0, 8.0]))
    assert c.c.gamma == -1.0 and c.c.has_R_delta == 1
    assert list(c.c.Q[:4]) == [2.0, 0.5, 0.25, 3.0] and c.c.R[0] == 1.5 and c.c.R_delta[0] == 0.7
    assert list(c.c.x_ref[:2]) == [0.1, 0.2] and c.c.u_ref[0] == 0.3 and c.c.last_u[0] == 9.0
    assert g.CostParams(0.0, np.eye(3), np.eye(2)).c.has_R_delta == 0
    with pytest.raises(ValueError):
        g.CostParams(1.0, np.eye(9), np.eye(1))


def test_synth_problem_is_seeded_and_shaped(built):
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    a = synth_problem(3, 64, 4, 1, 5, 3)
    b = synth_problem(3, 64, 4, 1, 5, 3)
    for k in ("X", "Y", "lambdas", "x0", "U"):
        assert np.array_equal(a[k], b[k])
    assert a["X"].shape == (64, 5) and a["Y"].shape == (64, 4) and a["U"].shape == (3, 5, 1)
    assert not np.array_equal(a["X"], synth_problem(4, 64, 4, 1, 5, 3)["X"])
    assert CONFIGS["C3"] == dict(N=2048, ds=4, da=1, H=20, B=256, gamma=-1.0)
    assert CONFIGS["C4"]["N"] == 4096 and CONFIGS["C2"]["B"] == 1


def test_shard_ranges_partition(built):
    from gaussian_process_mpc_amd.parallel import shard_range, shard_sizes
    for n, w in ((256, 8), (1024, 8), (10, 3), (5, 8), (1, 2)):
        covered = []
        for r in range(w):
            lo, hi = shard_range(n, w, r)
            covered += list(range(lo, hi))
        assert covered == list(range(n))
        assert sum(shard_sizes(n, w)) == n and max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1


def _gloo_worker(rank, world, port, ragged, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from gaussian_process_mpc_amd.parallel import gather_results, shard_range, shard_sizes, broadcast_kinv
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 7 if ragged else 8
    H, da = 3, 2
    cost_all = torch.arange(n, dtype=torch.float64) * 1.5
    grad_all = torch.arange(n * H * da, dtype=torch.float64).reshape(n, H, da)
    lo, hi = shard_range(n, world, rank)
    sizes = shard_sizes(n, world)
    c, g = gather_results(cost_all[lo:hi].clone(), grad_all[lo:hi].clone(), dist, sizes if ragged else None)
    c2, g2 = gather_results(cost_all[lo:hi].clone(), None, dist, sizes if ragged else None)
    from gaussian_process_mpc_amd.parallel import sharded_rollout
    fake = lambda x0b, Ub: {"cost": Ub.sum(dim=(1, 2)), "grad": 2.0 * Ub}      # stands in for the device rollout  # noqa: E731
    Uall = torch.arange(n * H * da, dtype=torch.float64).reshape(n, H, da) / 7.0
    cs, gs = sharded_rollout(fake, torch.zeros(3), Uall, dist)
    ok_sharded = bool(torch.equal(cs, Uall.sum(dim=(1, 2))) and torch.equal(gs, 2.0 * Uall))
    kinv = torch.full((2, 4, 4), float(rank + 1), dtype=torch.float64)
    broadcast_kinv(kinv, dist, src=0)
    ok = bool(torch.equal(c, cost_all) and torch.equal(g, grad_all) and torch.equal(c2, cost_all) and g2 is None
              and torch.all(kinv == 1.0) and ok_sharded)
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_gather_results_gloo_world2(ragged):
    """The N>1 path of bench.py / the sharded batch API: fused all_gather of [cost | grad] and the
    pack broadcast, world_size 2 over gloo on CPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + (50 if ragged else 0)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _gloo_worker_more_ranks_than_items(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from gaussian_process_mpc_amd.parallel import sharded_rollout
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, da = 3, 2
    fake = lambda x0b, Ub: {"cost": Ub.sum(dim=(1, 2)), "grad": 2.0 * Ub}      # noqa: E731
    ok = True
    for n in (1, 2):                                    # fewer trajectories than ranks: some ranks own an empty block
        Uall = torch.arange(n * H * da, dtype=torch.float64).reshape(n, H, da) / 3.0
        cs, gs = sharded_rollout(fake, torch.zeros(3), Uall, dist)
        ok = ok and bool(torch.equal(cs, Uall.sum(dim=(1, 2))) and torch.equal(gs, 2.0 * Uall))
        cs, gs = sharded_rollout(fake, torch.zeros(3), Uall, dist, want_grad=False)
        ok = ok and bool(torch.equal(cs, Uall.sum(dim=(1, 2))) and gs is None)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_rollout_more_ranks_than_trajectories_gloo_world3():
    """world > B: the ranks without a trajectory contribute an empty block and must still enter the collective (an
    ambiguous reshape of the (0, H, da) gradient used to raise on them while the others hung in all_gather)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 120
    procs = [ctx.Process(target=_gloo_worker_more_ranks_than_items, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True), (2, True)]


def test_bench_refuses_to_mislabel_the_rank_count():
    """bench.py --gpus N must run N ranks or fail: (i) with WORLD_SIZE already set to something else it exits non-zero
    before touching a device; (ii) with no WORLD_SIZE and fewer visible GPUs than N the launcher exits non-zero (here:
    zero GPUs) instead of printing a 1-GPU line -- the round-1 behaviour the driver would have recorded as `n_gpus: 1`."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout


def test_bench_launcher_notices_a_dead_rank_within_seconds():
    """bench.py's launcher polls every child: when rank 1 dies while rank 0 would sit in a collective for minutes, the job
    ends non-zero in well under 30 s (it used to block on rank 0's stdout until the process-group timeout)."""
    import time
    import types
    sys.path.insert(0, ROOT)
    import bench
    worker = ("import os, sys, time\n"
              "r = int(os.environ['RANK'])\n"
              "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
              "if r == 1:\n    time.sleep(1.0); sys.exit(7)\n"
              "print('{\"rank\": %d}' % r, flush=True)\n"
              "time.sleep(600)\n")
    t0 = time.time()
    rc = bench.spawn_ranks(types.SimpleNamespace(gpus=3, oversubscribe=True), argv=[sys.executable, "-c", worker])
    assert rc == 7 and time.time() - t0 < 30.0
    ok = "import os\nprint('{\"n\": %s}' % os.environ['RANK'], flush=True)\n"
    assert bench.spawn_ranks(types.SimpleNamespace(gpus=2, oversubscribe=True), argv=[sys.executable, "-c", ok]) == 0


def test_plants_match_the_reference_environments(golden):
    """the plant updates either side of the path, as plain classes, against the reference's own
    environment classes run by tests/golden/gen_golden.py (g9): cart-pole stepPhysics / step
    (src/environments/continuous_cartpole.py:71-101) and pendulum step_static / step
    (src/environments/adjustable_pendulum.py:135-178), bit for bit (same operations in the same order)."""
    from gaussian_process_mpc_amd.simulator import CartPolePlant, PendulumPlant
    z = golden("g9_closed_loop.npz")
    cp = CartPolePlant()
    assert [cp.gravity, cp.masscart, cp.masspole, cp.length, cp.force_mag, cp.tau] == list(z["cp_params"])
    nxt = np.array([cp.stepPhysics(float(f), tuple(s)) for s, f in zip(z["cp_states"], z["cp_forces"])])
    assert np.array_equal(nxt, z["cp_next"])
    cp = CartPolePlant(init_state=z["cp_chain_x0"])
    cp.reset()
    chain = []
    for a in z["cp_chain_actions"]:
        obs, rew, term, trunc, _ = cp.step(np.array([a]))
        assert rew == 1 and not term and not trunc
        chain.append(obs)
    assert np.array_equal(np.array(chain), z["cp_chain"])
    with pytest.raises(AssertionError):
        cp.step(np.array([1.0]))                                  # the reference asserts -1 < action < 1
    s0, _ = CartPolePlant(seed=3).reset()
    assert s0.shape == (4,) and np.all(np.abs(s0) <= 0.2)
    opts = dict(zip(("g", "m", "l", "dt", "max_torque", "max_speed"), z["pd_opts"]))
    st = np.array([PendulumPlant.step_static(s, u, opts) for s, u in zip(z["pd_states"], z["pd_u"])])
    assert np.array_equal(st, z["pd_static_next"])
    pd = PendulumPlant(g=10.0, max_speed=8, max_torque=2.0, init_state=(np.pi, 0.0))
    obs, _ = pd.reset()
    states, rewards = [obs], []
    for u in z["pd_chain_actions"]:
        obs, r, _, _, _ = pd.step(u)
        states.append(obs); rewards.append(r)
    assert np.array_equal(np.array(states), z["pd_chain"])
    np.testing.assert_allclose(rewards, z["pd_chain_reward"], rtol=1e-15)
    assert np.array_equal(PendulumPlant.step_static(z["pd_chain"][3], z["pd_chain_actions"][3], pd.options()), z["pd_chain"][4])


# ------------------------------------------------------------------------------------------------------------------
# round 5: lock-step multi-start driver (host logic; the GPU variant is tests/test_gpu_api.py::test_closed_loop_simulator_multistart)
# ------------------------------------------------------------------------------------------------------------------
def _rosen_batch(X):
    f = np.sum(100 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1 - X[:, :-1]) ** 2, axis=1)
    g = np.zeros_like(X)
    g[:, :-1] += -400 * X[:, :-1] * (X[:, 1:] - X[:, :-1] ** 2) - 2 * (1 - X[:, :-1])
    g[:, 1:] += 200 * (X[:, 1:] - X[:, :-1] ** 2)
    return f, g


def test_lockstep_multistart_reaches_the_bounded_optima_scipy_finds():
    """K projected L-BFGS searches advanced together, ONE batched evaluation per tick (multistart.py: the consumer of the
    trajectory batch inside a solve, src/mpc.py:269-330): on a bounded Rosenbrock every start ends where scipy's L-BFGS-B ends from
    the same point; evaluations = ticks + 1 (never one call per start); a region that returns NaN (the risk-sensitive cost's
    log det of a non-positive matrix, src/mpc.py:183) rejects trial points and drops starts that begin inside it."""
    from scipy.optimize import minimize
    from gaussian_process_mpc_amd.multistart import lockstep_lbfgs, make_starts
    n = 8
    lb, ub = -2 * np.ones(n), 0.8 * np.ones(n)
    X0 = make_starts(12, n, lb, ub, np.random.default_rng(1), warm=np.full(n, 0.5))
    assert np.array_equal(X0[0], np.zeros(n)) and np.array_equal(X0[1], np.full(n, 0.5)) and (X0 >= lb).all() and (X0 <= ub).all()
    calls = []

    def ev(X):
        calls.append(X.shape)
        return _rosen_batch(X)
    x, info = lockstep_lbfgs(ev, X0, lb, ub, max_ticks=2000, gtol=1e-6)
    assert info["converged"].all() and len(calls) == info["evaluations"] == info["ticks"] + 1
    assert all(c == (12, n) for c in calls)
    for k in (0, 1, 5):
        r = minimize(lambda v: _rosen_batch(v[None])[0][0], X0[k], jac=lambda v: _rosen_batch(v[None])[1][0], method="L-BFGS-B",
                     bounds=list(zip(lb, ub)), options={"gtol": 1e-6, "ftol": 1e-14})
        assert abs(info["f"][k] - r.fun) <= 1e-6 * max(1.0, abs(r.fun))
    assert np.isclose(_rosen_batch(x[None])[0][0], info["f"].min()) and (x <= ub + 1e-15).all()
    x2, info2 = lockstep_lbfgs(_rosen_batch, X0, lb, ub, max_ticks=2000, gtol=1e-6)      # deterministic: same bits
    assert np.array_equal(x, x2) and np.array_equal(info["f"], info2["f"])
    # four step lengths per start and tick (the line-search evaluations of an iteration as ONE batch of 4 K plans): fewer ticks, same optima
    shapes = []

    def ev4(X):
        shapes.append(X.shape)
        return _rosen_batch(X)
    x4, info4 = lockstep_lbfgs(ev4, X0, lb, ub, max_ticks=2000, gtol=1e-6, line_points=4)
    assert info4["converged"].all() and info4["ticks"] < info["ticks"] and set(shapes[1:]) == {(48, n)} and shapes[0] == (12, n)
    np.testing.assert_allclose(info4["f"], info["f"], rtol=1e-6)

    def with_nan(X):
        f, g = np.sum((X - 0.3) ** 2, axis=1), 2 * (X - 0.3)
        return np.where(X[:, 0] > 0.9, np.nan, f), g
    x, info = lockstep_lbfgs(with_nan, np.array([[0.95, 0, 0], [0.0, 0, 0], [-1, 1, 1.0]]), -np.ones(3), np.ones(3))
    assert list(info["alive"]) == [False, True, True] and np.allclose(x, 0.3) and info["best"] in (1, 2)


def _gloo_multistart_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from gaussian_process_mpc_amd.multistart import lockstep_lbfgs, make_starts
    from gaussian_process_mpc_amd.parallel import sharded_rollout
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K, H, da = 5, 4, 2                                   # ragged: 3 + 2 starts
    n = H * da
    lb, ub = -2 * np.ones(n), 0.8 * np.ones(n)
    X0 = make_starts(K, n, lb, ub, np.random.default_rng(7))
    blocks = []

    def fake_rollout(x0b, Ub):                            # stands in for the device rollout of this rank's block of plans
        blocks.append(Ub.shape[0])
        f, g = _rosen_batch(Ub.reshape(Ub.shape[0], n).numpy())
        return {"cost": torch.as_tensor(f), "grad": torch.as_tensor(g).reshape(-1, H, da)}

    def evaluate(X):
        c, g = sharded_rollout(fake_rollout, torch.zeros(3), torch.as_tensor(X.reshape(K, H, da)), dist)
        return c.numpy(), g.numpy().reshape(K, n)
    x, info = lockstep_lbfgs(evaluate, X0, lb, ub, max_ticks=1500, gtol=1e-6)
    xs, infos = lockstep_lbfgs(_rosen_batch, X0, lb, ub, max_ticks=1500, gtol=1e-6)          # the same solve in one process
    ok = bool(np.array_equal(x, xs) and np.array_equal(info["f"], infos["f"]) and info["ticks"] == infos["ticks"]
              and set(blocks) == {3 if rank == 0 else 2})
    q.put((rank, ok, x.tobytes()))
    dist.destroy_process_group()


def test_lockstep_multistart_sharded_over_two_gloo_ranks():
    """The sharded variant of the multi-start solve (RiskSensitiveMPC._solve_multistart with torch.distributed initialised): each rank
    evaluates ITS block of the K trial plans, one fused all_gather returns every [cost | grad] to every rank, and both ranks run
    the same deterministic host loop -- identical plans on both ranks, bit-equal to the single-process solve."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 170
    procs = [ctx.Process(target=_gloo_multistart_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)] and res[0][2] == res[1][2]


# ------------------------------------------------------------------------------------------------------------------
# round 5: build-time spill guard, compact bench record
# ------------------------------------------------------------------------------------------------------------------
def test_spill_guard_passes_on_the_shipped_build_and_fails_on_a_new_spill(built, tmp_path):
    """`make check` (tools/spill_guard.py over the <object>.res files the build writes): every shipped kernel within its VGPR-spill /
    scratch budget -- the whole-horizon instances for D <= 7 at ZERO spills --, and a kernel that starts to spill fails the check."""
    import subprocess
    build = os.path.join(PKG, "csrc", "build")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spill_guard.py"), build, "--all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("k_traj_persist<")]
    assert len(rows) >= 40
    for ln in rows:
        if not re.match(r"k_traj_persist<8, ", ln):
            assert " spill v  0 " in ln and ln.rstrip().endswith(" ok"), ln
    bad = tmp_path / "b"
    bad.mkdir()
    (bad / "x.o.res").write_text(
        "f.h:1:1: remark: Function Name: _Z14k_traj_persistILi5ELi4ELb1ELi1EEv11PersistArgs [-Rpass-analysis=kernel-resource-usage]\n"
        "f.h:1:1: remark:     VGPRs: 128 [-Rpass-analysis=kernel-resource-usage]\n"
        "f.h:1:1: remark:     ScratchSize [bytes/lane]: 48 [-Rpass-analysis=kernel-resource-usage]\n"
        "f.h:1:1: remark:     VGPRs Spill: 11 [-Rpass-analysis=kernel-resource-usage]\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spill_guard.py"), str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "OVER BUDGET" in r.stderr


def test_bench_stdout_record_is_compact():
    """The driver keeps an 8 KB tail of stdout: the default line (headline + roofline + cpu_baseline + every leg) must fit in 6 KB,
    each leg in ~250 bytes with value / ms_per_step / kernel / bound / frac / avg_launch_ms."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04", "bench_C3.json")))      # a full record of the previous format
    legs = {}
    for name, _ in bench.LEGS:
        legs[name] = {"value": 12345.678, "unit": "rollouts/s", "ms_per_step": 1.2345678, "kernel": "gpmpc_pair_kernel_sbs<5,4,4,true,false>",
                      "bound": "valu_fp64", "frac": 0.5021, "avg_launch_ms": 1.30712}
    full["extras"] = legs
    full["extras_full"] = {"x": "y" * 5000}
    line = json.dumps(bench.compact_record(full), separators=(",", ":"))
    assert len(line) < 6000, len(line)
    d = json.loads(line)
    assert "extras_full" not in d and set(d["extras"]) == {n for n, _ in bench.LEGS}
    assert all(len(json.dumps(v, separators=(",", ":"))) <= 250 for v in d["extras"].values())
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"]
    assert bench.short_kernel("k_step_fused<5,4,true,32,1> (one launch per horizon step: ...)") == "k_step_fused<5,4,true,32,1>"
