"""The N > 1 path on real hardware: two rank processes on the box's one GPU (gloo collectives) run the REAL HIP rollout
through parallel.sharded_rollout; the gathered [cost | grad] must equal the single-process batch bit for bit (the
trajectories are independent and every reduction has a fixed order), for an even split, a ragged split and a batch with
fewer trajectories than ranks.  SURVEY.md 8e; the reference has no multi-process code to compare with."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_on_one_card_match_the_single_process_batch(tmp_path):
    import gaussian_process_mpc_amd as g
    from gaussian_process_mpc_amd.synth import synth_problem
    g.require_gpu()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_multirank_worker.py"), str(r), "2", str(port), out], env=env)
             for r in range(2)]
    try:
        codes = [p.wait(timeout=600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0, 0]
    z = np.load(out)
    pb = synth_problem(23, 140, 3, 1, 5, 8)
    pack = g.GPPack(pb["X"], pb["Y"], z["kinv"], pb["lambdas"], pb["sigma_f"])
    cost = g.CostParams(-1.0, pb["Q"], pb["R"])
    for tag, B in (("even", 8), ("ragged", 7), ("fewer_than_ranks", 1)):
        r = g.rollout(pack, pb["x0"][:B], pb["U"][:B], cost, want_traj=False)
        assert np.array_equal(z[tag + "_cost"], r["cost"].cpu().numpy()), tag
        assert np.array_equal(z[tag + "_grad"], r["grad"].cpu().numpy()), tag
        assert z[tag + "_grad"].shape == (B, 5, 1)
    assert torch.isfinite(r["cost"]).all()


def test_rccl_backend_code_path_world1():
    """The one-GPU box cannot run two RCCL ranks, but it can run ONE: initialise the "nccl" (= RCCL) backend the way bench.py
    does and drive the RCCL branches of the sharding layer (all_gather_into_tensor of [cost|grad], the inverse-matrix
    broadcast, the max-reduce and barrier of the timing bracket) on real device tensors."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(HERE, "_nccl_world1_worker.py"), str(port)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_two_rccl_ranks_through_sharded_rollout(tmp_path):
    """Two RCCL ("nccl") ranks, one per GPU, through parallel.sharded_rollout: the gathered [cost | grad] equals the
    single-process batch bit for bit.  Needs two visible GPUs: SKIPPED on the one-GPU test box, runs on a multi-GPU node."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL with two ranks cannot run on one card)")
    import gaussian_process_mpc_amd as g
    from gaussian_process_mpc_amd.synth import synth_problem
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), GPMPC_TEST_BACKEND="nccl")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_multirank_worker.py"), str(r), "2", str(port), out], env=env)
             for r in range(2)]
    try:
        codes = [p.wait(timeout=600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0, 0]
    z = np.load(out)
    pb = synth_problem(23, 140, 3, 1, 5, 8)
    pack = g.GPPack(pb["X"], pb["Y"], z["kinv"], pb["lambdas"], pb["sigma_f"])
    cost = g.CostParams(-1.0, pb["Q"], pb["R"])
    for tag, B in (("even", 8), ("ragged", 7), ("fewer_than_ranks", 1)):
        r = g.rollout(pack, pb["x0"][:B], pb["U"][:B], cost, want_traj=False)
        assert np.array_equal(z[tag + "_cost"], r["cost"].cpu().numpy()), tag
        assert np.array_equal(z[tag + "_grad"], r["grad"].cpu().numpy()), tag
