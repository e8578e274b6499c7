#!/usr/bin/env python3
"""Long randomised parity soak of the FULL-covariance rollout (config 5's path) against the plain-C checker: means,
covariances, cost, and the analytic gradient held to complex-step directional derivatives of the C port; batch sizes on both
sides of the staged / scalar-broadcast switch.  Not part of the test-suite; run on the GPU box:
    python tools/soak_fullcov.py [n_cases] [seed]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
from oracle import cport, gpmpc_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
worst = {"means": 0.0, "covs": 0.0, "cost": 0.0, "ddir": 0.0}
bad = 0
t_start = time.time()
for case in range(n_cases):
    N = int(rng.choice([3, 40, 64, 65, 130, 200, 257, 449, 700]))
    ds = int(rng.integers(1, 6)); da = int(rng.integers(1, 3))
    H = int(rng.integers(1, 7))
    B = int(rng.choice([1, 2, 5, 40, 260, 700]))
    if N >= 449:
        B = min(B, 260)
    gamma = float(rng.choice([-1.0, 1e-5, 0.0, 0.5]))
    sb = str(rng.choice(["auto", "auto", "staged", "sbf"]))
    # round 4: the two-launch small-batch form (fullcov.hip::k_fc_head) on each tiling / column unroll / head split, and the four-launch form
    fc = str(rng.choice(["auto", "auto", "four", "two64", "two128", "two256c4", "two64c1r3", "two64c2r1"]))
    FC = {"auto": {}, "four": {"GPMPC_FC_FORM": "0"}, "two64": {"GPMPC_FC_FORM": "1", "GPMPC_FC_TILING": "2"},
          "two128": {"GPMPC_FC_FORM": "1", "GPMPC_FC_TILING": "4"}, "two256c4": {"GPMPC_FC_FORM": "1", "GPMPC_FC_TILING": "0", "GPMPC_FC_CU": "4"},
          "two64c1r3": {"GPMPC_FC_FORM": "1", "GPMPC_FC_TILING": "2", "GPMPC_FC_CU": "1", "GPMPC_FC_RSPLIT": "3"},
          "two64c2r1": {"GPMPC_FC_FORM": "1", "GPMPC_FC_TILING": "2", "GPMPC_FC_CU": "2", "GPMPC_FC_RSPLIT": "1"}}
    for k in ("GPMPC_FC_FORM", "GPMPC_FC_TILING", "GPMPC_FC_CU", "GPMPC_FC_RSPLIT", "GPMPC_FC_SHARED"):
        os.environ.pop(k, None)
    os.environ.update(FC[fc])
    # round 5: one lambda for all GPs on 40 % of the cases, the shared cross-unit kernel (pair_kernel_sbfx.h) forced on / off / by the plan
    shared = bool(rng.random() < 0.4)
    fsh = str(rng.choice(["on", "on", "off", "plan"])) if shared else "plan"
    if fsh != "plan":
        os.environ["GPMPC_FC_SHARED"] = "1" if fsh == "on" else "0"
    os.environ.pop("GPMPC_PAIR_SB", None)
    if sb == "staged":
        os.environ["GPMPC_PAIR_SB"] = "0"
    elif sb == "sbf":
        os.environ["GPMPC_PAIR_SB"] = "1"
    pb = synth_problem(7000 + case, N, ds, da, H, B, shared_lambda=shared)
    pb["Q"] = pb["Q"] + 0.02 * (np.ones((ds, ds)) - np.eye(ds))
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = g.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    r = g.rollout_fullcov(pack, pb["x0"], pb["U"], g.CostParams(gamma, pb["Q"], pb["R"]))
    pick = sorted({0, B - 1})
    dirs = rng.normal(size=(len(pick), 1, H, da))
    c = cport.rollout_fullcov(pb, kinv, gamma, x0=pb["x0"][pick], U=pb["U"][pick], dirs=dirs, nthreads=16)
    rel = lambda a, ref, fl: float(np.max(np.abs(a - ref) / np.maximum(np.abs(ref), fl)))   # noqa: E731
    err = {"means": rel(r["means"][pick].cpu().numpy(), c["means"], 1e-4),
           "covs": rel(r["covs"][pick].cpu().numpy(), c["covs"], 1e-2 * max(np.abs(c["covs"]).max(), 1e-30)),
           "cost": rel(r["cost"][pick].cpu().numpy(), c["cost"], 1e-6)}
    gr = r["grad"][pick].cpu().numpy()
    dd = np.array([float((gr[k] * dirs[k, 0]).sum()) for k in range(len(pick))])
    err["ddir"] = rel(dd, c["ddir"][:, 0], 1e-3)
    finite = all(bool(torch.isfinite(v).all()) for v in r.values())
    lab = ("sh-" + fsh) if shared else "distinct"
    ok = finite and err["means"] < 1e-5 and err["covs"] < 1e-4 and err["cost"] < 1e-6 and err["ddir"] < 1e-4
    bad += 0 if ok else 1
    print(f"case {case:3d}: N={N:4d} ds={ds} da={da} H={H} B={B:4d} gamma={gamma:g} {sb:6s} {fc:9s} {lab:8s} "
          + " ".join(f"{k} {v:.1e}" for k, v in err.items()) + ("" if ok else "   <-- FAIL"), flush=True)
    for k in worst:
        worst[k] = max(worst[k], err[k])
    del pack
print(f"{n_cases} cases in {time.time() - t_start:.0f} s, {bad} outside the tolerances; worst:", {k: f"{v:.1e}" for k, v in worst.items()})
sys.exit(1 if bad else 0)
