#!/usr/bin/env python3
"""Randomised parity sweep of the full-covariance rollout (BASELINE config 5's path) against the extension oracle.
    python tools/fuzz_fullcov.py [n_cases] [seed]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
from oracle import gpmpc_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
worst = {"means": 0.0, "covs": 0.0, "cost": 0.0, "grad": 0.0}
for case in range(n_cases):
    N = int(rng.choice([3, 17, 64, 65, 100, 129, 200]))
    ds = int(rng.integers(1, 5)); da = int(rng.integers(1, 3)); H = int(rng.integers(1, 5)); B = int(rng.integers(1, 4))
    gamma = float(rng.choice([-1.0, 1e-5, 0.0, 0.5]))
    pb = synth_problem(300 + case, N, ds, da, H, B)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = g.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    r = g.rollout_fullcov(pack, pb["x0"], pb["U"], g.CostParams(gamma, pb["Q"], pb["R"]))
    err = {k: 0.0 for k in worst}
    for b in range(B):
        o = O.objective_and_gradient_fullcov(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], gamma)
        scale = np.max(np.abs(o["covs"])) + 1e-300
        err["means"] = max(err["means"], float(np.max(np.abs(r["means"][b].cpu().numpy() - o["means"]) / np.maximum(np.abs(o["means"]), 1e-6))))
        err["covs"] = max(err["covs"], float(np.max(np.abs(r["covs"][b].cpu().numpy() - o["covs"])) / scale))
        err["cost"] = max(err["cost"], abs(r["cost"][b].item() - o["cost"]) / max(abs(o["cost"]), 1e-9))
        err["grad"] = max(err["grad"], float(np.max(np.abs(r["grad"][b].cpu().numpy() - o["grad"]) / np.maximum(np.abs(o["grad"]), 1e-5))))
    flag = "" if (err["means"] < 1e-5 and err["covs"] < 1e-4 and err["cost"] < 1e-5 and err["grad"] < 1e-3) else "   <-- CHECK"
    print(f"case {case:2d}: N={N:3d} ds={ds} da={da} H={H} B={B} gamma={gamma:g} " + " ".join(f"{k} {v:.1e}" for k, v in err.items()) + flag)
    for k in worst:
        worst[k] = max(worst[k], err[k])
print("worst:", {k: f"{v:.1e}" for k, v in worst.items()})
