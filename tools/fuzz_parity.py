#!/usr/bin/env python3
"""Randomised parity sweep of the HIP rollout against the CPU oracle over shapes, batch sizes, risk parameters and
kernel shapes (forced through the tuning overrides).  Not part of the test-suite (minutes of oracle time); run on the GPU box:
    python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
from oracle import gpmpc_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
worst = {"means": 0.0, "vars": 0.0, "cost": 0.0, "grad": 0.0}
for case in range(n_cases):
    N = int(rng.choice([1, 2, 17, 63, 64, 65, 100, 129, 200, 257, 320]))
    ds = int(rng.integers(1, 5)); da = int(rng.integers(1, 3)); H = int(rng.integers(1, 6)); B = int(rng.integers(1, 7))
    gamma = float(rng.choice([-1.0, 1e-5, 0.0, 0.5]))
    shape = rng.choice(["auto", "sb", "staged", "tb2"])
    for k in ("GPMPC_PAIR_SB", "GPMPC_PAIR_TB"):
        os.environ.pop(k, None)
    if shape == "sb":
        os.environ["GPMPC_PAIR_SB"] = "1"
    elif shape == "staged":
        os.environ["GPMPC_PAIR_SB"] = "0"
    elif shape == "tb2":
        os.environ["GPMPC_PAIR_SB"] = "1"; os.environ["GPMPC_PAIR_TB"] = "2"
    pb = synth_problem(100 + case, N, ds, da, H, B)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = g.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    r = g.rollout(pack, pb["x0"], pb["U"], g.CostParams(gamma, pb["Q"], pb["R"]))
    err = {k: 0.0 for k in worst}
    for b in range(B):
        o = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], gamma, mode="o2")
        rel = lambda a, c, fl: float(np.max(np.abs(a - c) / np.maximum(np.abs(c), fl)))   # noqa: E731
        err["means"] = max(err["means"], rel(r["means"][b].cpu().numpy(), o["means"], 1e-6))
        err["vars"] = max(err["vars"], rel(r["vars"][b].cpu().numpy(), o["vars"], 1e-12))
        err["cost"] = max(err["cost"], rel(np.array(r["cost"][b].item()), np.array(o["cost"]), 1e-9))
        err["grad"] = max(err["grad"], rel(r["grad"][b].cpu().numpy(), o["grad"], 1e-5))
    flag = "" if (err["means"] < 1e-5 and err["vars"] < 1e-4 and err["cost"] < 1e-5 and err["grad"] < 1e-3) else "   <-- CHECK"
    print(f"case {case:2d}: N={N:3d} ds={ds} da={da} H={H} B={B} gamma={gamma:g} shape={shape:6s} "
          + " ".join(f"{k} {v:.1e}" for k, v in err.items()) + flag)
    for k in worst:
        worst[k] = max(worst[k], err[k])
print("worst:", {k: f"{v:.1e}" for k, v in worst.items()})
