#!/usr/bin/env python3
"""Accuracy of the HIP rollout against the CPU oracle at C3 size (N=2048, ds=4, da=1), first H steps of 2 trajectories,
for the scalar-broadcast (expanded exponent) and the staged (direct exponent) pair kernels.  Run on the GPU box."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    import gaussian_process_mpc_amd as G
    from oracle import gpmpc_oracle as O
    from gaussian_process_mpc_amd.synth import synth_problem
    H = 3
    sn = float(sys.argv[2])
    pb = synth_problem(3, 2048, 4, 1, H, 2, sigma_n=sn)
    torch.set_num_threads(16)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(-1.0, pb["Q"], pb["R"]))
    em = ev = eg = 0.0
    for b in range(2):
        o = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0, mode="o2")
        em = max(em, np.abs(r["means"][b].cpu().numpy() / o["means"] - 1).max())
        ev = max(ev, np.abs(r["vars"][b].cpu().numpy() / o["vars"] - 1).max())
        eg = max(eg, np.abs(r["grad"][b].cpu().numpy() - o["grad"]).max() / np.abs(o["grad"]).max())
    print(f"{sys.argv[1]:8s} sigma_n={sn:g}: max rel err means {em:.2e} vars {ev:.2e} grad {eg:.2e}")
else:
    for sn in ("1e-2", "1e-3"):
        for name, env in (("sb", {"GPMPC_PAIR_SB": "1"}), ("staged", {"GPMPC_PAIR_SB": "0"})):
            subprocess.run([sys.executable, __file__, name, sn], env={**os.environ, **env})
