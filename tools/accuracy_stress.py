#!/usr/bin/env python3
"""Noise-level sweep of HIP-vs-oracle agreement (the cancelling N^2 sum gets harder as sigma_n shrinks: the experiments
of the reference use sigma_n = 1e-5).  N=512, ds=3, da=1, H=5, 2 trajectories.  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaussian_process_mpc_amd as G
from oracle import gpmpc_oracle as O
from gaussian_process_mpc_amd.synth import synth_problem
torch.set_num_threads(16)
for sn in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
    H = 5
    pb = synth_problem(2, 512, 3, 1, H, 2, sigma_n=sn)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = G.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(-1.0, pb["Q"], pb["R"]))
    em = ev = ef = 0.0
    for b in range(2):
        o = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0, mode="o2")
        f = O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0, mode="faithful", want_grad=False)
        em = max(em, np.abs(r["means"][b].cpu().numpy() - o["means"]).max() / np.abs(o["means"]).max())
        ev = max(ev, np.abs(r["vars"][b].cpu().numpy() / o["vars"] - 1).max())
        ef = max(ef, np.abs(f["vars"] / o["vars"] - 1).max())
    print(f"sigma_n={sn:g}: HIP vs oracle(O(N^2)): means {em:.1e} vars {ev:.1e} | oracle faithful (N^3 trace) vs oracle O(N^2): vars {ef:.1e}")
