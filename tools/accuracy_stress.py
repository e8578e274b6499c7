#!/usr/bin/env python3
"""Noise-level sweep of the variance accuracy (the cancelling N^2 sum gets harder as sigma_n shrinks: the experiments of the
reference use sigma_n = 1e-5).  N = 512, ds = 3, da = 1, H = 5, 2 trajectories.  Yardstick: the same rollout with every
operation in x87 extended precision on the same fp64 inputs (oracle/cport/gpmpc_cpu_ld.c).  Columns: max relative deviation
of the propagated variances from the yardstick for (1) the HIP path, (2) the reference's own op order in fp64 (oracle,
faithful mode: N^3 trace, src/tools/uncertainty_prop.py:399), (3) the same trace as an elementwise fp64 sum (oracle, O(N^2)
mode), (4) the plain-C fp64 checker.  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaussian_process_mpc_amd as G
from oracle import cport, gpmpc_oracle as O
from gaussian_process_mpc_amd.synth import synth_problem
torch.set_num_threads(16)
dev = lambda a, e: float(np.abs(a[:, 1:] / e[:, 1:] - 1).max())      # noqa: E731
for sn in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
    H = 5
    pb = synth_problem(2, 512, 3, 1, H, 2, sigma_n=sn)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    kinv = gp.Ky_inv.numpy()
    pack = G.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
    r = G.rollout(pack, pb["x0"], pb["U"], G.CostParams(-1.0, pb["Q"], pb["R"]), want_grad=False)
    e = cport.rollout_extended(pb, kinv, nthreads=16)
    c = cport.rollout(pb, kinv, -1.0, nthreads=16)
    o2 = np.stack([O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0, mode="o2", want_grad=False)["vars"] for b in range(2)])
    fa = np.stack([O.objective_and_gradient(gp, H, pb["x0"][b], pb["U"][b], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0, mode="faithful", want_grad=False)["vars"] for b in range(2)])
    em = float(np.abs(r["means"].cpu().numpy() - e["means"]).max() / np.abs(e["means"]).max())
    print(f"sigma_n={sn:<7g} variances vs the extended-precision yardstick: HIP {dev(r['vars'].cpu().numpy(), e['vars']):.1e} | "
          f"reference op order (fp64) {dev(fa, e['vars']):.1e} | elementwise fp64 sum {dev(o2, e['vars']):.1e} | C checker {dev(c['vars'], e['vars']):.1e}"
          f"   (HIP means {em:.1e})")
