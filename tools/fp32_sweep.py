#!/usr/bin/env python3
"""fp64 vs fp32 tolerance sweep of BASELINE config 3 (N=2048, d=4, H=20 shapes): the HIP rollout in fp64, with fp32
accumulation of the N^2 sum, and with fp32 exponent / exp / sum, against the fp64 CPU oracle (O(N^2) mode), as a
function of the noise level sigma_n that sets the conditioning of the variance sum.  H is cut to 3 steps and 3
trajectories so that the oracle finishes in seconds; errors are max relative differences over those."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
from oracle import gpmpc_oracle as O

cfg = CONFIGS["C3"]
H, B = 3, 3
torch.set_num_threads(16)
print(f"# N={cfg['N']}, ds={cfg['ds']}, da={cfg['da']}, H={H} of 20, {B} trajectories, gamma=-1; max relative error vs the fp64 CPU oracle")
for sn in (0.1, 0.03, 0.01, 0.001):
    pb = synth_problem(3, cfg["N"], cfg["ds"], cfg["da"], H, B, sigma_n=sn)
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    pack = g.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
    cost = g.CostParams(-1.0, pb["Q"], pb["R"])
    ref_m, ref_v = [], []
    for b in range(B):
        m, c = O.forward_propagate(gp, H, pb["x0"][b], torch.as_tensor(pb["U"][b]), mode="o2")
        ref_m.append(torch.stack(m).numpy()); ref_v.append(torch.stack([torch.diagonal(x) for x in c]).numpy())
    ref_m, ref_v = np.stack(ref_m), np.stack(ref_v)
    row = [f"sigma_n={sn:<6g}"]
    for prec in ("fp64", "fp32acc", "fp32"):
        r = g.rollout(pack, pb["x0"], pb["U"], cost, want_grad=False, precision=prec)
        em = np.max(np.abs(r["means"].cpu().numpy() - ref_m) / np.maximum(np.abs(ref_m), 1e-300))
        ev = np.max(np.abs(r["vars"].cpu().numpy()[:, 1:] - ref_v[:, 1:]) / np.abs(ref_v[:, 1:]))
        row.append(f"{prec}: means {em:.1e} vars {ev:.1e}")
    print(" | ".join(row))
