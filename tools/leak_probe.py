#!/usr/bin/env python3
"""Device-memory probe: 600 cycles of (new hyper-parameters -> new Ky_inv -> new pack -> solver callbacks, a batched call, a
graph capture and replay) at constant N; the device's free memory must not drift (packs, captured graphs and pinned staging
buffers are released with their pack).  Run on the GPU box: python tools/leak_probe.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
pb = synth_problem(3, 200, 2, 1, 5, 1)
mpc = g.RiskSensitiveMPC(1e-5, 5, 2, 1, pb["Q"], pb["R"])
for a, gp in enumerate(mpc.dynamics.gpr_err):
    gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(0.01))
mpc.dynamics.append_train_data(pb["X"][:150, :2], pb["X"][:150, 2:], pb["Y"][:150])
mpc.curr_state = torch.as_tensor(pb["x0"][0], device=mpc.device)
rng = np.random.default_rng(0)
def cycle(i):
    for a, gp in enumerate(mpc.dynamics.gpr_err):          # new hyper-parameters -> new Ky_inv -> new pack, same N
        gp.set_lambdas(pb["lambdas"][a] * (1.0 + 1e-3 * (i % 7)))
        gp.build_Ky_inv_mat()
    for _ in range(3):
        x = rng.uniform(-1, 1, 5)
        mpc.objective(x); mpc.gradient(x)
    r = mpc.evaluate_batch(rng.uniform(-1, 1, (4, 5, 1)))
    g.rollout(mpc.dynamics.pack(), pb["x0"][0], pb["U"][0], mpc._cost_params(), graph=True)
for i in range(30): cycle(i)
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]; a0 = torch.cuda.memory_allocated()
t0 = time.time()
for i in range(30, 630): cycle(i)
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]; a1 = torch.cuda.memory_allocated()
print("600 cycles in %.1f s; device free memory change %.2f MiB; torch allocated change %.2f MiB; N now %d" % (time.time() - t0, (free1 - free0) / 2**20, (a1 - a0) / 2**20, mpc.dynamics.gpr_err[0].num_train))
