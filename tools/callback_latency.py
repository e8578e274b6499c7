#!/usr/bin/env python3
"""Wall-clock latency of the solver callbacks (RiskSensitiveMPC.objective + gradient, B = 1) on the BASELINE small configs:
what Ipopt / L-BFGS sees per iteration, Python and synchronisation included."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem

first = True
for cid in ("C1", "C2", "C1", "C2"):          # (the first config of a fresh process also pays the device's clock ramp)
    cfg = CONFIGS[cid]
    pb = synth_problem(int(cid[1]), cfg["N"], cfg["ds"], cfg["da"], cfg["H"], 1)
    mpc = g.RiskSensitiveMPC(cfg["gamma"], cfg["H"], cfg["ds"], cfg["da"], pb["Q"], pb["R"])
    for a, gp in enumerate(mpc.dynamics.gpr_err):
        gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(pb["sigma_f"][a])); gp.set_sigma_n(np.array(pb["sigma_n"][a]))
    S, A = pb["X"][:, :cfg["ds"]], pb["X"][:, cfg["ds"]:]
    mpc.dynamics.append_train_data(S, A, pb["Y"])
    mpc.curr_state = torch.as_tensor(pb["x0"][0], device=mpc.device)
    rng = np.random.default_rng(0)
    xs = [rng.uniform(-1, 1, cfg["H"] * cfg["da"]) for _ in range(220)]
    for x in xs[:20]:
        mpc.objective(x); mpc.gradient(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in xs[20:]:
        mpc.objective(x); mpc.gradient(x)
    dt = (time.perf_counter() - t0) / 200
    print(f"{cid}: objective+gradient callback pair {dt * 1e3:.3f} ms wall-clock ({1 / dt:.0f} per second)"
          + ("   [first config of a fresh process: includes one-time warm-up of the runtime]" if first else ""))
    first = False
