#!/usr/bin/env python3
"""Shared-lambda pair kernel: GPs per workgroup (GPMPC_SHARED_NG, read at pack creation) at C3 sizes and at the pendulum /
cart-pole shapes; distinct-lambda kernel on the same data for reference (GPMPC_SHARED=0).  Run on the GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.rollout import CostParams, GPPack, rollout
from gaussian_process_mpc_amd.synth import synth_problem
dev = g.require_gpu()
for shape in (sys.argv[1:] or ["2048:4:1:20:256", "2048:4:1:20:32", "1024:4:1:20:64", "600:4:1:20:128", "4096:6:1:30:64", "400:2:1:10:512", "512:3:1:20:256"]):
    N, ds, da, H, B = (int(v) for v in shape.split(":"))
    pb = synth_problem(3, N, ds, da, H, B, shared_lambda=True)
    kinv = []
    gp = g.GaussianProcessRegression(ds + da)
    gp.set_lambdas(pb["lambdas"][0]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][0]))
    gp.append_train_data(pb["X"], pb["Y"][:, 0])
    kinv = gp.Ky_inv.unsqueeze(0).expand(ds, N, N).contiguous()
    cost = CostParams(-1.0, pb["Q"], pb["R"])
    x0, U = torch.as_tensor(pb["x0"], device=dev), torch.as_tensor(pb["U"], device=dev)
    line = f"N={N} ds={ds} H={H} B={B:3d}:"
    for tag, env in (("distinct-lambda kernel", {"GPMPC_SHARED": "0"}), ("NG=2", {"GPMPC_SHARED_NG": "2"}), ("NG=3", {"GPMPC_SHARED_NG": "3"}),
                     ("NG=4", {"GPMPC_SHARED_NG": "4"}), ("default", {})):
        if tag.startswith("NG=") and int(tag[3:]) > ds:
            continue
        for k in ("GPMPC_SHARED", "GPMPC_SHARED_NG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        pack = GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
        run = lambda: rollout(pack, x0, U, cost, want_traj=False, graph=B < 64)     # noqa: E731
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5 if N >= 2048 else 20
        for _ in range(reps): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        line += f"   {tag}: {dt * 1e3:8.3f} ms ({B / dt:8.0f}/s)"
        del pack
    print(line, flush=True)
