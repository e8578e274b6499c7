#!/usr/bin/env python3
"""C3 batch (B = 256) as ONE call against TWO half batches on two streams (the head kernel of one half then overlaps the
pair kernel of the other).  Whole-job rollouts/s, inputs resident.  Run on the GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.rollout import CostParams, GPPack, rollout
from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
cid = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(CONFIGS[cid]); B = 128 if cid == "C4" else cfg["B"]
dev = g.require_gpu()
pb = synth_problem(int(cid[1]), cfg["N"], cfg["ds"], cfg["da"], cfg["H"], B)
kinv = []
for a in range(cfg["ds"]):
    gp = g.GaussianProcessRegression(cfg["ds"] + cfg["da"])
    gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][a]))
    gp.append_train_data(pb["X"], pb["Y"][:, a]); kinv.append(gp.Ky_inv)
pack = GPPack(torch.as_tensor(pb["X"], device=dev), torch.as_tensor(pb["Y"], device=dev), torch.stack(kinv), pb["lambdas"], pb["sigma_f"])
del kinv
cost = CostParams(cfg["gamma"], pb["Q"], pb["R"])
x0, U = torch.as_tensor(pb["x0"], device=dev), torch.as_tensor(pb["U"], device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
h = B // 2
def one():
    return [rollout(pack, x0, U, cost, want_traj=False)]
def two():
    out = []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            out.append(rollout(pack, x0[k * h:(k + 1) * h], U[k * h:(k + 1) * h], cost, want_traj=False))
    return out
ref = one()[0]; t2 = two(); torch.cuda.synchronize()
assert torch.equal(torch.cat([t["cost"] for t in t2]), ref["cost"]) or torch.allclose(torch.cat([t["cost"] for t in t2]), ref["cost"], rtol=1e-12)
for name, fn in (("one call", one), ("two streams", two), ("one call", one), ("two streams", two)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{cid} B={B} {name:12s} {dt * 1e3:8.3f} ms per batch  {B / dt:8.1f} rollouts/s")
