import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
from oracle import gpmpc_oracle as O, cport
ds, da, N, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 3, 7
pb = synth_problem(40 + 8 * ds + da, N, ds, da, H, B)
kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
pack = g.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])
cost = g.CostParams(-1.0, pb["Q"], pb["R"])
print(pack.plan(B, H))
c = cport.rollout(pb, kinv, -1.0, x0=pb["x0"][:B], U=pb["U"][:B], nthreads=8)
for grad in (True, False):
    rs = [g.rollout(pack, pb["x0"][:B], pb["U"][:B], cost, want_grad=grad) for _ in range(3)]
    m = [r["means"].cpu().numpy() for r in rs]; v = [r["vars"].cpu().numpy() for r in rs]
    print("grad", grad, "repeatable:", np.array_equal(m[0], m[1]) and np.array_equal(m[0], m[2]) and np.array_equal(v[0], v[1]),
          "max mean rel err", float((np.abs(m[0] - c["means"]) / (np.abs(c["means"]) + 1e-9)).max()),
          "max var rel err", float((np.abs(v[0] - c["vars"]) / (np.abs(c["vars"]) + 1e-12)).max()),
          "t=1 per GP mean err", (np.abs(m[0][:, 1] - c["means"][:, 1]) / (np.abs(c["means"][:, 1]) + 1e-9)).max(0))
