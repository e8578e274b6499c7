import time, sys, numpy as np, torch
sys.path.insert(0, ".")
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
from oracle import gpmpc_oracle as O
for N in (256, 257, 320, 321, 1024, 1025):
    pb = synth_problem(3, N, 2, 1, 5, 1)
    kinv = torch.as_tensor(O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy(), device="cuda")
    X, Y = torch.as_tensor(pb["X"], device="cuda"), torch.as_tensor(pb["Y"], device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = g.GPPack(X, Y, kinv, pb["lambdas"], pb["sigma_f"]); torch.cuda.synchronize(); t1 = time.perf_counter()
    p.rebuild(X, Y, kinv, pb["lambdas"], pb["sigma_f"]); torch.cuda.synchronize(); t2 = time.perf_counter()
    del p; torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"N={N}: create+build {1e3*(t1-t0):.2f} ms, rebuild in place {1e3*(t2-t1):.2f} ms, destroy {1e3*(t3-t2):.2f} ms")
