#!/usr/bin/env python3
"""Phase timeline of the whole-horizon kernel (csrc/traj_persist.h), workgroup 0, horizon step 3; diagnostic build:
    make -C gaussian_process_mpc_amd/csrc BUILD=build_pst LIB=libgpmpc_hip_pst.so EXTRA="-DGPMPC_PERSIST_STAMPS -DGPMPC_STAMP_D=5"
    GPMPC_LIB_PATH=.../libgpmpc_hip_pst.so GPMPC_PERSIST=16 python tools/persist_stamps.py 300:4:1:10:256"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
from oracle import gpmpc_oracle as O
N, ds, da, H, B = (int(v) for v in sys.argv[1].split(":"))
pb = synth_problem(3, N, ds, da, H, B)
gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
pack = g.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
cost = g.CostParams(-1.0, pb["Q"], pb["R"])
print(pack.plan(B, H))
for _ in range(10):
    g.rollout(pack, pb["x0"], pb["U"], cost, want_traj=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L = ctypes.CDLL(os.environ["GPMPC_LIB_PATH"])
assert L.gpmpc_debug_persist_stamps(buf) == 0
st = np.array(list(buf), dtype=np.int64)
names = ["step start", "scalars (a,k)", "dets; G rows issued", "mean sums done", "G stores drained", "barrier", "mean combine", "loop + flush (wave 0)", "barrier",
         "combine + outputs", "barrier"]
for k in range(1, 11):
    print(f"  {names[k]:26s} +{st[k] - st[k - 1]:7d} cycles   (t = {st[k] - st[0]:7d})")
nw = 16
a, b_, c = st[16:16 + nw], st[32:32 + nw], st[48:48 + nw]
ok = a > 0
print("column loops of the waves (cycles):", (b_ - a)[ok].tolist())
print("flush (cycles):", (c - b_)[ok].tolist())
print("loop start skew:", (a[ok] - a[ok].min()).tolist())
