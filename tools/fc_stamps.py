#!/usr/bin/env python3
"""Phase timeline of the head kernel of the two-launch full-covariance rollout (csrc/fullcov.hip::k_fc_head), trajectory 0, horizon
step 3, the three kinds of workgroups of the first variance unit; diagnostic build:
    make -C gaussian_process_mpc_amd/csrc BUILD=build_fst LIB=libgpmpc_hip_fst.so EXTRA="-DGPMPC_FC_STAMPS"
    GPMPC_LIB_PATH=.../libgpmpc_hip_fst.so GPMPC_FC_FORM=1 python tools/fc_stamps.py 2048:4:1:20:1"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
N, ds, da, H, B = (int(v) for v in sys.argv[1].split(":"))
pb = synth_problem(3, N, ds, da, H, B)
dev = g.require_gpu()
kinv = []
for a in range(ds):
    gp = g.GaussianProcessRegression(ds + da)
    gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][a]))
    gp.append_train_data(pb["X"], pb["Y"][:, a])
    kinv.append(gp.Ky_inv)
pack = g.GPPack(pb["X"], pb["Y"], torch.stack(kinv), pb["lambdas"], pb["sigma_f"]).enable_fullcov()
cost = g.CostParams(-1.0, pb["Q"], pb["R"])
for _ in range(5):
    g.rollout_fullcov(pack, pb["x0"], pb["U"], cost)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L = ctypes.CDLL(os.environ["GPMPC_LIB_PATH"])
assert L.gpmpc_debug_fc_stamps(buf) == 0
st = np.array(list(buf), dtype=np.int64)
names = {0: "start", 1: "Z0 partial sums (all units)", 2: "Z0 combined", 3: "own-unit moment sums", 4: "moments + assemble", 5: "closing algebra of the own unit (Jacobians)", 8: "body start",
         9: "inverses (variance unit)", 10: "Cholesky", 11: "set-up stores + row loop", 12: "block sum", 13: "mean Jacobians | cross set-up",
         14: "column rows (G)"}
for base, title in ((0, "variance unit 0: column-row workgroup 0"), (16, "variance unit 0: mean workgroup"), (32, "variance unit 0: closing workgroup")):
    print(title)
    prev = st[base]
    for k in sorted(names):
        v = st[base + k]
        if v == 0:
            continue
        print(f"  {names[k]:34s} +{v - prev:7d}   (t = {v - st[base]:7d})")
        prev = v
