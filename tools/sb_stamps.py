#!/usr/bin/env python3
"""Per-workgroup timeline of ONE scalar-broadcast pair launch at a mid-size batch (diagnostic build):
    make -C gaussian_process_mpc_amd/csrc clean && make -C gaussian_process_mpc_amd/csrc -j8 EXTRA=-DGPMPC_SB_STAMPS
    python tools/sb_stamps.py [N B]      (ds = 4, da = 1: the D = 5 instances carry the stamps)
Prints when workgroups START relative to the first one (dispatch ramp), how long prologue / column loop / reduction take for
early and late workgroups, and how the launch's wall time splits."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
from oracle import gpmpc_oracle as O
N, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 16)
pb = synth_problem(3, N, 4, 1, 6, B)
gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
pack = g.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
cost = g.CostParams(-1.0, pb["Q"], pb["R"])
for _ in range(5):
    g.rollout(pack, pb["x0"], pb["U"], cost, want_traj=False)
torch.cuda.synchronize()
NS, NW = 8, 8192
buf = (ctypes.c_ulonglong * (NS * NW))()
fn = g.lib().gpmpc_debug_sb_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, NS * NW) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(NW, NS).astype(np.int64)
used = st[:, 0] > 0
st = st[used]
n = len(st)
t0 = st[:, 0].min()
start_us = (st[:, 0] - t0) / 100.0                 # s_memrealtime: 100 MHz
end_us = (st[:, 5] - t0) / 100.0
pro, loop, red = st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3]
hw = st[:, 6]
xcc = (hw >> 20) & 0xf if False else None
print(f"N={N} B={B}: {n} workgroups stamped; launch span {end_us.max():.1f} us (first start -> last end)")
q = lambda a: "  ".join(f"p{p}={np.percentile(a, p):9.1f}" for p in (0, 10, 50, 90, 100))       # noqa: E731
print("start offset [us]  ", q(start_us))
print("end offset   [us]  ", q(end_us))
print("prologue  [cycles] ", q(pro))
print("col loop  [cycles] ", q(loop))
print("reduction [cycles] ", q(red))
order = np.argsort(start_us)
for name, sel in (("first 25 % to start", order[: n // 4]), ("last 25 % to start", order[-(n // 4):])):
    print(f"{name:22s} start {start_us[sel].mean():6.1f} us  prologue {pro[sel].mean():8.0f}  loop {loop[sel].mean():8.0f}  reduction {red[sel].mean():7.0f} cycles  "
          f"lifetime {(end_us[sel] - start_us[sel]).mean():6.1f} us")
hist, edges = np.histogram(start_us, bins=12)
print("start-time histogram [us]:", " ".join(f"{edges[i]:.0f}-{edges[i+1]:.0f}:{hist[i]}" for i in range(len(hist))))
active = [(np.sum((start_us <= t) & (end_us > t))) for t in np.linspace(0, end_us.max(), 13)]
print("resident workgroups over the launch:", " ".join(str(a) for a in active))

try:
    hb = (ctypes.c_ulonglong * 16)()
    fh = g.lib().gpmpc_debug_head_stamps
    fh.argtypes = [ctypes.c_void_p]
    if fh(hb) == 0:
        h = np.array(list(hb), dtype=np.int64)
        names = ["finish_step (reduce the partials of step t-1, Jacobian rows)", "per-dimension scalars", "row loop (mean sums, G rows)", "block sum", "publish scalars"]
        print("head kernel, workgroup (0, 0, 0) of step 5 [cycles]: " + " | ".join(f"{n} {h[k + 1] - h[k]}" for k, n in enumerate(names)) + f" | total {h[5] - h[0]}")
except AttributeError:
    pass
