#!/usr/bin/env python3
"""Phase timeline of the fused small-batch step kernel (diagnostic build, see csrc/step_fused.h):
    make -C gaussian_process_mpc_amd/csrc clean && make -C gaussian_process_mpc_amd/csrc -j8 EXTRA=-DGPMPC_FUSED_STAMPS
    python tools/fused_stamps.py        (C2 sizes: D = 4)
    python tools/fused_stamps.py 1024:3:1:20:16      (any shape with D = 4; mid-size ones run the 256x64 scalar-broadcast form)
A diagnostic library built beside the product (make BUILD=build_fst LIB=libgpmpc_hip_fst.so EXTRA=-DGPMPC_FUSED_STAMPS) is loaded
with GPMPC_LIB_PATH."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
from oracle import gpmpc_oracle as O
cfg = dict(CONFIGS["C2"])
if len(sys.argv) > 1:
    cfg["N"], cfg["ds"], cfg["da"], cfg["H"], cfg["B"] = (int(v) for v in sys.argv[1].split(":"))
import os
assert cfg["ds"] + cfg["da"] == int(os.environ.get("GPMPC_STAMP_D", "4")), "the stamps are compiled into the D = GPMPC_STAMP_D instances (default 4; build with -DGPMPC_STAMP_D=<D>)"
pb = synth_problem(2, cfg["N"], cfg["ds"], cfg["da"], cfg["H"], cfg["B"])
gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
pack = g.GPPack(pb["X"], pb["Y"], gp.Ky_inv.numpy(), pb["lambdas"], pb["sigma_f"])
cost = g.CostParams(-1.0, pb["Q"], pb["R"])
for _ in range(50):
    g.rollout(pack, pb["x0"], pb["U"], cost, want_traj=False, graph=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert g.lib().gpmpc_debug_stamps(buf) == 0
st = np.array(list(buf), dtype=np.int64)
names_tile = ["start", "loads issued", "z0 reduced", "scalars published", "chunk staged", "columns done", "tile reduced", "written"]
names_mean = ["start", "loads issued", "z0 reduced", "scalars published", "B/A published", "N loop done", "block sum done", "sp written"]
names_fin = ["start", "loads issued", "z0 reduced", "scalars published", "moments summed", "combined", "rows written", "-"]
for off, names, tag in ((0, names_tile, "tile workgroup 0"), (16, names_mean, "mean-sum workgroup of GP 0"), (32, names_fin, "finish workgroup of GP 0")):
    t = st[off:off + 8]
    print(tag)
    for k in range(1, 7 if off == 32 else 8):
        print(f"  {names[k]:22s} +{t[k] - t[k-1]:6d} cycles   (t = {t[k] - t[0]:6d})")
if st[13] > st[12]:
    ns = (st[13] - st[12]) * 10.0
    print("tile workgroup 0: %.2f us by the 100 MHz counter, %d shader-clock ticks -> %.2f GHz" % (ns / 1e3, st[7] - st[0], (st[7] - st[0]) / ns))
print("probes (tile wg): lam load %d cycles, partz load %d cycles, M load %d cycles" % (st[8] - st[0], st[9] - st[8], st[10] - st[9]))
print("probes (mean wg): lam load %d cycles, partz load %d cycles, M load %d cycles" % (st[24] - st[16], st[25] - st[24], st[26] - st[25]))

# start / end of EVERY workgroup of trajectory 0 in the stamped launch (100 MHz counter): where the launch's time goes
wg = (ctypes.c_ulonglong * (2 * 8192))()
if hasattr(g.lib(), "gpmpc_debug_wg_times") and g.lib().gpmpc_debug_wg_times(wg) == 0:
    a = np.array(list(wg), dtype=np.int64).reshape(2, 8192)
    used = np.nonzero(a[1] > 0)[0]
    if len(used):
        t0 = a[0][used].min()
        start, end = (a[0][used] - t0) / 100.0, (a[1][used] - t0) / 100.0
        ds = cfg["ds"]
        ntile = len(used) - 2 * ds
        print(f"{len(used)} workgroups of trajectory 0 ({ntile} tiles + {ds} mean-sum + {ds} finish); launch span {end.max():.1f} us")
        for name, sel in (("tiles", slice(0, ntile)), ("mean sums", slice(ntile, ntile + ds)), ("finish", slice(ntile + ds, ntile + 2 * ds))):
            st_, en_ = start[sel], end[sel]
            print(f"  {name:10s} start {st_.min():6.1f} .. {st_.max():6.1f} us   end p50 {np.median(en_):6.1f}  p90 {np.percentile(en_, 90):6.1f}  max {en_.max():6.1f} us   lifetime p50 {np.median(en_ - st_):6.1f} max {(en_ - st_).max():6.1f} us")
