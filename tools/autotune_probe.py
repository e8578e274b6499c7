#!/usr/bin/env python3
"""gpmpc_pack_autotune over a grid of call shapes: for each, the default plan's time, the best candidate's and every candidate's.
    python tools/autotune_probe.py [--graph] [--shared-lambda] N:ds:da:H:B ..."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.rollout import GPPack
from gaussian_process_mpc_amd.synth import synth_problem
ap = argparse.ArgumentParser()
ap.add_argument("--graph", action="store_true"); ap.add_argument("--shared-lambda", action="store_true"); ap.add_argument("--verbose", action="store_true")
ap.add_argument("shapes", nargs="+")
a = ap.parse_args()
g.require_gpu()
packs = {}
for shape in a.shapes:
    N, ds, da, H, B = (int(v) for v in shape.split(":"))
    key = (N, ds, da)
    if key not in packs:
        packs.clear(); torch.cuda.empty_cache()
        pb = synth_problem(3, N, ds, da, H, 8, shared_lambda=a.shared_lambda)
        kinv = []
        for k in range(ds):
            gp = g.GaussianProcessRegression(ds + da)
            gp.set_lambdas(pb["lambdas"][k]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][k]))
            gp.append_train_data(pb["X"], pb["Y"][:, k]); kinv.append(gp.Ky_inv)
        packs[key] = GPPack(pb["X"], pb["Y"], torch.stack(kinv), pb["lambdas"], pb["sigma_f"])
    pack = packs[key]
    pack.autotune_clear()
    default = pack.plan(B, H, graph=a.graph)
    res = pack.autotune(B, H, graph=a.graph)
    best = min((r for r in res if r["ms"] > 0), key=lambda r: r["ms"])
    d0 = res[0]
    print(f"{shape:>18s} default {d0['ms']:9.4f} ms  best {best['ms']:9.4f} ms ({best['name']}: fused={best['fused']} tiling={best['tiling']} sb={best['sb']} "
          f"tb={best['tb']} pwaves={best['pwaves']} xcdmap={best.get('xcdmap', -1)} split={best['split']})  default/best {d0['ms'] / best['ms']:.3f}   [{default['form']} {default['tiling']}]", flush=True)
    if a.verbose:
        for r in sorted(res, key=lambda r: r['ms'] if r['ms'] > 0 else 1e9)[:10]:
            print("      ", f"{r['ms']:9.4f}", r['name'], {k: v for k, v in r.items() if k not in ('ms', 'name')})
