#!/usr/bin/env python3
"""A/B of the forms of the full-covariance rollout (fullcov.hip) at small batches: ms per rollout call and agreement of cost / gradient
with the four-launch form.

    python tools/fullcov_ab.py [--shapes 2048:4:1:20,...] [--batches 1,2,4,8]
"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.rollout import CostParams, GPPack, rollout_fullcov
from gaussian_process_mpc_amd.synth import synth_problem

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="2048:4:1:20,1024:4:1:20,512:3:1:20,300:4:1:10")
ap.add_argument("--batches", default="1,2,4,8,16")
ap.add_argument("--forms", default="old:0:-1,two64:1:2,two128:1:4,two256:1:0,default:-1:-1",
                help="name:GPMPC_FC_FORM:GPMPC_FC_TILING[:GPMPC_FC_CU[:GPMPC_FC_RSPLIT[:GPMPC_FC_SHARED]]] (-1: unset)")
ap.add_argument("--shared-lambda", action="store_true", help="one lambda for all GPs (the reference's experiments)")
args = ap.parse_args()
dev = g.require_gpu()
forms = [f.split(":") for f in args.forms.split(",")]

for shape in args.shapes.split(","):
    N, ds, da, H = (int(v) for v in shape.split(":"))
    bmax = max(int(b) for b in args.batches.split(","))
    pb = synth_problem(3, N, ds, da, H, bmax, shared_lambda=args.shared_lambda)
    X = torch.as_tensor(pb["X"], device=dev)
    Y = torch.as_tensor(pb["Y"], device=dev)
    kinv = []
    for a in range(ds):
        gp = g.GaussianProcessRegression(ds + da)
        gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][a]))
        gp.append_train_data(pb["X"], pb["Y"][:, a])
        kinv.append(gp.Ky_inv)
    pack = GPPack(X, Y, torch.stack(kinv), pb["lambdas"], pb["sigma_f"]).enable_fullcov()
    del kinv
    cost = CostParams(-1.0, pb["Q"], pb["R"], x_ref=pb["x_ref"], u_ref=pb["u_ref"])
    for B in (int(b) for b in args.batches.split(",")):
        x0 = torch.as_tensor(pb["x0"][:B], device=dev)
        U = torch.as_tensor(pb["U"][:B], device=dev)
        line, ref = f"{shape:>14s} B={B:<3d}", None
        for spec in forms:
            name, form, tiling = spec[:3]
            cu = spec[3] if len(spec) > 3 else "-1"
            rsp = spec[4] if len(spec) > 4 else "-1"
            fsh = spec[5] if len(spec) > 5 else "-1"
            for k, v in (("GPMPC_FC_FORM", form), ("GPMPC_FC_TILING", tiling), ("GPMPC_FC_CU", cu), ("GPMPC_FC_RSPLIT", rsp), ("GPMPC_FC_SHARED", fsh)):
                if v == "-1":
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            pack.reload_tuning()
            r = rollout_fullcov(pack, x0, U, cost)
            torch.cuda.synchronize()
            reps = max(5, min(100, int(2e9 / (B * H * ds * ds * N * N / 2 * 30))))
            dt = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps):
                    rollout_fullcov(pack, x0, U, cost)
                torch.cuda.synchronize()
                dt = min(dt, (time.perf_counter() - t0) / reps)
            if ref is None:
                ref = r
                err = 0.0
            else:
                err = max(float((r[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ("cost", "grad", "covs"))
            line += f"  {name}: {dt * 1e3:8.3f} ms ~{err:.0e}"
        print(line, flush=True)
    for k in ("GPMPC_FC_FORM", "GPMPC_FC_TILING", "GPMPC_FC_CU", "GPMPC_FC_RSPLIT", "GPMPC_FC_SHARED"):
        os.environ.pop(k, None)
