#!/usr/bin/env python3
"""A/B of one tuning variable (GPMPC_* environment variable, re-read by pack.reload_tuning()) over problem shapes, in ONE process
on ONE device: objective + gradient, each rollout replayed as one hipGraph, inputs resident, best of two interleaved rounds.
With --bitwise the gradients of every setting must equal those of the first one bit for bit.  Run on the GPU box:
    python tools/env_ab.py --var GPMPC_SB_UNROLL --values 1,4 [--bitwise] [N:ds:da:H:B ...]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.rollout import CostParams, GPPack, rollout
from gaussian_process_mpc_amd.synth import synth_problem

ap = argparse.ArgumentParser()
ap.add_argument("--var", required=True)
ap.add_argument("--values", required=True, help="comma separated; 'unset' removes the variable")
ap.add_argument("--bitwise", action="store_true")
ap.add_argument("--eager", action="store_true", help="plain launches instead of graph replay")
ap.add_argument("--shared-lambda", action="store_true")
ap.add_argument("--create-env", default="", help="NAME=VALUE[,NAME=VALUE] set while the packs are CREATED (variables read at gpmpc_pack_create)")
ap.add_argument("--sync-each", action="store_true", help="synchronise after every call (the latency a caller sees who needs each result)")
ap.add_argument("shapes", nargs="*")
args = ap.parse_args()
shapes = args.shapes or ["1024:4:1:20:8", "1024:4:1:20:12", "1024:4:1:20:16", "1024:4:1:20:24", "1024:4:1:20:32", "1024:4:1:20:64",
                         "2048:4:1:20:4", "2048:4:1:20:8", "2048:4:1:20:16", "2048:4:1:20:32", "512:3:1:20:64", "600:4:1:20:32",
                         "400:2:1:10:128", "4096:6:1:30:1", "4096:6:1:30:2"]
values = args.values.split(",")
for kv in filter(None, args.create_env.split(",")):
    os.environ[kv.split("=")[0]] = kv.split("=")[1]
dev = g.require_gpu()
packs = {}
for shape in shapes:
    N, ds, da, H, B = (int(v) for v in shape.split(":"))
    key = (N, ds, da)
    if key not in packs:
        packs.clear(); torch.cuda.empty_cache()
        pb = synth_problem(3, N, ds, da, H, 512, shared_lambda=args.shared_lambda)
        kinv = []
        for a in range(ds):
            gp = g.GaussianProcessRegression(ds + da)
            gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][a]))
            gp.append_train_data(pb["X"], pb["Y"][:, a]); kinv.append(gp.Ky_inv)
        packs[key] = (pb, GPPack(pb["X"], pb["Y"], torch.stack(kinv), pb["lambdas"], pb["sigma_f"]))
        del kinv
    pb, pack = packs[key]
    cost = CostParams(-1.0, pb["Q"], pb["R"])
    x0, U = torch.as_tensor(pb["x0"][:B], device=dev), torch.as_tensor(pb["U"][:B, :H], device=dev)
    ref, t = None, {}
    for v in values + values:
        if v == "unset":
            os.environ.pop(args.var, None)
        else:
            os.environ[args.var] = v
        pack.reload_tuning()
        run = lambda: rollout(pack, x0, U, cost, want_traj=False, graph=not args.eager)     # noqa: E731
        r = run(); torch.cuda.synchronize()
        gr = r["grad"].clone()
        ref = gr if ref is None else ref
        if args.bitwise:
            assert torch.equal(gr, ref), (shape, v)
        else:
            assert torch.allclose(gr, ref, rtol=1e-4, atol=1e-9), (shape, v)
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            run()
            if args.sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize(); t[v] = min(t.get(v, 1e9), (time.perf_counter() - t0) / reps)
    os.environ.pop(args.var, None)
    base = t[values[0]]
    print(f"N={N} ds={ds} H={H} B={B:3d}:" + "".join(f"   {args.var}={v}: {t[v] * 1e3:7.3f} ms ({B / t[v]:8.0f}/s) x{base / t[v]:.2f}" for v in values), flush=True)
