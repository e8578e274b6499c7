#!/usr/bin/env python3
"""Rollouts/s over (N, state_dim, batch size): where each kernel shape of plan_rollout (step.hip) sits against the
pair-evaluation rate of the large-batch kernel.  Objective + gradient, inputs resident, each rollout replayed as one
hipGraph up to B = 128.  `eff` = pairs/s relative to the pair rate measured at the largest batch of the same (N, ds).

    python tools/batch_map.py [--quick]            # optional GPMPC_TILING / GPMPC_PAIR_SB / GPMPC_FUSED overrides apply
"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.rollout import CostParams, GPPack, rollout, rollout_fullcov
from gaussian_process_mpc_amd.synth import synth_problem

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--fullcov", action="store_true", help="full-covariance rollout (gpmpc_rollout_fullcov; eager)")
ap.add_argument("--shapes", default="300:2:1:10,300:4:1:10,1024:4:1:20,2048:4:1:20,4096:6:1:30")
ap.add_argument("--batches", default="1,2,4,8,16,32,64,128,256")
args = ap.parse_args()
dev = g.require_gpu()

for shape in args.shapes.split(","):
    N, ds, da, H = (int(v) for v in shape.split(":"))
    bmax = max(int(b) for b in args.batches.split(","))
    pb = synth_problem(3, N, ds, da, H, bmax)
    X = torch.as_tensor(pb["X"], device=dev)
    Y = torch.as_tensor(pb["Y"], device=dev)
    kinv = []
    for a in range(ds):
        gp = g.GaussianProcessRegression(ds + da)
        gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][a]))
        gp.append_train_data(pb["X"], pb["Y"][:, a])
        kinv.append(gp.Ky_inv)
    pack = GPPack(X, Y, torch.stack(kinv), pb["lambdas"], pb["sigma_f"])
    del kinv
    cost = CostParams(-1.0, pb["Q"], pb["R"], x_ref=pb["x_ref"], u_ref=pb["u_ref"])
    rows = []
    for B in (int(b) for b in args.batches.split(",")):
        if N >= 4096 and B > 128:
            continue
        x0 = torch.as_tensor(pb["x0"][:B], device=dev)
        U = torch.as_tensor(pb["U"][:B], device=dev)
        graph = B <= 128 and not args.fullcov
        run = (lambda: rollout_fullcov(pack, x0, U, cost)) if args.fullcov else (lambda: rollout(pack, x0, U, cost, want_grad=True, want_traj=False, graph=graph))
        reps = max(10, min(200, int((1e9 if args.quick else 4e9) / (B * H * (ds * ds if args.fullcov else ds) * N * N / 2 * 30))))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        dt = float("inf")
        for _ in range(3):                                   # best of three blocks: the eager path is host-launch sensitive
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / reps)
        rows.append((B, dt, B / dt))
    pairs = H * (ds * N * (N + 1) / 2 + (ds * (ds - 1) / 2 * N * N if args.fullcov else 0))
    best = max(r[2] for r in rows)
    print(f"N={N} ds={ds} da={da} H={H}  (pair-evaluations per rollout {pairs:.3g})")
    for B, dt, rate in rows:
        print(f"   B={B:4d}  {dt * 1e3:9.3f} ms per batch  {rate:10.1f} rollouts/s  {rate * pairs / 1e12:7.3f} Tpairs/s  eff {rate / best:5.2f}")
    del pack
    torch.cuda.empty_cache()
