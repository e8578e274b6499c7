#!/usr/bin/env python3
"""GP-MPC rollout benchmark (BASELINE.json metric: rollouts/sec).

One *rollout* = for one action sequence U (H x da): H-step moment-matching propagation of all ds
GPs (means + variances) + risk-sensitive cost + gradient w.r.t. U, i.e. one objective+gradient
callback pair of the reference (src/mpc.py:202-255).  One bench *step* = one batched call of the hot
path over B trajectories per GPU, inputs already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W] [--config C3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Multi-GPU = weak scaling: every rank owns B trajectories of a
global batch of N*B (independent candidates; the GP pack is replicated, broadcast from rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector peak = fp64 MFMA peak (spec)
HBM_PEAK_GBS = 8000.0


def pair_flops(D):
    """Algorithmic work per pair-evaluation, forward+gradient, SURVEY.md 8(d) convention:
    2D+23 fp64 issue slots = 4D+37 flops (FMA = 2)."""
    return 4 * D + 37, 2 * D + 23


def build_kinv(g, pb, device):
    """Ky_inv of every GP on the device: Kf/Ky by the HIP kernel (src/gpr.py:163-170), inverse by
    torch.linalg.inv as the reference (src/gpr.py:171)."""
    import ctypes
    from gaussian_process_mpc_amd._lib import lib, ptr, stream_ptr, host_doubles, check
    X = torch.as_tensor(pb["X"], device=device)
    N, D = X.shape
    out = torch.empty((pb["ds"], N, N), dtype=torch.float64, device=device)
    Ky = torch.empty((N, N), dtype=torch.float64, device=device)
    for a in range(pb["ds"]):
        _, lp = host_doubles(pb["lambdas"][a])
        noise = float(np.float32(pb["sigma_n"][a] ** 2))          # src/gpr.py:170 adds a float32 diagonal
        check(lib().gpmpc_build_ky(N, D, ptr(X), lp, float(pb["sigma_f"][a]), noise, None, ptr(Ky), stream_ptr()),
              "gpmpc_build_ky")
        out[a] = torch.linalg.inv(Ky)
    return out


def cpu_baseline(pb, H_sample, reps=2, fullcov=False):
    """The reference CPU path (faithful-op restatement in oracle/, checked against the reference by
    tests/test_oracle_golden.py) timed on this box's host cores on a bounded sample: ONE trajectory,
    H_sample of the H steps, objective + gradient; scaled linearly to H steps (per-step cost is constant)."""
    from oracle import gpmpc_oracle as O
    H = pb["H"]
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])
    res = {}
    for mode in ("faithful", "o2"):
        best = float("inf")
        for r in range(reps + 1):
            t0 = time.perf_counter()
            fn = O.objective_and_gradient_fullcov if fullcov else O.objective_and_gradient
            fn(gp, H_sample, pb["x0"][0], pb["U"][0][:H_sample], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], -1.0, mode=mode)
            dt = time.perf_counter() - t0
            if r > 0:
                best = min(best, dt)
        res[mode] = 1.0 / (best * H / H_sample)
    if not fullcov:
        # plain-C / OpenMP port of the O(N^2) algorithm (oracle/cport): whole horizon, 2 trajectories
        from oracle import cport
        nthr = torch.get_num_threads()
        cport.rollout(pb, gp.Ky_inv.numpy(), -1.0, x0=pb["x0"][:1], U=pb["U"][:1, :2], nthreads=nthr)     # warm-up / build
        t0 = time.perf_counter()
        cport.rollout(pb, gp.Ky_inv.numpy(), -1.0, x0=pb["x0"][:2], U=pb["U"][:2], nthreads=nthr)
        res["cport"] = 2.0 / (time.perf_counter() - t0)
        t0 = time.perf_counter()                                        # the same port on ONE core (1 thread and all cores)
        cport.rollout(pb, gp.Ky_inv.numpy(), -1.0, x0=pb["x0"][:1], U=pb["U"][:1], nthreads=1)
        res["cport_1thread"] = 1.0 / (time.perf_counter() - t0)
    return res, gp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (default: the config's B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay each rollout as one hipGraph (small batches)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % max(torch.cuda.device_count(), 1)      # rehearsals may put several ranks on one card
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    import gaussian_process_mpc_amd as g
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    from gaussian_process_mpc_amd._lib import lib
    from gaussian_process_mpc_amd.parallel import shard_range, gather_results

    cfg = dict(CONFIGS[args.config])
    B = args.batch or cfg["B"]
    if args.config == "C4" and not args.batch:
        B = cfg["B"] // 8                                  # 1024 trajectories over 8 GPUs
    cid = int(args.config[1])
    pb = synth_problem(cid, cfg["N"], cfg["ds"], cfg["da"], cfg["H"], B * world)
    N, ds, da, H, D = cfg["N"], cfg["ds"], cfg["da"], cfg["H"], cfg["ds"] + cfg["da"]

    # GP pack: rank 0 inverts, everyone receives the same bits
    if rank == 0:
        kinv = build_kinv(g, pb, device)
    else:
        kinv = torch.empty((ds, N, N), dtype=torch.float64, device=device)
    if world > 1:
        dist.broadcast(kinv, src=0)
    t0 = time.perf_counter()
    pack = g.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"], device=device)
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) * 1e3
    del kinv

    lo, hi = shard_range(B * world, world, rank)
    x0 = torch.as_tensor(pb["x0"][lo:hi], device=device)
    U = torch.as_tensor(pb["U"][lo:hi], device=device)
    cost = g.CostParams(cfg["gamma"], pb["Q"], pb["R"])
    want_grad = not args.forward_only

    fullcov = args.config == "C5"          # config 5: full covariance propagation
    if fullcov:
        pack.enable_fullcov()

    def step():
        if fullcov:
            r = g.rollout_fullcov(pack, x0, U, cost, want_grad=want_grad)
        else:
            r = g.rollout(pack, x0, U, cost, want_grad=want_grad, want_traj=False, graph=args.graph)
        if world > 1:
            return gather_results(r["cost"], r.get("grad"), dist)
        return r["cost"], r.get("grad")

    for _ in range(args.warmup):
        c, gr = step()
    torch.cuda.synchronize()
    if not torch.isfinite(c).all() or (gr is not None and not torch.isfinite(gr).all()):
        raise SystemExit("non-finite rollout outputs")

    lib().gpmpc_timing_enable(0 if args.graph else 1)     # per-kernel events and graph replay exclude each other
    import ctypes
    ms, nl = ctypes.c_double(), ctypes.c_longlong()
    lib().gpmpc_pair_kernel_time(ctypes.byref(ms), ctypes.byref(nl), 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    lib().gpmpc_pair_kernel_time(ctypes.byref(ms), ctypes.byref(nl), 1)
    lib().gpmpc_timing_enable(0)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        rollouts = B * world * args.steps
        value = rollouts / elapsed
        pairs_per_launch = B * ds * N * (N + 1) / 2                 # one horizon step, all trajectories and GPs
        if fullcov:                                                 # + ds(ds-1)/2 cross units over all N^2 ordered pairs
            pairs_per_launch += B * (ds * (ds - 1) / 2) * N * N
        fl, slots = pair_flops(D)
        if not want_grad:
            fl, slots = 2 * D + 37, D + 22
        if fullcov and want_grad:          # full second moments: D(D+1)/2 instead of D accumulations per pair
            fl, slots = 4 * D + 37 + D * (D - 1), 2 * D + 23 + D * (D - 1) // 2
        if want_grad and not fullcov:
            # horizon step 1 has constant state inputs: its ds state-dimension P*V accumulations are not part of the
            # algorithm (the kernel skips them); average the per-launch count over the H launches of a rollout
            fl, slots = fl - 2.0 * ds / H, slots - 1.0 * ds / H
        launch_s = (ms.value / max(nl.value, 1)) * 1e-3 if nl.value else float('nan')
        achieved = pairs_per_launch * fl / launch_s / 1e12
        out = {
            "metric": "GP-MPC rollouts/sec (N train pts x H horizon x d dims)",
            "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: N={N}, d(state_dim)={ds}, action_dim={da}, H={H}, "
                                   f"B={B} trajectories per GPU, gamma={cfg['gamma']}, "
                                   + ("objective+gradient" if want_grad else "objective only"),
                       "N": N, "state_dim": ds, "action_dim": da, "H": H, "batch_per_gpu": B,
                       "parallelism": f"trajectory-sharded x{world}" if world > 1 else "single GPU"},
            "roofline": {
                "kernel": "gpmpc_pair_kernel", "bound": "mfma",
                "bound_note": "fp64 VALU issue (software exp dominates); priced against the fp64 peak, which is the "
                              "same 78.6 TFLOP/s for vector and MFMA on MI355X. HBM is not binding at B>=4.",
                "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                # Fabric-side bytes per launch from the PMC passes of the same command (FETCH_SIZE x 2 + WRITE_SIZE, the
                # gfx950 correction for wide reads; an upper bound here, the scalar-cache line fills are uncalibrated, and
                # Infinity-Cache hits are included): profiles/r01/pmc_c3_final.txt.  Only measured for the default C3
                # workload.  Algorithmic: 67 MB of M + 168 MB of column rows; the dispatch order lets 4 row tiles share
                # each fetch of a trajectory's rows and re-reads each M tile 4x (pair_kernel_sb.h) -- 0.5 TB/s, far
                # from binding.
                "traffic": 1.14e9 if (args.config == "C3" and B == 256 and want_grad and not fullcov) else None,
                "traffic_source": "profiles/r01/pmc_c3_final.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, x2 read correction)",
                "algorithmic_flops_per_pair": fl, "algorithmic_slots_per_pair": slots,
                "valu_slot_frac": pairs_per_launch * slots / launch_s / 39.3e12,
                "pairs_per_launch": pairs_per_launch, "avg_launch_ms": launch_s * 1e3, "launches": nl.value,
                "hbm_algorithmic_GBs": (8 * (ds * N * (N + 1) / 2 + (ds * (ds - 1) / 2 * N * N if fullcov else 0))) / launch_s / 1e9,
                "hbm_frac": (8 * (ds * N * (N + 1) / 2 + (ds * (ds - 1) / 2 * N * N if fullcov else 0))) / launch_s / 1e9 / HBM_PEAK_GBS,
            },
            "pack_build_ms": pack_ms,
        }
        if world == 1 and want_grad and not fullcov:
            # objective-only rate beside the headline, outside the timed region
            for _ in range(2):
                g.rollout(pack, x0, U, cost, want_grad=False, want_traj=False)
            torch.cuda.synchronize()
            tf = time.perf_counter()
            for _ in range(3):
                g.rollout(pack, x0, U, cost, want_grad=False, want_traj=False)
            torch.cuda.synchronize()
            out["forward_only_rollouts_per_s"] = 3 * B / (time.perf_counter() - tf)
        if not args.no_cpu_baseline and world == 1:
            H_s = 2 if N >= 1024 else H
            # the GPU box exposes every host core but grants a 16-core share per GPU: more threads than
            # that only oversubscribe (measured: 256 threads are >100x slower than 16)
            torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
            res, _ = cpu_baseline(pb, H_s, reps=1, fullcov=fullcov)
            out["cpu_baseline"] = {
                "value": res["faithful"], "unit": "rollouts/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"1 trajectory, {H_s} of {H} horizon steps, objective+gradient, faithful-op restatement "
                          f"(N^3 trace GEMM + autograd) scaled x{H / H_s:g}; best of 1 after 1 warm-up",
                "o2_value": res["o2"],
                "o2_note": "same oracle with the trace evaluated as an O(N^2) elementwise sum (algorithmic baseline)",
                "c_port_value": res.get("cport"),
                "c_port_value_1_thread": res.get("cport_1thread"),
                "c_port_note": "plain-C / OpenMP port of the O(N^2) algorithm with the analytic adjoint (oracle/cport), "
                               "2 trajectories over the whole horizon, same thread count; includes its own pack build",
                "gpu_over_cpu": value / res["faithful"], "gpu_over_cpu_o2": value / res["o2"],
                "gpu_over_c_port": (value / res["cport"]) if res.get("cport") else None,
            }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
