#!/usr/bin/env python3
"""GP-MPC rollout benchmark (BASELINE.json metric: rollouts/sec).

One *rollout* = for one action sequence U (H x da): H-step moment-matching propagation of all ds
GPs (means + variances) + risk-sensitive cost + gradient w.r.t. U, i.e. one objective+gradient
callback pair of the reference (src/mpc.py:202-255).  One bench *step* = one batched call of the hot
path over B trajectories per GPU, inputs already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W] [--config C3]

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment spawns N rank processes itself (one per
GPU, RCCL over xGMI) BEFORE anything touches a GPU in this process, relays rank 0's JSON line and
exits with the worst child status; under ``python -m torch.distributed.run --nproc-per-node N`` the
ranks already exist and WORLD_SIZE must equal N.  Either way the line's ``n_gpus`` is the number of
ranks that really ran (``dist.world_size``), never the flag.

Prints ONE JSON line on rank 0.  Multi-GPU = weak scaling: every rank owns B trajectories of a
global batch of N*B (independent candidates; the GP pack is replicated, broadcast from rank 0, and
no collective sits on the data path besides the fused all_gather of [cost | grad]).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector peak (= fp64 MFMA peak), 2.4 GHz x 256 CUs x 4 SIMDs x 32 flop/clk
FP64_PEAK_TSLOTS = 39.3      # the same in issue slots (one fp64 lane-instruction = one slot; an FMA is 2 flops)
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (default: the config's B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=5, help="timed repetitions of the CPU baseline legs (min is reported)")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay each rollout as one hipGraph (small batches)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the PCIe-inclusive and objective-only side measurements (profiling passes: only the timed workload runs)")
    ap.add_argument("--kinv-cache", default="",
                    help="file to load the inverse kernel matrices from / save them to (profiling passes: rocprofv3 --pmc crashes "
                         "inside rocSOLVER's 4096^2 LU, so an unprofiled run writes the file and the profiled runs read it)")
    ap.add_argument("--shared-lambda", action="store_true",
                    help="every GP gets the same length-scales (the regime of the reference's experiments): exponent and exp "
                         "are evaluated once per pair for all GPs (pair_kernel_sbs.h)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal only: allow more ranks than visible GPUs (ranks share cards; use --backend gloo)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, created before this process initialises any GPU state
# ----------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    import torch                                   # device_count() does not initialise the GPU on this image
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not args.oversubscribe:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible (pass --oversubscribe --backend gloo for a "
              f"rehearsal with several ranks per card)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stdout is filtered down to the JSON line (gloo prints connection banners on stdout)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    rc = 0
    try:
        for line in procs[0].stdout:
            if line.lstrip().startswith("{"):
                sys.stdout.write(line)
                sys.stdout.flush()
        for p in procs:
            code = p.wait()
            rc = rc or code
            if code != 0:                          # one rank died: the others would wait in a collective for ever
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------------------------
def pair_work(D, ds, want_grad, fullcov):
    """Algorithmic work per pair-evaluation of the FULL pair kernel: forward+gradient
    2D+23 fp64 issue slots = 4D+37 flops (FMA = 2); objective only D+22 / 2D+37; the full-covariance
    rollout accumulates D(D+1)/2 second moments instead of D."""
    if not want_grad:
        return 2 * D + 37, D + 22
    if fullcov:
        return 4 * D + 37 + D * (D - 1), 2 * D + 23 + D * (D - 1) // 2
    return 4 * D + 37, 2 * D + 23


def build_kinv(pb, device):
    """Ky_inv of every GP on the device: Kf/Ky by the HIP kernel (src/gpr.py:163-170), inverse by
    torch.linalg.inv as the reference (src/gpr.py:171)."""
    import numpy as np
    import torch
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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pb, cfg, fullcov, reps):
    """The reference CPU path timed on this box's host cores on a BOUNDED sample of the same workload: ONE trajectory,
    H_s of the H horizon steps, objective + gradient, scaled linearly to H (the per-step cost is constant).

    * faithful  - oracle/gpmpc_oracle.py in faithful-op mode (per-call Ky_inv @ y, N^3 trace GEMM, autograd): checked
                  against the imported reference by tests/test_oracle_golden.py; all cores and ONE thread;
    * o2        - the same with the trace as an O(N^2) elementwise sum (algorithmic baseline);
    * c_port    - plain-C / OpenMP port of the O(N^2) algorithm with the analytic adjoint (oracle/cport), whole horizon,
                  pack build excluded (difference of a 3- and a 1-trajectory run).
    min of ``reps`` after one warm-up, threads = the box's 16-core share per GPU."""
    import torch
    from oracle import gpmpc_oracle as O
    H, N = cfg["H"], cfg["N"]
    H_s = 2 if N >= 1024 else H
    ncores = min(16, len(os.sched_getaffinity(0)))        # more threads than the share only oversubscribe (256: >100x slower)
    gamma = cfg["gamma"]
    fn = O.objective_and_gradient_fullcov if fullcov else O.objective_and_gradient
    gp = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"])

    def timed(mode, hs, nrep):
        best = float("inf")
        for r in range(nrep + 1):
            t0 = time.perf_counter()
            fn(gp, hs, pb["x0"][0], pb["U"][0][:hs], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], gamma, mode=mode)
            dt = time.perf_counter() - t0
            if r > 0:
                best = min(best, dt)
        return 1.0 / (best * H / hs)

    res = {"cores": ncores, "cpu_model": cpu_model(), "H_sample": H_s}
    torch.set_num_threads(ncores)
    res["faithful"] = timed("faithful", H_s, reps)
    res["o2"] = timed("o2", H_s, max(1, min(reps, 3)))
    if not fullcov:
        torch.set_num_threads(1)
        hs1 = 1 if N >= 1024 else H_s
        res["faithful_1_thread"] = timed("faithful", hs1, max(1, min(reps, 2)))
        res["H_sample_1_thread"] = hs1
        torch.set_num_threads(ncores)
        # plain-C / OpenMP port.  It calls omp_set_num_threads, which also moves torch's thread count: the thread
        # count reported as `cores` was fixed above and is restored below.
        from oracle import cport
        kin = gp.Ky_inv.numpy()
        cport.rollout(pb, kin, gamma, x0=pb["x0"][:1], U=pb["U"][:1, :2], nthreads=ncores)       # build / warm-up
        best = float("inf")
        for _ in range(max(1, min(reps, 3))):
            t0 = time.perf_counter(); cport.rollout(pb, kin, gamma, x0=pb["x0"][:1], U=pb["U"][:1], nthreads=ncores); t1 = time.perf_counter() - t0
            t0 = time.perf_counter(); cport.rollout(pb, kin, gamma, x0=pb["x0"][:3], U=pb["U"][:3], nthreads=ncores); t3 = time.perf_counter() - t0
            best = min(best, (t3 - t1) / 2.0)
        res["cport"] = 1.0 / best
        t0 = time.perf_counter(); cport.rollout(pb, kin, gamma, x0=pb["x0"][:1], U=pb["U"][:1, :2], nthreads=1); ta = time.perf_counter() - t0
        t0 = time.perf_counter(); cport.rollout(pb, kin, gamma, x0=pb["x0"][:1], U=pb["U"][:1, :4], nthreads=1); tb = time.perf_counter() - t0
        res["cport_1_thread"] = 1.0 / ((tb - ta) / 2.0 * H)
        torch.set_num_threads(ncores)
    return res


def measured_traffic(config, B, want_grad, kernel_hint):
    """HBM-side bytes per launch of the dominant kernel from the tracked PMC summary of THIS workload
    (profiles/r02/pmc_<config>.json, written by tools/prof_pmc.sh: separate --pmc passes of `bench.py --config <config>`;
    FETCH_SIZE x 2 for the gfx950 wide-read correction + WRITE_SIZE, both in KiB).  None when no summary matches."""
    path = os.path.join(ROOT, "profiles", "r02", f"pmc_{config}.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, None
    if d.get("batch_per_gpu") != B or bool(d.get("want_grad", True)) != bool(want_grad):
        return None, None
    k = d.get("dominant_kernel", {})
    if "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        return None, None
    traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    src = (f"profiles/r02/pmc_{config}.json (HEAD {d.get('head', '?')}; `{d.get('command', '?')}`; kernel {k.get('name', '?')}; "
           f"2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; Infinity-Cache hits are included in FETCH_SIZE)")
    return traffic, src


# ----------------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    import ctypes
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) exist (WORLD_SIZE): refusing to report a "
                         f"{world}-rank number as {args.gpus} GPUs")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py: no HIP device visible (the rollout has no CPU fallback)")
    if world > ndev and not args.oversubscribe:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s) visible (rehearsals: --oversubscribe --backend gloo)")
    local = local % ndev                                   # rehearsals may put several ranks on one card
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus

    import gaussian_process_mpc_amd as g
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    from gaussian_process_mpc_amd._lib import lib
    from gaussian_process_mpc_amd.parallel import shard_range, gather_results

    cfg = dict(CONFIGS[args.config])
    B = args.batch or cfg["B"]
    if args.config == "C4" and not args.batch:
        B = cfg["B"] // 8                                  # 1024 trajectories over 8 GPUs
    cid = int(args.config[1])
    pb = synth_problem(cid, cfg["N"], cfg["ds"], cfg["da"], cfg["H"], B * world, shared_lambda=args.shared_lambda)
    N, ds, da, H, D = cfg["N"], cfg["ds"], cfg["da"], cfg["H"], cfg["ds"] + cfg["da"]

    # GP pack: rank 0 inverts, everyone receives the same bits
    if rank == 0:
        if args.kinv_cache and os.path.exists(args.kinv_cache):
            kinv = torch.load(args.kinv_cache, map_location=device)
        else:
            kinv = build_kinv(pb, device)
            if args.kinv_cache:
                torch.save(kinv.cpu(), args.kinv_cache)
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

    def step(U_dev=U):
        if fullcov:
            r = g.rollout_fullcov(pack, x0, U_dev, cost, want_grad=want_grad)
        else:
            r = g.rollout(pack, x0, U_dev, cost, want_grad=want_grad, want_traj=False, graph=args.graph)
        if world > 1:
            return gather_results(r["cost"], r.get("grad"), dist)
        return r["cost"], r.get("grad")

    for _ in range(args.warmup):
        c, gr = step()
    torch.cuda.synchronize()
    if not torch.isfinite(c).all() or (gr is not None and not torch.isfinite(gr).all()):
        raise SystemExit("non-finite rollout outputs")

    L = lib()
    L.gpmpc_timing_enable(0 if args.graph else 1)         # per-kernel events and graph replay exclude each other
    ms, nl = ctypes.c_double(), ctypes.c_longlong()
    L.gpmpc_pair_kernel_time(ctypes.byref(ms), ctypes.byref(nl), 1)
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
    tcls = []
    for cls in (0, 1):                                     # 0: the full pair kernel, 1: its horizon-step-1 variant
        L.gpmpc_pair_kernel_time_class(cls, ctypes.byref(ms), ctypes.byref(nl))
        tcls.append((ms.value, nl.value))
    L.gpmpc_timing_enable(0)
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
        fl, slots = pair_work(D, ds, want_grad, fullcov)
        full_ms, full_n = tcls[0]
        launch_s = (full_ms / full_n) * 1e-3 if full_n else float("nan")
        achieved = pairs_per_launch * fl / launch_s / 1e12
        m_bytes = 8 * (ds * N * (N + 1) / 2 + (ds * (ds - 1) / 2 * N * N if fullcov else 0))     # M read once per launch
        traffic, traffic_src = measured_traffic(args.config, B, want_grad, None)
        sm = "sbf" if fullcov else "sb"
        out = {
            "metric": "GP-MPC rollouts/sec (N train pts x H horizon x d dims)",
            "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: N={N}, d(state_dim)={ds}, action_dim={da}, H={H}, "
                                   f"B={B} trajectories per GPU, gamma={cfg['gamma']}, "
                                   + ("full covariance, " if fullcov else "")
                                   + ("objective+gradient" if want_grad else "objective only"),
                       "N": N, "state_dim": ds, "action_dim": da, "H": H, "batch_per_gpu": B,
                       "parallelism": f"trajectory-sharded x{world}" if world > 1 else "single GPU"},
            "dist": {"world_size": dist.get_world_size() if world > 1 else 1,
                     "backend": dist.get_backend() if world > 1 else None,
                     "visible_devices": ndev, "launcher": "torch.distributed.run / external" if "TORCHELASTIC_RUN_ID" in os.environ
                     else ("bench.py spawn" if world > 1 else "single process")},
            "roofline": {
                "kernel": f"gpmpc_pair_kernel_{sm} (full variant; the cheaper horizon-step-1 variant is reported under first_step_variant)",
                "bound": "valu_fp64",
                "bound_note": "fp64 VALU issue: no MFMA instruction is executed (fp64 MFMA shares the fp64 VALU's issue capacity on "
                              "MI355X, profiles/r01/ubench_mfma_f64_overlap.txt) and HBM is not binding at B >= 4; priced against the "
                              "fp64 vector peak of 78.6 TFLOP/s",
                "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src,
                "definition": "achieved = pairs_per_launch x algorithmic_flops_per_pair / avg_launch_ms of the full pair kernel "
                              "(HIP events on its launch stream over the timed region); algorithmic flops per pair = 4D+37 "
                              "(objective+gradient, SURVEY.md 8d; FMA = 2)",
                "algorithmic_flops_per_pair": fl, "pairs_per_launch": pairs_per_launch,
                "avg_launch_ms": launch_s * 1e3, "launches": full_n,
                "first_step_variant": {"avg_launch_ms": (tcls[1][0] / tcls[1][1]) if tcls[1][1] else None, "launches": tcls[1][1]},
                "frac_survey_8d_slots": pairs_per_launch * slots / launch_s / 1e12 / FP64_PEAK_TSLOTS,
                "frac_survey_8d_slots_note": f"SURVEY.md 8d's primary convention: {slots} algorithmic fp64 issue slots per pair "
                                             f"against {FP64_PEAK_TSLOTS}e12 slots/s (counts ocml's 19-slot exp; the kernel's table exp "
                                             f"issues 7)",
                "hbm_algorithmic_GBs": m_bytes / launch_s / 1e9, "hbm_frac": m_bytes / launch_s / 1e9 / HBM_PEAK_GBS,
            },
            "pack_build_ms": pack_ms,
        }
        if world == 1 and not args.graph and not args.no_extras:
            # PCIe-inclusive rate, outside the timed region: U from pinned host memory in, [cost | grad] back out, per step
            Uh = torch.as_tensor(pb["U"][lo:hi]).pin_memory()
            Ud = torch.empty_like(U)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(max(2, min(args.steps, 5))):
                Ud.copy_(Uh, non_blocking=True)
                c, gr = step(Ud)
                c.cpu()
                if gr is not None:
                    gr.cpu()
            torch.cuda.synchronize()
            out["pcie_inclusive_rollouts_per_s"] = max(2, min(args.steps, 5)) * B / (time.perf_counter() - tp)
        if world == 1 and want_grad and not fullcov and not args.no_extras:
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
            res = cpu_baseline(pb, cfg, fullcov, args.cpu_reps)
            H_s = res["H_sample"]
            out["cpu_baseline"] = {
                "value": res["faithful"], "unit": "rollouts/s", "cores": res["cores"], "kind": "port",
                "cpu_model": res["cpu_model"], "host_cpus_visible": len(os.sched_getaffinity(0)),
                "sample": f"1 trajectory, {H_s} of {H} horizon steps, objective+gradient, faithful-op restatement of the reference "
                          f"(N^3 trace GEMM + autograd; oracle/gpmpc_oracle.py) scaled x{H / H_s:g}; torch CPU fp64 on "
                          f"{res['cores']} threads; min of {args.cpu_reps} after 1 warm-up",
                "faithful_1_thread": res.get("faithful_1_thread"),
                "faithful_1_thread_sample": (f"{res['H_sample_1_thread']} of {H} steps scaled, 1 thread, min of "
                                             f"{max(1, min(args.cpu_reps, 2))}") if "faithful_1_thread" in res else None,
                "o2_value": res["o2"],
                "o2_note": "same oracle with the trace evaluated as an O(N^2) elementwise sum (algorithmic baseline)",
                "c_port_value": res.get("cport"), "c_port_value_1_thread": res.get("cport_1_thread"),
                "c_port_note": "plain-C / OpenMP port of the O(N^2) algorithm with the analytic adjoint (oracle/cport), whole "
                               "horizon, its own pack build excluded (3-trajectory minus 1-trajectory run), same thread count",
                "gpu_over_cpu": value / res["faithful"], "gpu_over_cpu_o2": value / res["o2"],
                "gpu_over_c_port": (value / res["cport"]) if res.get("cport") else None,
            }
        assert out["n_gpus"] == args.gpus

        def _clean(v):                       # strict JSON: no NaN / Infinity (graph replay has no per-kernel timing)
            if isinstance(v, dict):
                return {k: _clean(x) for k, x in v.items()}
            if isinstance(v, float) and (v != v or v in (float("inf"), float("-inf"))):
                return None
            return v
        print(json.dumps(_clean(out), allow_nan=False), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
