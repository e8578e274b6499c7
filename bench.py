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
    ap.add_argument("--n-train", type=int, default=0,
                    help="training-set size instead of the config's N (side measurements, e.g. the mid-size case N = 1024, B = 16; "
                         "the line then names the changed N and carries no PMC sidecar)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=5, help="timed repetitions of the CPU baseline legs (min is reported)")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay each rollout as one hipGraph (small batches)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the secondary workloads the default run times after the headline (C4, C5, C3 with one lambda, C3 at B = 1 "
                         "as one graph, the closed loop): each is a short bench.py child process, reported under `extras`")
    ap.add_argument("--full-json", action="store_true",
                    help="print the FULL record on stdout (every note and definition: ~4 KB per line, 14 KB with the legs) instead of "
                         "the compact one; the default run prints the compact record and writes the full one to stderr")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the PCIe-inclusive and objective-only side measurements (profiling passes: only the timed workload runs)")
    ap.add_argument("--kinv-cache", default="",
                    help="file to load the inverse kernel matrices from / save them to (profiling passes: rocprofv3 --pmc crashes "
                         "inside rocSOLVER's 4096^2 LU, so an unprofiled run writes the file and the profiled runs read it)")
    ap.add_argument("--shared-lambda", action="store_true",
                    help="every GP gets the same length-scales (the regime of the reference's experiments): exponent and exp "
                         "are evaluated once per pair for all GPs (pair_kernel_sbs.h)")
    ap.add_argument("--closed-loop", action="store_true",
                    help="time the CALLERS of the path instead (SURVEY.md 8 f1/f2): Simulator.run on the pendulum plant with the "
                         "training set growing by one observation per step; prints ONE JSON line with ms per environment step "
                         "split into solve / pack build / inverse update")
    ap.add_argument("--cl-pretrain", type=int, default=200)
    ap.add_argument("--cl-steps", type=int, default=200)
    ap.add_argument("--cl-horizon", type=int, default=10)
    ap.add_argument("--cl-starts", type=int, default=1,
                    help="closed loop: lock-step multi-start solve with this many starts (RiskSensitiveMPC.n_starts; every solver "
                         "iteration is one batched rollout of that many candidate plans)")
    ap.add_argument("--cl-distinct", action="store_true", help="closed loop: a different lambda per GP (no shared inverse)")
    ap.add_argument("--cl-async", action="store_true", help="closed loop: the every-64 full rebuild on a side stream (catch-up + swap)")
    ap.add_argument("--cl-newton", action="store_true",
                    help="closed loop: every 64 appends a Newton-Schulz polish of the incrementally updated inverse instead of the full rebuild")
    ap.add_argument("--cl-rebuild", action="store_true", help="closed loop: the reference's O(N^3) rebuild on every step")
    ap.add_argument("--dist-timeout", type=float, default=120.0, help="process-group timeout in seconds (--gpus > 1)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal only: allow more ranks than visible GPUs (ranks share cards; use --backend gloo)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, created before this process initialises any GPU state
# ----------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv=None, poll_s=0.2):
    """Start the N rank processes and relay rank 0's JSON line.  ALL children are polled: a rank that dies -- any rank, not
    only rank 0 -- ends the job within a second with a non-zero status (the survivors would otherwise sit in a collective until
    the process-group timeout).  `argv`: the per-rank command (tests substitute a stand-in for bench.py)."""
    import selectors
    import torch                                   # device_count() does not initialise the GPU on this image
    ndev = torch.cuda.device_count()
    if argv is None and ndev < args.gpus and not args.oversubscribe:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible (pass --oversubscribe --backend gloo for a "
              f"rehearsal with several ranks per card)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = argv if argv is not None else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stdout is filtered down to the JSON line (gloo prints connection banners on stdout)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    rc = 0
    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    out_open = True
    try:
        while True:
            if out_open:
                for _key, _ev in sel.select(timeout=poll_s):
                    line = procs[0].stdout.readline()
                    if not line:
                        sel.unregister(procs[0].stdout)
                        out_open = False
                    elif line.lstrip().startswith("{"):
                        sys.stdout.write(line)
                        sys.stdout.flush()
            else:
                time.sleep(poll_s)
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:                                # one rank died: the others would wait in a collective
                rc = bad[0][1]
                print(f"bench.py: rank {bad[0][0]} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                break
            if all(c == 0 for c in codes):
                if out_open:                       # drain what rank 0 wrote before it exited
                    for line in procs[0].stdout:
                        if line.lstrip().startswith("{"):
                            sys.stdout.write(line)
                            sys.stdout.flush()
                break
    finally:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        deadline = time.time() + 5.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
    return rc if rc else 0


# ----------------------------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------------------------
def pair_work(D, ds, want_grad, fullcov, shared=False):
    """Algorithmic work per pair-evaluation of the FULL pair kernel: forward+gradient
    2D+23 fp64 issue slots = 4D+37 flops (FMA = 2); objective only D+22 / 2D+37; the full-covariance
    rollout accumulates D(D+1)/2 second moments instead of D.  With one lambda for all GPs the exponent and the exp of a
    pair are common to its ds GPs: of the 4D+37 flops, 2D+3 (weight x exp, row sum, P.V) are per GP and 2D+34 per pair, so
    the algorithmic count per pair-and-GP is (2D+34)/ds + 2D+3 (slots: (D+21)/ds + D+2)."""
    if shared and want_grad and not fullcov:
        return (2 * D + 34) / ds + 2 * D + 3, (D + 21) / ds + D + 2
    if shared and not want_grad and not fullcov:
        return (2 * D + 35) / ds + 2, (D + 21) / ds + 1
    if not want_grad:
        return 2 * D + 37, D + 22
    if fullcov:
        return 4 * D + 37 + D * (D - 1), 2 * D + 23 + D * (D - 1) // 2
    return 4 * D + 37, 2 * D + 23


def pair_instr(D, ds, want_grad, fullcov, shared_ng=0):
    """STATIC instruction mix of the pair kernel's column loop per pair-evaluation (one (i, j) term of one GP), read off
    the ISA of the instances bench.py times: (fp64-rate VALU instructions, integer VALU instructions,
    executed flops with FMA = 2 and add / mul = 1; cvt / fract / ldexp are fp64-rate issue slots but not flops).
    pair_kernel_sb.h:  exponent 1 add + D fma | exp cvt, fract, 2 fma, mul, fma, ldexp | P = M e mul | r add | v D fma | w ds fma.
    pair_kernel_sbs.h (shared lambda, groups of shared_ng GPs): exponent and exp once per pair and group.
    pair_kernel_sbf.h (full S): v D fma, W ds(ds+1)/2 fma."""
    exp_f64, exp_int = 7, 3                      # table exp: 7 fp64-rate (3 of them fma, 1 mul) + 3 integer
    head_f64 = 1 + D + exp_f64                   # per pair, shared by the GPs of a group in the shared-lambda kernel
    head_fl = 1 + 2 * D + (2 * 3 + 1)
    if not want_grad:
        per_f64, per_fl = 2, 2                   # P = M e, r += P
    elif fullcov:
        nw = ds * (ds + 1) // 2
        per_f64, per_fl = 2 + D + nw, 2 + 2 * (D + nw)
    else:
        per_f64, per_fl = 2 + D + ds, 2 + 2 * (D + ds)
    g = float(shared_ng) if shared_ng else 1.0
    return head_f64 / g + per_f64, exp_int / g, head_fl / g + per_fl


def build_kinv(pb, device):
    """Ky_inv of every GP on the device: Kf/Ky by the HIP kernel (src/gpr.py:163-170), inverse by
    torch.linalg.inv as the reference (src/gpr.py:171)."""
    import numpy as np
    import torch
    from gaussian_process_mpc_amd._lib import lib, ptr, stream_ptr, host_doubles, check
    X = torch.as_tensor(pb["X"], device=device)
    N, D = X.shape
    Ky = torch.empty((pb["ds"], N, N), dtype=torch.float64, device=device)
    for a in range(pb["ds"]):
        _, lp = host_doubles(pb["lambdas"][a])
        noise = float(np.float32(pb["sigma_n"][a] ** 2))          # src/gpr.py:170 adds a float32 diagonal
        check(lib().gpmpc_build_ky(N, D, ptr(X), lp, float(pb["sigma_f"][a]), noise, None, ptr(Ky[a]), stream_ptr()),
              "gpmpc_build_ky")
    return torch.linalg.inv(Ky)                                   # ONE batched LU over the (ds, N, N) stack


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


def reference_time_ratio(config):
    """oracle-faithful time / reference time on identical inputs, measured in the build container by
    tools/check_cpu_baseline_vs_reference.py (the reference cannot travel to the GPU box)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03", "cpu_baseline_vs_reference.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    row = d.get("configs", {}).get(config) or d.get("configs", {}).get("C3")
    if not row:
        return None
    return {"ratio": row["oracle_over_reference_time"], "measured_on": f"{config if config in d['configs'] else 'C3'} sizes, H = {row['H_timed']}, "
            f"{d.get('threads')} threads of {d.get('cpu')} (build container); reference {row['reference_s']:.3f} s, oracle "
            f"{row['oracle_faithful_s']:.3f} s, costs agree to {row['cost_rel_diff']:.1e}"}


def measured_traffic(config, B, want_grad, kernel_hint):
    """HBM-side bytes per launch of the dominant kernel from the tracked PMC summary of THIS workload
    (profiles/r02/pmc_<config>.json, written by tools/prof_pmc.sh: separate --pmc passes of `bench.py --config <config>`;
    FETCH_SIZE x 2 for the gfx950 wide-read correction + WRITE_SIZE, both in KiB).  None when no summary matches."""
    d, rnd, name = None, None, None
    sh = "_shared" if kernel_hint == "sbs" else ""
    for rnd in ("r05", "r04", "r03", "r02"):
        for name in (f"pmc_{config}_B{B}{sh}.json", f"pmc_{config}{sh}.json"):
            try:
                with open(os.path.join(ROOT, "profiles", rnd, name)) as f:
                    d = json.load(f)
            except (OSError, ValueError):
                d = None
                continue
            if d.get("batch_per_gpu") == B:
                break
            d = None
        if d is not None:
            break
    if d is None:
        return None, None
    if d.get("batch_per_gpu") != B or bool(d.get("want_grad", True)) != bool(want_grad) or \
            bool(d.get("shared_lambda", False)) != bool(kernel_hint == "sbs"):
        return None, None
    k = d.get("dominant_kernel", {})
    if "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        return None, None
    traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    src = (f"profiles/{rnd}/{name} (HEAD {d.get('head', '?')}; `{d.get('command', '?')}`; kernel {k.get('name', '?')}; "
           f"2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; Infinity-Cache hits are included in FETCH_SIZE)")
    return traffic, src


# ----------------------------------------------------------------------------------------------------------------------
# secondary workloads of the default run
# ----------------------------------------------------------------------------------------------------------------------
LEGS = [
    # name, bench.py arguments (each leg: its own process, its own barrier + synchronize bracket, its own roofline block)
    ("C4", ["--config", "C4", "--steps", "3", "--warmup", "1"]),
    ("C5", ["--config", "C5", "--steps", "3", "--warmup", "1"]),
    ("C3_shared_lambda", ["--config", "C3", "--shared-lambda", "--steps", "5", "--warmup", "2"]),
    ("C5_shared_lambda", ["--config", "C5", "--shared-lambda", "--steps", "3", "--warmup", "1"]),
    ("C3_B1_graph", ["--config", "C3", "--batch", "1", "--graph", "--steps", "100", "--warmup", "20"]),
    ("C4_B1_graph", ["--config", "C4", "--batch", "1", "--graph", "--steps", "20", "--warmup", "5"]),
    ("C3_B8_graph", ["--config", "C3", "--batch", "8", "--graph", "--steps", "50", "--warmup", "10"]),
    ("C5_B1", ["--config", "C5", "--batch", "1", "--steps", "20", "--warmup", "5"]),
    ("N300_B256_shared_lambda", ["--config", "C3", "--n-train", "300", "--batch", "256", "--shared-lambda", "--steps", "20", "--warmup", "5"]),
    ("N300_B256", ["--config", "C3", "--n-train", "300", "--batch", "256", "--steps", "20", "--warmup", "5"]),
    ("closed_loop_newton", ["--closed-loop", "--cl-newton", "--cl-steps", "50"]),
    ("closed_loop_newton_16_starts", ["--closed-loop", "--cl-newton", "--cl-steps", "50", "--cl-starts", "16"]),
]


def short_kernel(name):
    """Kernel instance without the explanatory parenthetical the full record appends."""
    return (name or "").split(" (")[0]


def run_legs(timeout_s=240):
    """The other BASELINE configs and regimes, each as a short `bench.py` child process AFTER the headline's timed region (the
    headline line stays what it was; a child starts with a fresh GPU context and frees everything on exit).  Returns
    ({name: COMPACT record}, {name: full record}): the driver keeps 8 KB of stdout, so the stdout line carries at most ~250 bytes
    per leg -- value, ms_per_step, kernel, bound, frac, avg_launch_ms -- and the full records (commands, workload strings, every
    roofline key) go to stderr and gpurun_out/.  A leg that fails or times out is recorded as such, never dropped silently."""
    out, full = {}, {}
    for name, argv in LEGS:
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--no-cpu-baseline", "--no-extras", "--no-legs", "--full-json"]
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            lines = [ln for ln in p.stdout.splitlines() if ln.lstrip().startswith("{")]
            if p.returncode != 0 or not lines:
                out[name] = {"error": f"exit status {p.returncode}"}
                full[name] = {"error": f"exit status {p.returncode}", "stderr_tail": p.stderr[-400:], "command": " ".join(argv)}
                continue
            d = json.loads(lines[-1])
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"timed out after {timeout_s} s"}
            full[name] = dict(out[name], command=" ".join(argv))
            continue
        d["command"] = "bench.py " + " ".join(argv)
        d["leg_wall_s"] = time.perf_counter() - t0
        full[name] = d
        rec = {"value": _sig(d.get("value")), "unit": d.get("unit"), "ms_per_step": _sig(d.get("ms_per_step"))}
        r = d.get("roofline")
        if r:
            rec.update({"kernel": short_kernel(r.get("kernel")), "bound": r.get("bound"), "frac": _sig(r.get("frac"), 3),
                        "avg_launch_ms": _sig(r.get("avg_launch_ms"))})
        if "step_ms" in d:                                 # closed loop: ms per environment step
            rec.update({"p95": _sig(d["step_ms"].get("p95")), "max": _sig(d["step_ms"].get("max")),
                        "solve": _sig((d.get("split_ms_mean") or {}).get("solve")),
                        "inverse_update": _sig((d.get("inverse_update_ms") or {}).get("append_mean")),
                        "starts": d.get("starts")})
            rec.pop("ms_per_step", None)
        out[name] = {k: v for k, v in rec.items() if v is not None}
    return out, full


def _sig(v, n=4):
    """Round to n significant digits (compact JSON); passes None / non-floats through."""
    if not isinstance(v, float) or v != v or v in (float("inf"), float("-inf")) or v == 0.0:
        return v
    from math import floor, log10
    return round(v, n - 1 - int(floor(log10(abs(v)))))


# keys of the full roofline / cpu_baseline blocks that stay on the stdout line (numbers and short identifiers; the explanatory
# strings -- bound_note, definition, *_note, traffic_source, sample details -- go to stderr and DESIGN.md section 5)
ROOF_KEEP = ("kernel", "bound", "served_from", "achieved", "peak", "unit", "frac", "traffic", "traffic_measured_in_run", "traffic_sidecar",
             "algorithmic_flops_per_pair", "pairs_per_launch", "avg_launch_ms", "launches", "sub_batches_per_call", "issue_util",
             "executed_flops_frac", "hbm_frac", "valu_frac_algorithmic", "first_step_variant", "kernel_timed_in", "plan")
CPU_KEEP = ("value", "unit", "cores", "kind", "sample", "cpu_model", "faithful_1_thread", "o2_value", "c_port_value",
            "c_port_value_1_thread", "oracle_over_reference_time", "gpu_over_cpu", "gpu_over_c_port")


def compact_record(full):
    """The stdout line: every top-level key of the contract, `roofline` and `cpu_baseline` with their numbers, `extras` compact."""
    out = {}
    for k, v in full.items():
        if k == "roofline":
            r = {kk: v[kk] for kk in ROOF_KEEP if kk in v}
            r["kernel"] = short_kernel(r.get("kernel"))
            if isinstance(r.get("plan"), dict):
                r["plan"] = {kk: r["plan"][kk] for kk in ("form", "tiling", "workgroups", "launches_per_step", "split") if kk in r["plan"]}
            r["kernel_timed_in"] = "timed region" if r.get("kernel_timed_in") == "the timed region" else "separate uncaptured pass"
            r["notes"] = "definitions: DESIGN.md section 5; full record on stderr (BENCH_FULL) and in gpurun_out/bench_full.json"
            out[k] = {kk: (_sig(vv, 5) if isinstance(vv, float) else vv) for kk, vv in r.items()}
        elif k == "cpu_baseline":
            out[k] = {kk: (_sig(v[kk], 4) if isinstance(v[kk], float) else v[kk]) for kk in CPU_KEEP if kk in v}
        elif k in ("extras_full", "multi_gpu_notes"):
            continue
        elif k == "sustained":
            out[k] = {kk: _sig(vv) if isinstance(vv, float) else vv for kk, vv in v.items()}
        else:
            out[k] = v
    return out


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
        from datetime import timedelta
        # a rank that never arrives must fail the job well inside the driver's 600 s, not after the 10-minute default
        tmo = timedelta(seconds=args.dist_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)
        assert dist.get_world_size() == args.gpus

    import gaussian_process_mpc_amd as g
    from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
    from gaussian_process_mpc_amd._lib import lib
    from gaussian_process_mpc_amd.parallel import shard_range, gather_results

    cfg = dict(CONFIGS[args.config])
    if args.n_train:
        cfg["N"] = args.n_train
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
    bcast_ms = None
    if world > 1:
        torch.cuda.synchronize(); dist.barrier()
        tb0 = time.perf_counter()
        dist.broadcast(kinv, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - tb0) * 1e3       # Ky_inv broadcast (once per data update; not in the rate)
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
    if (not torch.isfinite(c).all() or (gr is not None and not torch.isfinite(gr).all())) and \
            not os.environ.get("GPMPC_BENCH_TIMING_EXPERIMENT"):      # (timing-only experimental kernel builds compute wrong numbers)
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
    for cls in (0, 1, 2):                                  # 0: the full pair kernel, 1: its horizon-step-1 variant, 2: fused step kernel
        L.gpmpc_pair_kernel_time_class(cls, ctypes.byref(ms), ctypes.byref(nl))
        tcls.append((ms.value, nl.value))
    L.gpmpc_timing_enable(0)
    multi = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # instrumentation of the first real multi-GPU run (outside the timed region): were it to scale below expectation, these
        # say whether the ranks differ (per-rank elapsed of the timed region, rollout-only rate without the collective) or the
        # collective costs (latency of the fused all_gather of [cost | grad] alone, 100 repetitions)
        el_all = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(el_all, torch.tensor([elapsed], dtype=torch.float64, device=device))
        torch.cuda.synchronize()
        tq = time.perf_counter()
        for _ in range(max(2, min(args.steps, 5))):
            if fullcov:
                g.rollout_fullcov(pack, x0, U, cost, want_grad=want_grad)
            else:
                g.rollout(pack, x0, U, cost, want_grad=want_grad, want_traj=False, graph=args.graph)
        torch.cuda.synchronize()
        local_rate = max(2, min(args.steps, 5)) * (hi - lo) / (time.perf_counter() - tq)
        lr_all = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(lr_all, torch.tensor([local_rate], dtype=torch.float64, device=device))
        c0 = torch.zeros(hi - lo, dtype=torch.float64, device=device)
        g0 = torch.zeros((hi - lo, H, da), dtype=torch.float64, device=device) if want_grad else None
        for _ in range(5):
            gather_results(c0, g0, dist)
        torch.cuda.synchronize(); dist.barrier()
        tg = time.perf_counter()
        for _ in range(100):
            gather_results(c0, g0, dist)
        torch.cuda.synchronize()
        ag_us = (time.perf_counter() - tg) / 100 * 1e6
        multi = {"per_rank_elapsed_s": [float(t.item()) for t in el_all],
                 "per_rank_rollouts_per_s_without_collective": [float(t.item()) for t in lr_all],
                 "rollouts_per_s_min_max_over_ranks": [min(float(t.item()) for t in lr_all), max(float(t.item()) for t in lr_all)],
                 "all_gather_cost_grad_latency_us": ag_us, "all_gather_bytes_per_rank": 8 * (hi - lo) * (1 + (H * da if want_grad else 0)),
                 "kinv_broadcast_ms": bcast_ms, "kinv_bytes": 8 * ds * N * N}
        elapsed = float(tmax.item())

    if rank == 0:
        rollouts = B * world * args.steps
        value = rollouts / elapsed
        pairs_per_launch = B * ds * N * (N + 1) / 2                 # one horizon step, all trajectories and GPs
        # (one lambda, 2 <= ds <= 4: the cross units run in their own kernel, pair_kernel_sbfx.h -- beside the timed launch at small batches --;
        # the timed launch then covers the variance units only, and so do its pair count and weight bytes)
        fc_plan = pack.plan_fullcov(B, H, want_grad=want_grad) if fullcov else {}
        cross_elsewhere = bool(fc_plan.get("shared_cross_units"))
        if fullcov and not cross_elsewhere:                         # + ds(ds-1)/2 cross units over all N^2 ordered pairs
            pairs_per_launch += B * (ds * (ds - 1) / 2) * N * N
        shared = bool(pack.shared_lambda) and not fullcov and os.environ.get("GPMPC_SHARED", "") != "0"
        fl, slots = pair_work(D, ds, want_grad, fullcov, shared)
        full_ms, full_n = tcls[0]
        fused_path = full_n == 0 and tcls[2][1] > 0            # small batches: one fused launch per horizon step
        if fused_path:
            full_ms, full_n = tcls[2]
        if not full_n and args.graph and world == 1:
            # graph replay carries no per-kernel events: time the dominant kernel in a short UNCAPTURED pass outside the timed region
            # (same launches, same inputs; HIP events on the launch stream)
            L.gpmpc_timing_enable(1)
            L.gpmpc_pair_kernel_time(ctypes.byref(ms), ctypes.byref(nl), 1)
            for _ in range(max(3, min(args.steps, 20))):
                g.rollout(pack, x0, U, cost, want_grad=want_grad, want_traj=False, graph=False)
            torch.cuda.synchronize()
            tc2 = []
            for cls in (0, 1, 2):
                L.gpmpc_pair_kernel_time_class(cls, ctypes.byref(ms), ctypes.byref(nl))
                tc2.append((ms.value, nl.value))
            L.gpmpc_timing_enable(0)
            tcls = tc2
            full_ms, full_n = tcls[0]
            fused_path = full_n == 0 and tcls[2][1] > 0
            if fused_path:
                full_ms, full_n = tcls[2]
            kernel_timed_in = "a separate uncaptured pass after the timed region (graph replay has no per-kernel events)"
        else:
            kernel_timed_in = "the timed region"
        launch_s = (full_ms / full_n) * 1e-3 if full_n else float("nan")
        # mid-size batches run as concurrent sub-batches (step.hip split_count): one timed launch then covers B / S trajectories
        plan = pack.plan(B, H, want_grad=want_grad, graph=args.graph) if not fullcov else pack.plan_fullcov(B, H, want_grad=want_grad)
        persist = plan.get("form") == "persist"                # whole-horizon kernel: ONE launch per rollout call
        per_rollout = 1 if persist else (H if fused_path else max(H - 1, 1))
        n_timed_calls = args.steps if kernel_timed_in == "the timed region" else max(3, min(args.steps, 20))
        n_sub = max(1, int(round(full_n / float(n_timed_calls * per_rollout)))) if full_n else 1
        if persist:
            pairs_per_launch *= H
        pairs_per_launch /= n_sub
        achieved = pairs_per_launch * fl / launch_s / 1e12
        m_bytes = 8 * (ds * N * (N + 1) / 2 + (ds * (ds - 1) / 2 * N * N if (fullcov and not cross_elsewhere) else 0))     # M read once per launch
        sm = "sbf" if fullcov else ("sbs" if shared else "sb")
        # (no sidecar for the launch that covers the variance units only: the recorded counters belong to the per-unit launch)
        traffic, traffic_src = (None, None) if (args.n_train or cross_elsewhere) else measured_traffic(args.config, B, want_grad, sm)
        ng = 0
        if shared:                                             # GPs per workgroup: gpmpc_sbs_group (gpmpc_internal.h)
            cap = max(2, min(4, 48 // (1 + D + ds)))
            groups = (ds + cap - 1) // cap
            ng = (ds + groups - 1) // groups
        i_f64, i_int, i_fl = pair_instr(D, ds, want_grad, fullcov, ng)
        # every VALU instruction (fp64-rate or integer) occupies its SIMD for 4 cycles per wave64 at the spec clock
        issue_util = (i_f64 + i_int) * 4.0 * pairs_per_launch / 64.0 / (1024 * launch_s * 2.4e9)
        executed_frac = pairs_per_launch * i_fl / launch_s / 1e12 / FP64_PEAK_TFLOPS
        kname = plan.get("kernel") or f"gpmpc_pair_kernel_{sm}"
        if fullcov and plan.get("form") == "four_launch":
            kname = f"gpmpc_pair_kernel_{sm}"
        kname += (" (whole horizon of one trajectory per workgroup: every step's mean sums, pair sums and finish work)" if persist else
                  " (one launch per horizon step: mean sums + finish work + pair tiles; NOT a pair-only time)" if fused_path
                  else " (full variant; the cheaper horizon-step-1 variant is reported under first_step_variant)")
        # Which roofline binds the dominant kernel: per launch it must move the upper triangles of the weight
        # matrices once (8 bytes per pair when ONE trajectory streams them, 8 / B with B trajectories sharing a launch) and issue the
        # column loop's VALU instructions; whichever takes longer at the machine's peaks is the bound.
        m_launch = m_bytes                                      # every timed launch (a sub-batch's too) streams the matrices itself
        t_hbm = m_launch / (HBM_PEAK_GBS * 1e9)
        t_valu = (i_f64 + i_int) * 4.0 * pairs_per_launch / 64.0 / (1024 * 2.4e9)
        hbm_bound = t_hbm > t_valu
        # bytes of the weight stream that one step re-reads fit the 256 MiB Infinity Cache (guide: a table stays resident while
        # table + everything else touched between two uses fits): they are then served on-die, at the IC's ~8.6 TB/s, not by HBM
        ic_resident = m_bytes <= 200e6
        out = {
            "metric": "GP-MPC rollouts/sec (N train pts x H horizon x d dims)",
            "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}{' with N changed' if args.n_train else ''}: N={N}, d(state_dim)={ds}, action_dim={da}, H={H}, "
                                   f"B={B} trajectories per GPU, gamma={cfg['gamma']}, "
                                   + ("full covariance, " if fullcov else "")
                                   + ("one lambda for all GPs (shared-lambda kernel), " if args.shared_lambda else "")
                                   + ("objective+gradient" if want_grad else "objective only"),
                       "N": N, "state_dim": ds, "action_dim": da, "H": H, "batch_per_gpu": B,
                       "shared_lambda": bool(args.shared_lambda),
                       "parallelism": f"trajectory-sharded x{world}" if world > 1 else "single GPU"},
            "dist": {"world_size": dist.get_world_size() if world > 1 else 1,
                     "backend": dist.get_backend() if world > 1 else None,
                     "visible_devices": ndev, "launcher": "torch.distributed.run / external" if "TORCHELASTIC_RUN_ID" in os.environ
                     else ("bench.py spawn" if world > 1 else "single process")},
            "roofline": ({
                "kernel": kname,
                "bound": "hbm", "served_from": "infinity_cache" if ic_resident else "hbm",
                "bound_note": f"weight stream: one launch must read the upper triangles of the {ds} weight matrices ({m_launch / 1e6:.1f} MB "
                              f"algorithmic) and that takes longer at 8 TB/s ({t_hbm * 1e6:.1f} us) than the column loop's VALU "
                              f"instructions at the spec clock ({t_valu * 1e6:.1f} us); "
                              + ("the matrices fit the 256 MiB Infinity Cache and are re-read every horizon step, so they are served "
                                 "on-die (guide: ~8.6 TB/s for IC reads) -- still priced against the 8 TB/s HBM figure" if ic_resident
                                 else "the matrices exceed the Infinity Cache: HBM3E, 8 TB/s spec (6.3 TB/s achievable per the guide)"),
                "achieved": m_launch / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": m_launch / launch_s / 1e9 / HBM_PEAK_GBS,
                "valu_frac_algorithmic": achieved / FP64_PEAK_TFLOPS,
            } if hbm_bound else {
                "kernel": kname,
                "bound": "valu_fp64",
                "bound_note": "fp64 VALU issue: no MFMA instruction is executed (fp64 MFMA shares the fp64 VALU's issue capacity on "
                              "MI355X, profiles/r01/ubench_mfma_f64_overlap.txt) and HBM is not binding at B >= 4; priced against the "
                              "fp64 vector peak of 78.6 TFLOP/s",
                "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
            }) | {
                "kernel_timed_in": kernel_timed_in, "plan": plan,
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_sidecar": traffic_src.split(" ")[0] if traffic_src else None,
                "traffic_measured_in_run": False,
                "traffic_note": "PMC counters need separate rocprofv3 --pmc passes: `traffic` is the per-launch figure of the tracked "
                                "sidecar named in traffic_source (its HEAD is stated there), null when none matches this workload",
                "definition": "achieved = pairs_per_launch x algorithmic_flops_per_pair / avg_launch_ms of the dominant kernel "
                              "(HIP events on its launch stream over the timed region); algorithmic flops per pair = 4D+37 "
                              "(objective+gradient, SURVEY.md 8d; FMA = 2; prices a 19-slot libm exp where the kernel executes a "
                              "7-slot table exp, so `frac` is an ALGORITHMIC rate, not a utilisation -- the two bounded figures are "
                              "issue_util and executed_flops_frac)",
                "algorithmic_flops_per_pair": fl, "algorithmic_slots_per_pair_survey_8d": slots, "pairs_per_launch": pairs_per_launch,
                "avg_launch_ms": launch_s * 1e3, "launches": full_n, "sub_batches_per_call": n_sub,
                "first_step_variant": {"avg_launch_ms": (tcls[1][0] / tcls[1][1]) if tcls[1][1] else None, "launches": tcls[1][1]},
                "issue_util": None if fused_path else issue_util,
                "issue_util_note": f"static VALU instructions of the column loop per pair ({i_f64:g} fp64-rate + {i_int:g} integer) x 4 "
                                   "cycles x pairs / 64 lanes over 1024 SIMDs x launch time x 2.4 GHz (spec clock; the chip holds "
                                   "~2.25 GHz under this kernel): <= 1 by construction",
                "executed_flops_frac": None if fused_path else executed_frac,
                "executed_flops_note": f"flops the kernel EXECUTES per pair ({i_fl:g}: FMA = 2, add / mul = 1; conversions, fract and "
                                       "ldexp are issue slots but not flops) against the 78.6 TFLOP/s peak",
                "hbm_algorithmic_GBs": m_launch / launch_s / 1e9, "hbm_frac": m_launch / launch_s / 1e9 / HBM_PEAK_GBS,
                "hbm_note": "weight-matrix bytes one launch must read at least once (upper triangles, 8 bytes per pair; sub-batches of a "
                            "split call each read them) / avg_launch_ms / 8 TB/s",
            },
            "pack_build_ms": pack_ms,
        }
        if multi is not None:
            out["multi_gpu"] = multi
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
        if world == 1 and not args.graph and not args.no_extras:
            # sustained rate: the same step repeated for ~5 s, outside the timed region (the timed K steps are a sub-second
            # burst; the kernel runs the chip on its power cap, so this is the number a long job sees)
            torch.cuda.synchronize()
            ts, ns = time.perf_counter(), 0
            while time.perf_counter() - ts < 5.0:
                for _ in range(4):
                    step()
                ns += 4
                torch.cuda.synchronize()
            out["sustained"] = {"seconds": time.perf_counter() - ts, "steps": ns, "rollouts_per_s": ns * B / (time.perf_counter() - ts)}
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
        default_run = (args.config == "C3" and not args.batch and not args.n_train and not args.shared_lambda and not args.graph
                       and not args.forward_only)
        if world == 1 and default_run and not args.no_legs and not args.no_extras:
            out["extras"], out["extras_full"] = run_legs()
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
                "oracle_over_reference_time": (reference_time_ratio(args.config) or {}).get("ratio"),
                "oracle_over_reference_note": (reference_time_ratio(args.config) or {}).get("measured_on"),
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
        full = _clean(out)
        if args.full_json:
            full.pop("extras_full", None)
            print(json.dumps(full, allow_nan=False), flush=True)
        else:
            # stdout: the compact record (< 6 KB: the driver keeps an 8 KB tail); stderr + gpurun_out/: everything
            print("BENCH_FULL " + json.dumps(full, allow_nan=False), file=sys.stderr, flush=True)
            try:
                if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                    with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
                        json.dump(full, f)
            except OSError:
                pass
            line = json.dumps(compact_record(full), allow_nan=False, separators=(",", ":"))
            print(line, flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def run_closed_loop(args):
    """Simulator.run (src/simulator.py:37-60) on PendulumPlant, unrolled here so that every phase of an environment step
    can be timed: solve (get_optimal_trajectory: ~50-300 objective+gradient callbacks of the B = 1 rollout), pack build
    (O(N^2): beta, folded weights), and the update of Ky_inv for the new observation (src/simulator.py:55 ->
    src/gpr.py:171: O(N^3) rebuild per GP in the reference; here one O(N^2) Schur append shared by the GPs with identical
    hyper-parameters, and a full rebuild every 64 appends)."""
    import numpy as np
    import torch
    import gaussian_process_mpc_amd as g
    dev = g.require_gpu()
    rng = np.random.default_rng(0)
    plant = g.PendulumPlant()
    H = args.cl_horizon
    mpc = g.RiskSensitiveMPC(1e-5, H, 2, 1, Q=2 * np.eye(2), R=0.001 * np.eye(1))
    mpc.n_starts = max(1, args.cl_starts)
    for k, gp in enumerate(mpc.dynamics.gpr_err):         # hypers before data, as in pretrain_uncertainty.py:100-105
        gp.set_lambdas(np.array([0.5, 0.5, 0.5]) * (1.0 + (0.1 * k if args.cl_distinct else 0.0)))
        gp.set_sigma_n(1e-3)
    n0 = args.cl_pretrain
    S = np.column_stack((rng.uniform(-np.pi, np.pi, n0), rng.uniform(-8, 8, n0)))
    A = rng.uniform(-2, 2, (n0, 1))
    NS = np.empty_like(S)
    for i in range(n0):
        plant.state = S[i].copy()
        NS[i] = plant.step(A[i])[0]
    # phase boundaries: the whole device, or -- with the side-stream rebuild -- the stream the loop itself works on
    sync = (lambda: torch.cuda.current_stream(dev.index).synchronize()) if args.cl_async else (lambda: torch.cuda.synchronize(dev))  # noqa: E731
    t0 = time.perf_counter(); mpc.dynamics.append_train_data(S, A, NS); sync()
    t_first_build = (time.perf_counter() - t0) * 1e3
    mpc.set_lb([-2.0]); mpc.set_ub([2.0]); mpc.set_xref(np.zeros(2))
    obs, _ = plant.reset()
    # warm-up, untimed: library load, graph capture, and ONE whole environment step (the first incremental append pays the
    # one-time load of its kernels: 15-60 ms)
    mpc.dynamics.pack(); a0 = mpc.get_optimal_trajectory(obs)[0, :]
    nxt0 = plant.step(a0)[0]
    if args.cl_newton:
        mpc.dynamics.gpr_err[0]._newton_refresh(); sync()      # untimed: the GEMM library loads its fp64 kernels on first use (~12 ms once)
    mpc.dynamics.append_train_data(obs, a0, nxt0, incremental=not args.cl_rebuild, async_rebuild=args.cl_async,
                                   refresh="newton" if args.cl_newton else None); sync()
    obs = nxt0
    n0 += 1
    rows, ticks, plan_cost = [], [], []
    for it in range(args.cl_steps):
        sync(); ta = time.perf_counter()
        mpc.dynamics.pack(); sync(); tb = time.perf_counter()
        action = mpc.get_optimal_trajectory(obs)[0, :]; tc = time.perf_counter()
        if mpc.last_solve_info is not None:                # (untimed bookkeeping: the cost of the plan that was returned)
            ticks.append(mpc.last_solve_info["evaluations"]); plan_cost.append(float(np.min(mpc.last_solve_info["f"])))
        else:
            plan_cost.append(float(mpc.objective(np.asarray(mpc.last_traj))))
        tc2 = time.perf_counter()
        nxt, _, _, _, _ = plant.step(action); td = time.perf_counter()
        before = mpc.dynamics.gpr_err[0]._appends_since_rebuild
        mpc.dynamics.append_train_data(obs, action, nxt, incremental=not args.cl_rebuild); sync(); te = time.perf_counter()
        rows.append((tb - ta, tc - tb, td - tc2, te - td, mpc.dynamics.gpr_err[0]._appends_since_rebuild <= before))
        obs = nxt
    r = np.array([[a, b, c, d] for a, b, c, d, _ in rows]) * 1e3
    full = np.array([x[4] for x in rows])
    total = r.sum(axis=1)
    out = {
        "metric": "closed-loop GP-MPC, ms per environment step (Simulator.run on the pendulum plant)", "unit": "ms",
        "value": float(np.median(total)), "higher_is_better": False, "n_gpus": 1, "data": "synthetic", "dtype": "f64",
        "config": {"workload": f"PendulumPlant, ds=2, da=1, H={H}, training set {n0} -> {n0 + args.cl_steps}, lambda = 0.5 "
                               + ("x (1, 1.1) per GP (distinct)" if args.cl_distinct else "for every GP (the reference's regime)")
                               + ", sigma_n = 1e-3, solver " + str(mpc.solver_used)
                               + (", full O(N^3) rebuild per step (the reference's update)" if args.cl_rebuild else
                                  ", O(N^2) Schur append per step, full rebuild every 64"
                                  + (" on a side stream (catch-up + swap)" if args.cl_async else "")
                                  + (" -- replaced by a Newton-Schulz polish of the updated inverse" if args.cl_newton else ""))},
        "steps": args.cl_steps,
        "step_ms": {"median": float(np.median(total)), "mean": float(total.mean()), "max": float(total.max()),
                    "p95": float(np.percentile(total, 95)), "max_over_median": float(total.max() / np.median(total))},
        # the solve's share varies with the stand-in optimiser's iteration count; what the data path adds per step is this:
        "step_ms_excluding_solve": {"median": float(np.median(total - r[:, 1])), "max": float((total - r[:, 1]).max()),
                                    "max_on_full_rebuild_steps": float((total - r[:, 1])[full].max()) if full.any() else None},
        "worst_non_solve_step": (lambda k: {"index": int(k), "n_train": int(n0 + k), "pack_build": float(r[k, 0]), "plant": float(r[k, 2]),
                                             "inverse_update": float(r[k, 3]), "full_rebuild": bool(full[k])})(int(np.argmax(total - r[:, 1]))),
        "worst_step_with_median_solve_over_median_step": float((np.median(r[:, 1]) + (total - r[:, 1]).max()) / np.median(total)),
        "split_ms_mean": {"pack_build": float(r[:, 0].mean()), "solve": float(r[:, 1].mean()), "plant": float(r[:, 2].mean()),
                          "inverse_update": float(r[:, 3].mean())},
        "inverse_update_ms": {"append_mean": float(r[~full, 3].mean()) if (~full).any() else None,
                              "full_rebuild_mean": float(r[full, 3].mean()) if full.any() else None,
                              "full_rebuilds": int(full.sum())},
        "first_build_ms": t_first_build,
        "starts": mpc.n_starts,
        "batched_evaluations_per_solve_mean": float(np.mean(ticks)) if ticks else None,
        "planned_cost_mean": float(np.mean(plan_cost)) if plan_cost else None,
        "solve_callbacks_note": "solve = scipy L-BFGS-B stand-in on the objective / gradient callbacks (cyipopt absent): optimiser "
                                "results are unpinned, the timing split is what is reported",
    }
    print(json.dumps(out), flush=True)
    return 0


def main():
    args = parse_args()
    if args.closed_loop:
        return run_closed_loop(args)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
