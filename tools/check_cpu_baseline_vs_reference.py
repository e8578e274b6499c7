#!/usr/bin/env python3
"""Is bench.py's `cpu_baseline` (the oracle's faithful-op mode, oracle/gpmpc_oracle.py) the reference's CPU path within
noise?  BUILD CONTAINER ONLY: imports the reference from /root/reference (only shim: an empty `cyipopt` module), times
`RiskSensitiveMPC.objective` + `gradient` of the reference against `objective_and_gradient(mode="faithful")` of the oracle
on identical inputs -- BASELINE config C2 (whole horizon) and C3 sizes at H = 2 (what bench.py samples) -- checks that the
values agree, and writes the time ratios to profiles/r03/cpu_baseline_vs_reference.json, which bench.py quotes as
`cpu_baseline.oracle_over_reference_time` (ratio < 1: the oracle is FASTER than the reference, i.e. the quoted GPU / CPU
factor is conservative).

    PYTHONDONTWRITEBYTECODE=1 python tools/check_cpu_baseline_vs_reference.py [--threads 8] [--reps 3]
"""
import argparse, json, os, sys, time, types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GPMPC_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.modules.setdefault("cyipopt", types.ModuleType("cyipopt"))
from src.mpc import RiskSensitiveMPC                                    # noqa: E402  (the reference)
from oracle import gpmpc_oracle as O                                    # noqa: E402
from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
torch.set_num_threads(args.threads)


def best(fn, reps):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); t.append(time.perf_counter() - t0)
    return min(t), out


rows = {}
for name, hs in (("C2", None), ("C3", 2)):
    cfg = CONFIGS[name]
    N, ds, da, H, gamma = cfg["N"], cfg["ds"], cfg["da"], hs or cfg["H"], cfg["gamma"]
    pb = synth_problem(int(name[1]), N, ds, da, cfg["H"], 1)
    mpc = RiskSensitiveMPC(gamma, H, ds, da, pb["Q"], pb["R"], None)
    for a in range(ds):
        g = mpc.dynamics.gpr_err[a]
        g.set_lambdas(pb["lambdas"][a]); g.set_sigma_n(float(pb["sigma_n"][a])); g.set_sigma_f(1.0)
    mpc.dynamics.append_train_data(pb["X"][:, :ds], pb["X"][:, ds:], pb["Y"])
    mpc.curr_state = torch.tensor(pb["x0"][0]).type(torch.float64)
    x = pb["U"][0, :H].reshape(-1).copy()

    def run_ref():
        mpc.curr_cost = None
        c = mpc.objective(x.copy())
        return c, np.asarray(mpc.gradient(x.copy())).reshape(H, da)

    kinv = np.stack([g.Ky_inv.detach().numpy() for g in mpc.dynamics.gpr_err])
    lam = np.stack([torch.exp(g.log_lambdas).detach().numpy() for g in mpc.dynamics.gpr_err])
    gp = O.GPBundle(pb["X"], pb["Y"], lam, pb["sigma_f"], pb["sigma_n"], Ky_inv=kinv)

    def run_orc():
        r = O.objective_and_gradient(gp, H, pb["x0"][0], pb["U"][0, :H], pb["x_ref"], pb["u_ref"], pb["Q"], pb["R"], gamma,
                                     mode="faithful")
        return r["cost"], r["grad"]

    t_ref, (c_ref, g_ref) = best(run_ref, args.reps)
    t_orc, (c_orc, g_orc) = best(run_orc, args.reps)
    rows[name] = {"N": N, "state_dim": ds, "H_timed": H, "reference_s": t_ref, "oracle_faithful_s": t_orc,
                  "oracle_over_reference_time": t_orc / t_ref, "cost_rel_diff": abs(c_orc / c_ref - 1.0),
                  "grad_max_rel_diff": float(np.max(np.abs(g_orc - g_ref) / np.maximum(np.abs(g_ref), 1e-12)))}
    print(name, rows[name], flush=True)

out = {"threads": args.threads, "reps": args.reps, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
       "what": "min wall-clock of objective+gradient, reference (imported from /root/reference) vs oracle faithful mode, same inputs",
       "configs": rows}
os.makedirs(os.path.join(ROOT, "profiles", "r03"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "r03", "cpu_baseline_vs_reference.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote profiles/r03/cpu_baseline_vs_reference.json")
