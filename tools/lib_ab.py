#!/usr/bin/env python3
"""A/B of BUILDS of the library (and / or GPMPC_* settings) over problem shapes on ONE device: every variant runs in its own process
(GPMPC_LIB_PATH selects the .so), all variants of a round back to back, two rounds, best time per variant.  Objective + gradient,
each rollout replayed as one hipGraph and synchronised (the latency a solver callback sees), inputs resident.  The gradients of every
variant are compared with those of the first (bitwise with --bitwise, else rtol 1e-6).

    python tools/lib_ab.py --variants base=gaussian_process_mpc_amd/csrc/libgpmpc_hip_r03.so new=gaussian_process_mpc_amd/csrc/libgpmpc_hip.so \
        [--variants 'x=lib.so,GPMPC_TILING=5'] [--eager] [--queued] [--shared-lambda] --shapes N:ds:da:H:B,...
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import numpy as np, torch
    import gaussian_process_mpc_amd as g
    from gaussian_process_mpc_amd.rollout import CostParams, GPPack, rollout
    from gaussian_process_mpc_amd.synth import synth_problem
    dev = g.require_gpu()
    out, packs = {}, {}
    for shape in a.shapes:
        N, ds, da, H, B = (int(v) for v in shape.split(":"))
        key = (N, ds, da)
        if key not in packs:
            packs.clear(); torch.cuda.empty_cache()
            pb = synth_problem(3, N, ds, da, H, 512, shared_lambda=a.shared_lambda)
            kinv = []
            for k in range(ds):
                gp = g.GaussianProcessRegression(ds + da)
                gp.set_lambdas(pb["lambdas"][k]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][k]))
                gp.append_train_data(pb["X"], pb["Y"][:, k]); kinv.append(gp.Ky_inv)
            packs[key] = (pb, GPPack(pb["X"], pb["Y"], torch.stack(kinv), pb["lambdas"], pb["sigma_f"]))
            del kinv
        pb, pack = packs[key]
        cost = CostParams(-1.0, pb["Q"], pb["R"])
        x0, U = torch.as_tensor(pb["x0"][:B], device=dev), torch.as_tensor(pb["U"][:B, :H], device=dev)
        run = lambda: rollout(pack, x0, U, cost, want_traj=False, graph=not a.eager)     # noqa: E731
        r = run(); torch.cuda.synchronize()
        grad = r["grad"].cpu().numpy().copy(); cst = r["cost"].cpu().numpy().copy()
        for _ in range(5): run()
        torch.cuda.synchronize()
        reps = max(10, min(200, int(2e9 / (B * H * ds * N * N / 2 * 30))))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
                if not a.queued:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps)
        out[shape] = {"ms": best * 1e3, "grad": grad.tolist(), "cost": cst.tolist()}
    json.dump(out, open(a.child_out, "w"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="+", action="append", default=[])
    ap.add_argument("--eager", action="store_true"); ap.add_argument("--queued", action="store_true")
    ap.add_argument("--shared-lambda", action="store_true"); ap.add_argument("--bitwise", action="store_true")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--child-out", default="")
    ap.add_argument("--shapes", default="", help="comma separated N:ds:da:H:B")
    a = ap.parse_args()
    a.shapes = [x for x in a.shapes.split(",") if x]
    if a.child_out:
        return child(a)
    import numpy as np
    variants = [v for grp in a.variants for v in grp]
    res = {}
    for rnd in range(a.rounds):
        for v in variants:
            name, spec = v.split("=", 1)
            parts = spec.split(",")
            env = dict(os.environ)
            if parts[0]:
                env["GPMPC_LIB_PATH"] = os.path.abspath(parts[0])
            for kv in parts[1:]:
                env[kv.split("=")[0]] = kv.split("=")[1]
            tmp = f"/tmp/lib_ab_{os.getpid()}_{name}.json"
            cmd = [sys.executable, os.path.abspath(__file__), "--child-out", tmp] + (["--eager"] if a.eager else []) + \
                  (["--queued"] if a.queued else []) + (["--shared-lambda"] if a.shared_lambda else []) + ["--shapes", ",".join(a.shapes)]
            p = subprocess.run(cmd, env=env, capture_output=True, text=True)
            if p.returncode != 0:
                print(f"variant {name} failed:\n{p.stderr[-2000:]}", flush=True)
                continue
            d = json.load(open(tmp)); os.remove(tmp)
            for s, r in d.items():
                cur = res.setdefault(s, {}).setdefault(name, r)
                cur["ms"] = min(cur["ms"], r["ms"])
    names = [v.split("=", 1)[0] for v in variants]
    for s in a.shapes:
        if s not in res or names[0] not in res[s]:
            continue
        ref = res[s][names[0]]
        line = f"{s:>18s}"
        for n in names:
            if n not in res[s]:
                line += f"   {n}: failed"; continue
            r = res[s][n]
            gr, g0 = np.array(r["grad"]), np.array(ref["grad"])
            same = np.array_equal(gr, g0) and np.array_equal(np.array(r["cost"]), np.array(ref["cost"]))
            if a.bitwise:
                assert same, (s, n)
            # different kernel forms sum the cancelling N^2 terms in different orders: gradients agree to ~1e-4 of their scale
            dev = float(np.max(np.abs(gr - g0)) / (np.max(np.abs(g0)) + 1e-300))
            assert dev < 1e-4, (s, n, dev)
            line += f"   {n}: {r['ms']:8.4f} ms x{ref['ms'] / r['ms']:.3f}{' =' if same else f' ~{dev:.0e}'}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
