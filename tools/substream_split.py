#!/usr/bin/env python3
"""Mid-size batches: ONE gpmpc_rollout call against the same batch split into S sub-batches on S streams (the head
kernel / launch ramp of one sub-batch overlaps the pair kernel of another).  Objective + gradient, eager launches, inputs
resident.  Run on the GPU box:  python tools/substream_split.py [N:ds:da:H:B ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.rollout import CostParams, GPPack, rollout
from gaussian_process_mpc_amd.synth import synth_problem

shapes = sys.argv[1:] or ["1024:4:1:20:8", "1024:4:1:20:16", "1024:4:1:20:32", "1024:4:1:20:64", "2048:4:1:20:8",
                          "2048:4:1:20:16", "2048:4:1:20:32", "300:2:1:10:64", "512:3:1:20:32"]
dev = g.require_gpu()
packs = {}
for shape in shapes:
    N, ds, da, H, B = (int(v) for v in shape.split(":"))
    key = (N, ds, da)
    if key not in packs:
        pb = synth_problem(3, N, ds, da, H, 256)
        kinv = []
        for a in range(ds):
            gp = g.GaussianProcessRegression(ds + da)
            gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(1.0)); gp.set_sigma_n(np.array(pb["sigma_n"][a]))
            gp.append_train_data(pb["X"], pb["Y"][:, a]); kinv.append(gp.Ky_inv)
        packs[key] = (pb, GPPack(pb["X"], pb["Y"], torch.stack(kinv), pb["lambdas"], pb["sigma_f"]))
    pb, pack = packs[key]
    cost = CostParams(-1.0, pb["Q"], pb["R"])
    x0, U = torch.as_tensor(pb["x0"][:B], device=dev), torch.as_tensor(pb["U"][:B, :H], device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]

    def run(S):
        if S == 1:
            return [rollout(pack, x0, U, cost, want_traj=False)]
        out, h = [], (B + S - 1) // S
        ev = torch.cuda.Event(); ev.record()
        for k in range(S):
            if k * h >= B:
                break
            with torch.cuda.stream(streams[k]):
                streams[k].wait_event(ev)
                out.append(rollout(pack, x0[k * h:(k + 1) * h], U[k * h:(k + 1) * h], cost, want_traj=False))
        for k in range(min(S, len(out))):
            torch.cuda.current_stream().wait_stream(streams[k])
        return out

    ref = run(1)[0]["grad"]
    line = f"N={N} ds={ds} H={H} B={B:3d}:"
    for S in (1, 2, 4, 8):
        if S > B:
            continue
        got = torch.cat([r["grad"] for r in run(S)])
        torch.cuda.synchronize()
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-9)
        for _ in range(3): run(S)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for _ in range(reps): run(S)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        line += f"  S={S}: {dt * 1e3:7.3f} ms ({B / dt:8.0f}/s)"
    print(line, flush=True)
