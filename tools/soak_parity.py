#!/usr/bin/env python3
"""Long randomised parity soak of the diagonal rollout against the plain-C checker (oracle/cport): shapes, ragged N, batch
sizes on both sides of every kernel-selection threshold, forced kernel shapes (tuning overrides + reload_tuning), eager /
graph / solver-callback entries, objective-only calls.  Not part of the test-suite; run on the GPU box:
    python tools/soak_parity.py [n_cases] [seed]
Prints one line per case and a summary; exits non-zero if a case misses the north-star tolerances."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import synth_problem
from oracle import cport, gpmpc_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
KNOBS = ("GPMPC_PERSIST", "GPMPC_FUSED_SB", "GPMPC_PAIR_SB", "GPMPC_PAIR_TB", "GPMPC_TILING", "GPMPC_FUSED", "GPMPC_NO_FIRST", "GPMPC_HEAD_CHUNKS", "GPMPC_SHARED",
         "GPMPC_SB_UNROLL", "GPMPC_SPLIT", "GPMPC_SHARED_NG", "GPMPC_XCDMAP")
SHAPES = {"auto": {}, "sb256": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "0"}, "sb64": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2"},
          "sb_tb1": {"GPMPC_PAIR_SB": "1", "GPMPC_PAIR_TB": "1"}, "staged": {"GPMPC_PAIR_SB": "0", "GPMPC_FUSED": "0"},
          "staged_tb4": {"GPMPC_PAIR_SB": "0", "GPMPC_FUSED": "0", "GPMPC_PAIR_TB": "4"}, "nofirst": {"GPMPC_NO_FIRST": "1"},
          "chunks3": {"GPMPC_FUSED": "0", "GPMPC_HEAD_CHUNKS": "3"}, "sb64_chunks8": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2", "GPMPC_HEAD_CHUNKS": "8"},
          # round 3: four columns in flight / one, concurrent sub-batches forced on and off, shared-lambda group sizes
          "sb64_u4": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2", "GPMPC_SB_UNROLL": "4"}, "sb64_u1_split4": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2", "GPMPC_SB_UNROLL": "1", "GPMPC_SPLIT": "4"},
          "sb256_split2": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "0", "GPMPC_SPLIT": "2"}, "nosplit": {"GPMPC_SPLIT": "1"},
          "sh_ng2": {"GPMPC_PAIR_SB": "1", "GPMPC_SHARED_NG": "2"}, "sh_off": {"GPMPC_SHARED": "0"}, "sh_sb64": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2", "GPMPC_SPLIT": "3"},
          # 256x128 tiles, two trajectories per wave (work list 4), alone and as two concurrent sub-batches
          # one launch per step on 256x64 tiles (step_fused.h, Q = 0) forced on wherever the scalar-broadcast path can run, and off
          "fsb": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2", "GPMPC_FUSED_SB": "1"}, "fsb_split2": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2", "GPMPC_FUSED_SB": "1", "GPMPC_SPLIT": "2"},
          "fsb_off": {"GPMPC_FUSED_SB": "0"}, "fsb_ng2": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2", "GPMPC_FUSED_SB": "1", "GPMPC_SHARED_NG": "2"},
          "fsb32": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "5", "GPMPC_FUSED_SB": "1"}, "fsb16": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "6", "GPMPC_FUSED_SB": "1"},
          "persist16": {"GPMPC_PERSIST": "16"}, "persist8": {"GPMPC_PERSIST": "8"}, "persist_off": {"GPMPC_PERSIST": "0"},      # round 4: whole-horizon kernel
          # round 5: the one-launch forms in the XCD-aware dispatch order forced on (any B > 1) and off
          "fsb_xcd1": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "2", "GPMPC_FUSED_SB": "1", "GPMPC_XCDMAP": "1"}, "fsb32_xcd1": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "5", "GPMPC_FUSED_SB": "1", "GPMPC_XCDMAP": "1"},
          "xcd0": {"GPMPC_XCDMAP": "0"}, "xcd1_nopersist": {"GPMPC_XCDMAP": "1", "GPMPC_PERSIST": "0"},
          "sb128": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "4"}, "sb128_split2": {"GPMPC_PAIR_SB": "1", "GPMPC_TILING": "4", "GPMPC_SPLIT": "2"}}
worst = {"means": 0.0, "vars": 0.0, "cost": 0.0, "grad": 0.0}
bad = 0
t_start = time.time()
for case in range(n_cases):
    N = int(rng.choice([1, 3, 31, 64, 65, 127, 128, 200, 257, 300, 449, 512, 700, 1025, 1500, 2049]))
    ds = int(rng.integers(1, 7)); da = int(rng.integers(1, 3))
    if ds + da > 8:
        ds = 8 - da
    H = int(rng.integers(1, 10))
    B = int(rng.choice([1, 1, 2, 3, 4, 7, 16, 40, 130, 600, 1500]))
    if N >= 700:
        B = min(B, 130)
    gamma = float(rng.choice([-1.0, 1e-5, 0.0, 0.5]))
    shape = str(rng.choice(list(SHAPES)))
    entry = str(rng.choice(["eager", "graph", "callback", "autograd"] if B == 1 else ["eager", "eager", "graph", "autograd"]))
    grad = bool(rng.random() < 0.8) or entry in ("callback", "autograd")
    shared = bool(rng.random() < 0.45)            # one lambda for every GP: the shared-lambda kernel wherever the sb path runs
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(SHAPES[shape])
    pb = synth_problem(5000 + case, N, ds, da, H, B, shared_lambda=shared)
    pb["Q"] = pb["Q"] + 0.01 * (np.ones((ds, ds)) - np.eye(ds))
    kinv = O.GPBundle(pb["X"], pb["Y"], pb["lambdas"], pb["sigma_f"], pb["sigma_n"]).Ky_inv.numpy()
    pack = g.GPPack(pb["X"], pb["Y"], kinv, pb["lambdas"], pb["sigma_f"])          # reads the overrides
    cp = g.CostParams(gamma, pb["Q"], pb["R"])
    if entry == "callback":
        cg = pack.objective_gradient(pb["x0"][0], pb["U"][0], cp)
        r = {"cost": torch.tensor(cg[:1]), "grad": torch.tensor(cg[1:]).reshape(1, H, da)}
    elif entry == "autograd":                      # the differentiable boundary: RolloutFunction -> CostFunction -> backward
        from gaussian_process_mpc_amd.autograd import CostFunction, RolloutFunction
        Ut = torch.tensor(pb["U"], device=pack.device, requires_grad=True)
        m, v = RolloutFunction.apply(torch.tensor(pb["x0"], device=pack.device), Ut, pack)
        cst = CostFunction.apply(m, torch.diag_embed(v), Ut, cp)
        cst.sum().backward()
        r = {"means": m.detach(), "vars": v.detach(), "cost": cst.detach(), "grad": Ut.grad}
    else:
        r = g.rollout(pack, pb["x0"], pb["U"], cp, want_grad=grad, graph=(entry == "graph"))
        if entry == "graph":                       # replay once more: the captured graph, not the capture run
            r = g.rollout(pack, pb["x0"], pb["U"], cp, want_grad=grad, graph=True)
    pick = sorted({0, B // 2, B - 1})
    c = cport.rollout(pb, kinv, gamma, x0=pb["x0"][pick], U=pb["U"][pick], nthreads=16)
    rel = lambda a, ref, fl: float(np.max(np.abs(a - ref) / np.maximum(np.abs(ref), fl))) if a.size else 0.0   # noqa: E731
    err = {k: 0.0 for k in worst}
    if "means" in r:
        err["means"] = rel(r["means"][pick].cpu().numpy(), c["means"], 1e-3)      # means crossing zero: 1e-8 absolute
        err["vars"] = rel(r["vars"][pick].cpu().numpy(), c["vars"], 1e-8)
    err["cost"] = rel(r["cost"][pick].cpu().numpy(), c["cost"], 1e-6)
    if grad:
        err["grad"] = rel(r["grad"][pick].cpu().numpy(), c["grad"], 1e-3)
    finite = all(bool(torch.isfinite(v).all()) for v in r.values())
    ok = finite and err["means"] < 1e-5 and err["vars"] < 1e-4 and err["cost"] < 1e-6 and err["grad"] < 1e-4
    bad += 0 if ok else 1
    print(f"case {case:3d}: N={N:4d} ds={ds} da={da} H={H} B={B:4d} gamma={gamma:g} {shape:14s} {entry:8s} shared={int(shared)} grad={int(grad)} "
          + " ".join(f"{k} {v:.1e}" for k, v in err.items()) + ("" if ok else "   <-- FAIL"), flush=True)
    for k in worst:
        worst[k] = max(worst[k], err[k])
    del pack
print(f"{n_cases} cases in {time.time() - t_start:.0f} s, {bad} outside the tolerances; worst:", {k: f"{v:.1e}" for k, v in worst.items()})
sys.exit(1 if bad else 0)
