import sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, ".")
import gaussian_process_mpc_amd as g
from gaussian_process_mpc_amd.synth import CONFIGS, synth_problem
for cid in ("C1", "C2"):
    cfg = CONFIGS[cid]
    pb = synth_problem(int(cid[1]), cfg["N"], cfg["ds"], cfg["da"], cfg["H"], 1)
    mpc = g.RiskSensitiveMPC(cfg["gamma"], cfg["H"], cfg["ds"], cfg["da"], pb["Q"], pb["R"])
    for a, gp in enumerate(mpc.dynamics.gpr_err):
        gp.set_lambdas(pb["lambdas"][a]); gp.set_sigma_f(np.array(pb["sigma_f"][a])); gp.set_sigma_n(np.array(pb["sigma_n"][a]))
    mpc.dynamics.append_train_data(pb["X"][:, :cfg["ds"]], pb["X"][:, cfg["ds"]:], pb["Y"])
    mpc.curr_state = torch.as_tensor(pb["x0"][0], device=mpc.device)
    rng = np.random.default_rng(0)
    xs = [rng.uniform(-1, 1, cfg["H"] * cfg["da"]) for _ in range(420)]
    for x in xs[:20]:
        mpc.objective(x); mpc.gradient(x)
    pr = cProfile.Profile(); pr.enable()
    for x in xs[20:]:
        mpc.objective(x); mpc.gradient(x)
    pr.disable()
    print("==", cid, "captures", g.lib().gpmpc_pack_graph_captures(mpc.dynamics.pack().handle))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
