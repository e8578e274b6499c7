#!/usr/bin/env python3
"""The experiment behind the reference's README figures (src/experiments/pretrain_uncertainty.py:82-121): 2-D state,
2-D action, true dynamics s' = s + a, 400 training transitions along an L-shaped corridor from (4, -4) over (4, 0) to
(0, 0); hyper-parameters lambda = 0.5, sigma_f = 1, sigma_n = 1e-5; Q = 2 I, R = 0, horizon 6, start (4, -4), target 0.
Risk-averse (gamma = -1) plans along the corridor, risk-neutral (gamma = 1e-5) cuts across where there is no data.

    python examples/readme_uncertainty_experiment.py [--steps 6]

The training triple is the reference's own data file set, stored as arrays in tests/golden/g9_closed_loop.npz.
Needs an MI355X and the built library; the solver is cyipopt when importable, else scipy's L-BFGS-B on the same
callbacks (so trajectories are not the reference's Ipopt iterates)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_mpc_amd import RiskSensitiveMPC   # noqa: E402


def build_mpc(gamma, data, horizon=6):
    mpc = RiskSensitiveMPC(gamma, horizon, 2, 2, 2 * np.identity(2), np.zeros((2, 2)), None)
    for i in range(2):
        mpc.dynamics.gpr_err[i].set_sigma_n(1e-5)
        mpc.dynamics.gpr_err[i].set_lambdas([0.5, 0.5, 0.5, 0.5])
        mpc.dynamics.gpr_err[i].set_sigma_f(1.)
    mpc.dynamics.append_train_data(data["exp_states"], data["exp_actions"], data["exp_next_states"])
    mpc.set_ub([1, 1]); mpc.set_lb([-1, -1])
    mpc.set_xref(np.array([0., 0.])); mpc.set_uref(np.array([0., 0.]))
    return mpc


def distance_to_data(points, states):
    return np.sqrt(((points[:, None, :] - states[None, :, :]) ** 2).sum(axis=2)).min(axis=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    data = np.load(os.path.join(ROOT, "tests", "golden", "g9_closed_loop.npz"))
    for gamma in (-1.0, 1e-5):
        mpc = build_mpc(gamma, data)
        s = np.array([4.0, -4.0])
        plan = mpc.get_optimal_trajectory(s)
        r = mpc.evaluate_batch(plan[None], s)
        means = r["means"][0].cpu().numpy()
        print(f"gamma = {gamma:g} ({mpc.solver_used}): planned cost {r['cost'][0].item():.4f} (zero input: {mpc.objective(np.zeros(12)):.4f})")
        print("  planned inputs:", np.array2string(plan, precision=2).replace("\n", ""))
        print("  predicted means:", np.array2string(means, precision=2).replace("\n", ""))
        print("  predicted variances (last step):", r["vars"][0, -1].cpu().numpy())
        print("  mean distance of the predicted states to the training states: %.3f" % distance_to_data(means[1:], data["exp_states"]).mean())
        path = [s.copy()]
        for _ in range(args.steps):                      # closed loop on the true dynamics s' = s + a
            a = mpc.get_optimal_trajectory(s)[0]
            s = s + a
            path.append(s.copy())
        path = np.array(path)
        print("  closed-loop path:", np.array2string(path, precision=2).replace("\n", ""))
        print("  closed-loop distance to data: %.3f, final |s| %.3f" % (distance_to_data(path, data["exp_states"]).mean(), np.linalg.norm(path[-1])))


if __name__ == "__main__":
    main()
