#!/usr/bin/env python3
"""Closed-loop risk-sensitive GP-MPC on the pendulum plant, the shape of the reference's
src/experiments/pretrain_uncertainty.py: pre-train the GP on random transitions, then run the MPC loop
(Simulator.run, src/simulator.py:37-60) with the model growing by one observation per step.

    python examples/pendulum_closed_loop.py [--pretrain 200] [--steps 25] [--horizon 10]

Needs an MI355X and the built library; no gym, no cyipopt (the stand-in solver is scipy's L-BFGS-B on the same
objective / gradient callbacks, so the trajectories are NOT the reference's Ipopt trajectories)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_mpc_amd import PendulumPlant, RiskSensitiveMPC, Simulator   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrain", type=int, default=200)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--gamma", type=float, default=1e-5)
    ap.add_argument("--starts", type=int, default=1, help="K > 1: lock-step multi-start solve, one batched rollout per tick (mpc.n_starts)")
    args = ap.parse_args()

    rng = np.random.default_rng(0)
    plant = PendulumPlant()
    mpc = RiskSensitiveMPC(args.gamma, args.horizon, 2, 1, Q=2 * np.eye(2), R=0.001 * np.eye(1))
    for gp in mpc.dynamics.gpr_err:                      # hypers before data, as in pretrain_uncertainty.py:100-105
        gp.set_lambdas(np.array([0.5, 0.5, 0.5]))
        gp.set_sigma_n(1e-3)
    # random transitions around the whole state space
    S = np.column_stack((rng.uniform(-np.pi, np.pi, args.pretrain), rng.uniform(-8, 8, args.pretrain)))
    A = rng.uniform(-2, 2, (args.pretrain, 1))
    NS = np.empty_like(S)
    for i in range(args.pretrain):
        plant.state = S[i].copy()
        NS[i] = plant.step(A[i])[0]
    mpc.dynamics.append_train_data(S, A, NS)
    mpc.set_lb([-2.0]); mpc.set_ub([2.0])
    mpc.set_xref(np.zeros(2))
    mpc.n_starts = args.starts

    sim = Simulator(mpc, plant, num_iters=args.steps, incremental=True)
    t0 = time.perf_counter()
    hist = sim.run()
    dt = time.perf_counter() - t0
    th = np.array([h[0][0] for h in hist])
    print(f"{len(hist)} MPC steps in {dt:.2f} s ({dt / len(hist) * 1e3:.1f} ms per step, solver: {mpc.solver_used}); "
          f"training set {args.pretrain} -> {mpc.dynamics.gpr_err[0].num_train} points")
    print("theta:", np.array2string(th[:: max(1, len(th) // 10)], precision=2))


if __name__ == "__main__":
    main()
