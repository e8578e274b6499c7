"""Seeded synthetic GP-MPC workloads: identical inputs for the oracle,
the CPU baseline and the GPU path.  numpy only."""
import numpy as np

# BASELINE.json configs -> (N, ds, da, H, B, gamma).  d == state_dim; da = 1 except the toy config.
CONFIGS = {
    "C1": dict(N=100, ds=2, da=2, H=10, B=1, gamma=1e-5),
    "C2": dict(N=512, ds=3, da=1, H=20, B=1, gamma=-1.0),
    "C3": dict(N=2048, ds=4, da=1, H=20, B=256, gamma=-1.0),
    "C4": dict(N=4096, ds=6, da=1, H=30, B=1024, gamma=-1.0),
    "C5": dict(N=2048, ds=4, da=1, H=20, B=256, gamma=-1.0),
}


def synth_problem(config_id, N, ds, da, H, B, sigma_n=1e-2, lam_range=(2.0, 6.0), shared_lambda=False):
    """states ~ U(-2,2), actions ~ U(-1,1), next = s + 0.1 tanh(s) + 0.1 sum(a) (smooth, bounded);
    lambdas ~ U(lam_range) per GP and dimension (non-proportional on purpose); sigma_f = 1;
    x0 ~ U(-1,1), U ~ U(-1,1); Q = 0.1 I, R = 0.01 I (so that 1 + gamma Q var > 0 for gamma = -1).
    shared_lambda: every GP gets the length-scales drawn for GP 0 -- the regime of the reference's own experiments
    (one lambda for all GPs: src/experiments/pretrain_uncertainty.py:100-105, pretrain_pendulum.py:54-55)."""
    rng = np.random.default_rng(1000 + config_id)
    D = ds + da
    S = rng.uniform(-2, 2, size=(N, ds))
    A = rng.uniform(-1, 1, size=(N, da))
    Y = S + 0.1 * np.tanh(S) + 0.1 * A.sum(axis=1, keepdims=True)
    lam = rng.uniform(lam_range[0], lam_range[1], size=(ds, D))
    if shared_lambda:
        lam = np.tile(lam[:1], (ds, 1))
    return {
        "X": np.concatenate((S, A), axis=1), "Y": Y, "lambdas": lam,
        "sigma_f": np.ones(ds), "sigma_n": np.full(ds, sigma_n),
        "x0": rng.uniform(-1, 1, size=(B, ds)), "U": rng.uniform(-1, 1, size=(B, H, da)),
        "Q": 0.1 * np.eye(ds), "R": 0.01 * np.eye(da),
        "x_ref": np.zeros(ds), "u_ref": np.zeros(da),
        "N": N, "ds": ds, "da": da, "H": H, "B": B,
    }
