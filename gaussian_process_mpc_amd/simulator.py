"""Closed loop around the accelerated path: the reference's ``Simulator.run`` (src/simulator.py:37-60) for any
environment with the gym step API, plus a dependency-free pendulum plant so the loop can run where ``gym`` is not
installed.  (the caller of the path; rendering / video recording are out of scope.)"""
import numpy as np


class Simulator(object):
    """env: anything with ``reset() -> (obs, info)`` and ``step(action) -> (obs, reward, terminated, truncated, info)``."""

    def __init__(self, mpc, env, num_iters=500, record=False, video_folder=None, name_prefix=None, incremental=False):
        if record:
            raise NotImplementedError("video recording needs gym's RecordVideo wrapper: wrap the env before passing it in")
        self.mpc = mpc
        self.env = env
        self.num_iters = num_iters
        self.history = []
        self.incremental = incremental      # O(N^2) Ky_inv append per step instead of the reference's O(N^3) rebuild

    def run(self):
        obs, _ = self.env.reset()
        for _ in range(self.num_iters):
            action = self.mpc.get_optimal_trajectory(obs)[0, :]                  # src/simulator.py:47
            next_obs, reward, terminated, truncated, _ = self.env.step(action)   # :48
            self.history.append((np.array(obs), np.array(action), float(reward)))
            if terminated or truncated:
                break
            self.mpc.dynamics.append_train_data(obs, action, next_obs, incremental=self.incremental)   # :55
            obs = next_obs
        if hasattr(self.env, "close"):
            self.env.close()
        return self.history


class PendulumPlant(object):
    """The pendulum update of the reference's adjustable pendulum (src/environments/adjustable_pendulum.py:135-156),
    state (theta, theta_dot) observed directly, without gym / pygame."""

    def __init__(self, g=10.0, m=1.0, l=1.0, dt=0.05, max_torque=2.0, max_speed=8.0, init_state=(np.pi, 0.0)):
        self.g, self.m, self.l, self.dt = g, m, l, dt
        self.max_torque, self.max_speed = max_torque, max_speed
        self.init_state = np.array(init_state, dtype=np.float64)
        self.state = self.init_state.copy()

    def reset(self):
        self.state = self.init_state.copy()
        return self.state.copy(), {}

    def step(self, u):
        th, thdot = self.state
        u = float(np.clip(u, -self.max_torque, self.max_torque)[0])
        wrapped = ((th + np.pi) % (2 * np.pi)) - np.pi
        cost = wrapped ** 2 + 0.1 * thdot ** 2 + 0.001 * u ** 2
        thdot = np.clip(thdot + (3 * self.g / (2 * self.l) * np.sin(th) + 3.0 / (self.m * self.l ** 2) * u) * self.dt,
                        -self.max_speed, self.max_speed)
        self.state = np.array([th + thdot * self.dt, thdot])
        return self.state.copy(), -cost, False, False, {}
