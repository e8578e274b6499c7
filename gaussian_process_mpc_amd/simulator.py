"""Closed loop around the accelerated path: the reference's ``Simulator.run`` (src/simulator.py:37-60) for any
environment with the gym step API, plus dependency-free plants -- the update rules of the reference's two
environments as plain classes -- so the loop can run where ``gym`` / ``pygame`` are not installed.
(the callers of the path; rendering / video recording are out of scope.)
The plants are pinned to the reference's own environment classes by tests/golden/g9_closed_loop.npz."""
import math

import numpy as np


class Simulator(object):
    """env: anything with ``reset() -> (obs, info)`` and ``step(action) -> (obs, reward, terminated, truncated, info)``."""

    def __init__(self, mpc, env, num_iters=500, record=False, video_folder=None, name_prefix=None, incremental=False,
                 async_rebuild=False, refresh=None):
        if record:
            raise NotImplementedError("video recording needs gym's RecordVideo wrapper: wrap the env before passing it in")
        self.mpc = mpc
        self.env = env
        self.num_iters = num_iters
        self.history = []
        self.incremental = incremental      # O(N^2) Ky_inv append per step instead of the reference's O(N^3) rebuild
        self.async_rebuild = async_rebuild  # ... and its periodic full rebuild on a side stream (off the step's critical path)
        self.refresh = refresh              # ... or "newton": that rebuild replaced by a Newton-Schulz polish (0.6 instead of 2.2 ms at N = 400)

    def run(self):
        obs, _ = self.env.reset()
        for _ in range(self.num_iters):
            action = self.mpc.get_optimal_trajectory(obs)[0, :]                  # src/simulator.py:47
            next_obs, reward, terminated, truncated, _ = self.env.step(action)   # :48
            self.history.append((np.array(obs), np.array(action), float(reward)))
            if terminated or truncated:
                break
            self.mpc.dynamics.append_train_data(obs, action, next_obs, incremental=self.incremental,
                                                async_rebuild=self.async_rebuild if self.incremental else None,
                                                refresh=self.refresh if self.incremental else None)               # :55
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

    @staticmethod
    def step_static(state, u, options):
        """Stateless update (src/environments/adjustable_pendulum.py:158-178): ``options`` holds g, m, l, dt,
        max_torque, max_speed; ``u`` is array-like (its first entry is the torque)."""
        th, thdot = state[0], state[1]
        g, m, l, dt = options["g"], options["m"], options["l"], options["dt"]
        u = np.clip(u, -options["max_torque"], options["max_torque"])[0]
        newthdot = thdot + (3 * g / (2 * l) * np.sin(th) + 3.0 / (m * l ** 2) * u) * dt
        newthdot = np.clip(newthdot, -options["max_speed"], options["max_speed"])
        return np.array([th + newthdot * dt, newthdot])

    def options(self):
        return {"g": self.g, "m": self.m, "l": self.l, "dt": self.dt, "max_torque": self.max_torque, "max_speed": self.max_speed}

    def step(self, u):
        th, thdot = self.state
        u = float(np.clip(u, -self.max_torque, self.max_torque)[0])
        wrapped = ((th + np.pi) % (2 * np.pi)) - np.pi
        cost = wrapped ** 2 + 0.1 * thdot ** 2 + 0.001 * u ** 2
        thdot = np.clip(thdot + (3 * self.g / (2 * self.l) * np.sin(th) + 3.0 / (self.m * self.l ** 2) * u) * self.dt,
                        -self.max_speed, self.max_speed)
        self.state = np.array([th + thdot * self.dt, thdot])
        return self.state.copy(), -cost, False, False, {}


class CartPolePlant(object):
    """The continuous cart-pole of the reference (src/environments/continuous_cartpole.py): semi-implicit Euler update
    ``stepPhysics`` (:71-87), ``step`` (:89-101: force = force_mag * action, reward 1, never terminates), ``reset``
    (:129-132: uniform(-0.2, 0.2) start), without gym / pygame.  State (x, x_dot, theta, theta_dot)."""

    def __init__(self, seed=None, init_state=None):
        self.gravity, self.masscart, self.masspole = 9.8, 1.0, 0.1
        self.total_mass = self.masspole + self.masscart
        self.length = 0.5                                # half the pole's length
        self.polemass_length = self.masspole * self.length
        self.force_mag, self.tau = 30.0, 0.02
        self.min_action, self.max_action = -1.0, 1.0
        self.np_random = np.random.default_rng(seed)
        self.init_state = None if init_state is None else np.array(init_state, dtype=np.float64)
        self.state = None

    def stepPhysics(self, force, state=None):
        x, x_dot, theta, theta_dot = self.state if state is None else state
        costheta, sintheta = math.cos(theta), math.sin(theta)
        temp = (force + self.polemass_length * theta_dot * theta_dot * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / \
            (self.length * (4.0 / 3.0 - self.masspole * costheta * costheta / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        return (x + self.tau * x_dot, x_dot + self.tau * xacc, theta + self.tau * theta_dot, theta_dot + self.tau * thetaacc)

    def step(self, action):
        a = float(np.asarray(action, dtype=np.float64).reshape(-1)[0])
        if not (-1 < a < 1):
            raise AssertionError("action must lie strictly inside (-1, 1)")      # continuous_cartpole.py:92
        self.state = self.stepPhysics(self.force_mag * a)
        return np.array(self.state), 1, False, False, {}

    def reset(self):
        self.state = self.np_random.uniform(low=-0.2, high=0.2, size=(4,)) if self.init_state is None else self.init_state.copy()
        return np.array(self.state), None
