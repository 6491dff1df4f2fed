"""Double-pendulum trajectory generator (offline data generation for BASELINE config 2; CPU, scipy).

Same contract as the reference's simulate_pendulum.py (`simulate_double_pendulum(data_path, simulation_params_dict)`
writes `double_pendulum.npy` of shape [number_trajectories, T, 4] = (theta1, dtheta1, theta2, dtheta2); reference
simulate_pendulum.py:10-95): unit masses/lengths by default, every trajectory starts at rest with total potential
energy `energy_over_g`*g, a burn-in of `initial_time` is discarded, trajectories whose energy drifts more than 1e-3
are rejected.  The integrator and the Lagrangian mechanics are written from the textbook equations of motion.
"""
from __future__ import annotations

import os

import numpy as np
from scipy.integrate import odeint

G = 9.81


def _rhs(state, _t, l1, l2, m1, m2):
    th1, w1, th2, w2 = state
    dlt = th1 - th2
    c, s = np.cos(dlt), np.sin(dlt)
    den = m1 + m2 * s * s
    a1 = (m2 * G * np.sin(th2) * c - m2 * s * (l1 * w1 * w1 * c + l2 * w2 * w2) - (m1 + m2) * G * np.sin(th1)) / (l1 * den)
    a2 = ((m1 + m2) * (l1 * w1 * w1 * s - G * np.sin(th2) + G * np.sin(th1) * c) + m2 * l2 * w2 * w2 * s * c) / (l2 * den)
    return w1, a1, w2, a2


def total_energy(y, l1=1.0, l2=1.0, m1=1.0, m2=1.0):
    th1, w1, th2, w2 = np.asarray(y).T
    pot = -(m1 + m2) * l1 * G * np.cos(th1) - m2 * l2 * G * np.cos(th2)
    kin = 0.5 * m1 * (l1 * w1) ** 2 + 0.5 * m2 * ((l1 * w1) ** 2 + (l2 * w2) ** 2 + 2 * l1 * l2 * w1 * w2 * np.cos(th1 - th2))
    return kin + pot


def simulate_double_pendulum(data_path='./data/', simulation_params_dict=None, rng=None, save=True):
    prm = dict(simulation_params_dict or {})
    m1, m2, l1, l2 = prm.get('m1', 1), prm.get('m2', 1), prm.get('L1', 1), prm.get('L2', 1)
    energy_over_g = prm.get('energy_over_g', 4)
    initial_time, simulation_time = prm.get('initial_time', 50), prm.get('simulation_time', 50)
    dt_sim, dt_save = prm.get('dt_simulation', 1e-2), prm.get('dt_saving', 2e-2)
    n_traj = prm.get('number_trajectories', 1000)
    rng = rng or np.random.default_rng()
    every = int(dt_save // dt_sim)
    t = np.linspace(0, initial_time + simulation_time, int((initial_time + simulation_time) // dt_sim))
    runs, rejected = [], 0
    while len(runs) < n_traj:
        th1 = rng.uniform() * 2 * np.pi
        h1 = l1 * (1.0 - np.cos(th1))
        arg = 1 - ((energy_over_g - m1 * h1) / m2 - h1) / l2  # second arm angle giving the prescribed potential energy
        sign = rng.integers(2) * 2 - 1   # drawn before the validity test, like the reference (simulate_pendulum.py:63-66 evaluates
        if not -1.0 <= arg <= 1.0:       # arccos(...) * (randint(2)*2-1) and only then checks for NaN): same stream consumption
            continue
        th2 = np.arccos(arg) * sign
        y0 = np.array([th1, 0.0, th2, 0.0])
        y = odeint(_rhs, y0, t, args=(l1, l2, m1, m2))
        e0 = total_energy(y0[None], l1, l2, m1, m2)[0]
        if np.max(np.abs(total_energy(y, l1, l2, m1, m2) - e0) / np.abs(e0)) > 1e-3:
            rejected += 1
            continue
        runs.append(y[int(initial_time // dt_sim)::every])
    arr = np.stack(runs, 0)
    if save:
        os.makedirs(data_path, exist_ok=True)
        np.save(os.path.join(data_path, 'double_pendulum.npy'), arr)
    return arr


if __name__ == '__main__':
    simulate_double_pendulum()
