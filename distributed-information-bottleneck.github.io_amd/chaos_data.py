"""Trajectory generators for the chaotic maps of the reference's chaos experiments (reference chaos/chaos_data.py:3-55):
`generate_data(system_name, number_iterations, number_skip_iterations, **system_params)` with the same system names,
parameter names and defaults.  Host-side NumPy (a serial recurrence); the trajectories feed the measurement models
and, once symbolised, `dib_amd.ctw.estimate_entropy`.

Differences from the reference, both deliberate: the trajectory is written into a preallocated array (the reference
appends 1.1 M Python lists and stacks them), and an optional `seed` makes the random initial condition reproducible
(the reference draws it from NumPy's global state; `seed=None` keeps that behaviour)."""
from __future__ import annotations

import numpy as np

SYSTEMS = ("logistic", "henon", "ikeda")


def _initial(rng, n):
    return rng.random(n) if rng is not None else np.random.rand(n)


def generate_data(system_name, number_iterations=1_000_000, number_skip_iterations=100_000, seed=None, **system_params):
    """Returns [number_iterations, state_dimensionality] float64 after discarding the transient."""
    rng = np.random.default_rng(seed) if seed is not None else None
    total = int(number_iterations) + int(number_skip_iterations)
    if system_name == "logistic":                      # chaos_data.py:17-25
        r = system_params.get("r", 3.7115)
        out = np.empty((total, 1))
        x = float(_initial(rng, 1)[0])
        for i in range(total):
            out[i, 0] = x
            x = x * (1.0 - x) * r
    elif system_name == "henon":                       # chaos_data.py:26-35
        a, b = system_params.get("a", 1.4), system_params.get("b", 0.3)
        out = np.empty((total, 2))
        x, y = (float(v) for v in _initial(rng, 2))
        for i in range(total):
            out[i, 0], out[i, 1] = x, y
            x, y = 1 - a * x ** 2 + b * y, x
    elif system_name == "ikeda":                       # chaos_data.py:36-52 (notation of Davidchack et al. 2000)
        a, b = system_params.get("a", 1.0), system_params.get("b", 0.9)
        kappa, eta = system_params.get("kappa", 0.4), system_params.get("eta", 6)
        out = np.empty((total, 2))
        x, y = (float(v) for v in _initial(rng, 2))
        for i in range(total):
            out[i, 0], out[i, 1] = x, y
            phi = kappa - eta / (1.0 + x ** 2 + y ** 2)
            x, y = a + b * (x * np.cos(phi) - y * np.sin(phi)), b * (x * np.sin(phi) + y * np.cos(phi))
    else:
        raise ValueError(f"System {system_name} not implemented.")
    return out[int(number_skip_iterations):]
