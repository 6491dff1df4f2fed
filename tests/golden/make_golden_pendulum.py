"""Golden fixture for the double-pendulum data path (BASELINE config 2): EXECUTES the reference's simulate_pendulum.py (numpy /
scipy only, imported as it is) with a seeded global NumPy stream and short times, then the reference's
data.fetch_double_pendulum (lifted by AST: data.py imports tensorflow / nodegam at module scope) on the file it wrote.
Build container only; stores numbers, no reference source.

    python tests/golden/make_golden_pendulum.py   ->  tests/golden/pendulum.npz
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, lift  # noqa: E402

SEED = 20241008
PARAMS = dict(number_trajectories=10, initial_time=1.0, simulation_time=2.0, dt_simulation=1e-2, dt_saving=2e-2)
TIME_DELTA = 0.4


def main():
    spec = importlib.util.spec_from_file_location("ref_simulate_pendulum", os.path.join(REF, "simulate_pendulum.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    with tempfile.TemporaryDirectory() as d:
        np.random.seed(SEED)
        with np.errstate(invalid="ignore"):
            ref.simulate_double_pendulum(data_path=d, simulation_params_dict=dict(PARAMS))
        traj = np.load(os.path.join(d, "double_pendulum.npy"))
        g = lift(os.path.join(REF, "data.py"), {"fetch_double_pendulum"}, {"os": os})
        out = g["fetch_double_pendulum"](data_path=d, pendulum_time_delta=TIME_DELTA)
    np.savez_compressed(os.path.join(HERE, "pendulum.npz"), seed=SEED, time_delta=TIME_DELTA, trajectories=traj,
                        x_train=out["x_train"], y_train=out["y_train"], x_valid=out["x_valid"], y_valid=out["y_valid"],
                        feature_dimensionalities=np.array(out["feature_dimensionalities"]),
                        **{"param_" + k: v for k, v in PARAMS.items()})
    print("trajectories", traj.shape, "x_train", out["x_train"].shape, "x_valid", out["x_valid"].shape, out["loss"])


if __name__ == "__main__":
    main()
