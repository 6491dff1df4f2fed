"""Dataset dicts for the hot path, mirroring the reference's `data.DATASETS[name](**kwargs)` contract
(reference data.py:69-81, 397-406).  Shipped: the Boolean circuit (data.py:21-81, pure numpy) and the
synthetic tabular generator BASELINE.md section 4 defines for the north-star benchmark.  The NODE-GAM tabular
fetchers of the reference are stubs returning None and need network: out of scope.
"""
from __future__ import annotations

import numpy as np

from . import losses

PAPER_CIRCUIT = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, [1, 0, 1], [2, 8, 7], [0, 4, 3], [1, 11, 5], [2, 6, 12], [2, 13, 9],
                 [1, 14, 10], [0, 15, 2], [0, 17, 16]]  # reference data.py:40
_GATES = [np.logical_and, np.logical_or, np.logical_xor]


def random_circuit(number_input_gates, rng=None):
    """reference data.py:27-37: repeatedly join two live wires with a random gate until one remains."""
    rng = rng or np.random
    live = list(range(number_input_gates))
    spec = list(range(number_input_gates))
    while len(live) > 1:
        gate = int(rng.choice(len(_GATES)))
        a, b = [int(v) for v in rng.choice(live, size=2, replace=False)]
        live.append(len(spec))
        live.remove(a)
        live.remove(b)
        spec.append([gate, a, b])
    return spec


def truth_table(circuit_specification, number_input_gates):
    """reference data.py:42-53: full truth table in np.meshgrid (default 'xy') order."""
    grids = np.meshgrid(*[[0, 1]] * number_input_gates)
    table = np.reshape(np.stack(grids, -1), [-1, number_input_gates])
    for gate, a, b in circuit_specification[number_input_gates:]:
        table = np.concatenate([table, np.int32(_GATES[gate](table[:, a], table[:, b]))[:, None]], -1)
    return table


def fetch_boolean_circuit(**kwargs):
    """reference data.py:21-81."""
    number_input_gates = kwargs.get('boolean_number_input_gates', 10)
    if kwargs.get('boolean_random_circuit', False):
        spec = random_circuit(number_input_gates)
    else:
        spec, number_input_gates = PAPER_CIRCUIT, 10
    table = truth_table(spec, number_input_gates)
    x_train = 2 * table[:, :number_input_gates] - 1  # x -> {-1, +1}
    y_train = table[:, -1]
    feature_dimensionalities = [1] * number_input_gates
    return dict(x_train=x_train, y_train=y_train, x_valid=x_train, y_valid=y_train,
                feature_dimensionalities=feature_dimensionalities, number_features=len(feature_dimensionalities),
                output_dimensionality=1, output_activation_fn=None,
                loss=losses.BinaryCrossentropy(from_logits=True), loss_is_info_based=True, metrics=['accuracy'],
                circuit_specification=spec)


def fetch_synthetic_tabular(**kwargs):
    """BASELINE.md section 4 / SURVEY 8(d): x ~ N(0,1) [n, F] from default_rng(20241008);
    y = 1[sum_{j<8} w_j x_j + 0.5 x_0 x_1 > 0], w ~ N(0,1) from the same generator; BCE-from-logits."""
    n = int(kwargs.get('synthetic_rows', 1 << 20))
    F = int(kwargs.get('synthetic_features', 64))
    nv = int(kwargs.get('synthetic_valid_rows', min(n, 1 << 16)))
    rng = np.random.default_rng(20241008)
    x = rng.standard_normal((n, F), dtype=np.float32)
    w = rng.standard_normal(8).astype(np.float32)
    k = min(8, F)
    y = ((x[:, :k] @ w[:k] + (0.5 * x[:, 0] * x[:, 1] if F > 1 else 0.0)) > 0).astype(np.float32)
    return dict(x_train=x, y_train=y, x_valid=x[:nv], y_valid=y[:nv], feature_dimensionalities=[1] * F,
                number_features=F, output_dimensionality=1, output_activation_fn=None,
                loss=losses.BinaryCrossentropy(from_logits=True), loss_is_info_based=True, metrics=['accuracy'])


def fetch_double_pendulum(**kwargs):
    """reference data.py:83-147: predict the state `pendulum_time_delta` seconds ahead; 4 features
    (theta1 as a unit vector [2], dtheta1 [1], theta2 as a unit vector [2], dtheta2 [1]); loss 'infonce'.
    Generates the trajectories on first use (simulate_pendulum, `pendulum_number_trajectories` to keep it small).
    Defect A11 of SURVEY App. A (the reference's np.split makes the FIRST 10 % the training set) is not replicated:
    10 % of the trajectories are held out for validation."""
    import os
    data_path = kwargs.get('data_path', './data/')
    fname = os.path.join(data_path, 'double_pendulum.npy')
    if not os.path.exists(fname):
        from . import simulate_pendulum
        prm = {}
        if kwargs.get('pendulum_number_trajectories'):
            prm['number_trajectories'] = int(kwargs['pendulum_number_trajectories'])
        simulate_pendulum.simulate_double_pendulum(data_path=data_path, simulation_params_dict=prm,
                                                   rng=np.random.default_rng(kwargs.get('seed', 0)))
    arr = np.load(fname)
    time_delta = kwargs.get('pendulum_time_delta', 2.)

    def preprocess(a):  # data.py:100-107: angles -> (sin, -cos), velocities as is
        return np.stack([np.sin(a[:, :, 0]), -np.cos(a[:, :, 0]), a[:, :, 1], np.sin(a[:, :, 2]), -np.cos(a[:, :, 2]),
                         a[:, :, 3]], -1)

    n_valid = max(1, int(arr.shape[0] * 0.1))
    train, valid = preprocess(arr[n_valid:]), preprocess(arr[:n_valid])
    steps = int(time_delta / 0.02)  # dt_saving of the simulation (data.py:116-117)
    pair = lambda a: (a[:, :-steps].reshape(-1, 6).astype(np.float32), a[:, steps:].reshape(-1, 6).astype(np.float32))
    x_train, y_train = pair(train)
    x_valid, y_valid = pair(valid)
    dims = [2, 1, 2, 1]
    return dict(x_train=x_train, y_train=y_train, x_valid=x_valid, y_valid=y_valid, feature_dimensionalities=dims,
                number_features=len(dims), output_dimensionality=6, output_activation_fn=None, loss='infonce',
                loss_is_info_based=True, feature_labels=['theta1', 'theta1_dot', 'theta2', 'theta2_dot'])


DATASETS = {'boolean_circuit': fetch_boolean_circuit, 'synthetic_tabular': fetch_synthetic_tabular,
            'double_pendulum': fetch_double_pendulum}
