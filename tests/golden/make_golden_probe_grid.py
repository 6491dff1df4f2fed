"""Fixture for the per-particle information map (SURVEY 8(f) rank 4 remainder: probe-grid MI bounds) from the reference
NOTEBOOK's own statements, executed on the NumPy stand-in for TensorFlow (tests/golden/tf_numpy_shim.py).  Run here only:
    python tests/golden/make_golden_probe_grid.py
The body of the innermost loop of code cell 8's "Now use probe points along with a bunch of real points to get the info for
points on a grid" block (from `sampled_u_probes = tf.random.normal(` to `loo_per = ...`; notebook lines ~548-566) is taken
from the .ipynb verbatim (dedented, nothing is copied into this repository) and run on small random Gaussians.  Writes
tests/golden/probe_grid_bounds.npz."""
import json
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import tf_numpy_shim as tf  # noqa: E402

NB = "/root/reference/complex_systems/InfoDecomp_Amorphous_plasticity_per_particle_measurements_and_set_transformer.ipynb"


def main():
    nb = json.load(open(NB))
    cells = ["".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code"]
    big = next(c for c in cells if "set_transformer = tf.keras.Model(inp, x)" in c)
    start = big.index("              sampled_u_probes = tf.random.normal(")
    end = big.index("              upper_bounds_per.append(loo_per)")
    body = textwrap.dedent(big[start:end])
    body = "\n".join(l for l in body.splitlines() if not l.strip().startswith("lower_bounds_per.append"))
    rng = np.random.default_rng(5)
    M, N, E = 9, 40, 6
    mus_probes = rng.standard_normal((M, E)) * 1.5
    logvars_probes = rng.standard_normal((M, E)) * 0.5 - 3.0          # already includes the -3 offset (nb: "- 3")
    mus_data = rng.standard_normal((N, E)) * 1.5
    logvars_data = rng.standard_normal((N, E)) * 0.5 - 3.0
    mus_data[:3] = mus_probes[:3] + 0.05 * rng.standard_normal((3, E))   # some data near the probes: non-trivial ratios
    eps = rng.standard_normal((M, E))
    tf.push_eps([eps])
    g = {"tf": tf, "np": np, "mus_probes": mus_probes, "logvars_probes": logvars_probes,
         "stddevs_probes": np.exp(logvars_probes / 2.), "mus_data": mus_data, "logvars_data": logvars_data,
         "stddevs_data": np.exp(logvars_data / 2.), "probe_ind_start": 0, "probe_ind_end": M, "embedding_dimension": E,
         "normalization_factor": (2. * np.pi) ** (E / 2.)}
    exec(compile(body, "nb:cell8[probe grid inner loop]", "exec"), g)
    np.savez_compressed(os.path.join(HERE, "probe_grid_bounds.npz"), mus_probes=mus_probes, logvars_probes=logvars_probes,
                        mus_data=mus_data, logvars_data=logvars_data, eps=eps, sampled_u_probes=np.asarray(g["sampled_u_probes"]),
                        infonce_per=np.asarray(g["infonce_per"]), loo_per=np.asarray(g["loo_per"]))
    print("infonce_per", np.asarray(g["infonce_per"])[:4], "loo_per", np.asarray(g["loo_per"])[:4])


if __name__ == "__main__":
    main()
