"""Golden fixture of the exhaustive subset informations I(X_S;Y) of the paper's 10-input Boolean circuit: the statements of
the reference notebook complex_systems/InfoDecomp_Boolean_circuits.ipynb are EXECUTED, not restated -
  * the circuit cell (gates, circuit_specification, apply_gates, compute_entropy, compute_info, truth table, entropy_y) and
  * the "Let's look at the information in all subsets" statements (all_on_off_combos, 'ij' meshgrid, all_mis)
are cut out of the notebook's code cells by their source text at generation time.  Only the numbers are stored
(tests/golden/subset_mi.npz).  Run in the build container only (/root/reference is absent on the GPU box):

    python tests/golden/make_golden_subset_mi.py
"""
import json
import os

import numpy as np

NB = "/root/reference/complex_systems/InfoDecomp_Boolean_circuits.ipynb"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    cells = ["".join(c["source"]) for c in json.load(open(NB))["cells"] if c["cell_type"] == "code"]
    circuit = next(s for s in cells if "def compute_info" in s and "circuit_specification = [0, 1, 2" in s)
    circuit = circuit[circuit.index("gates = [np.logical_and"):]          # from the gate table on (skips the imports)
    circuit = circuit[:circuit.index("print(f'Entropy of Y")]
    g = {"np": np}
    exec(compile(circuit, NB, "exec"), g)
    train = next(s for s in cells if "Let's look at the information in all subsets" in s)
    a = train.index("all_on_off_combos = np.meshgrid")
    b = train.index("plt.figure", a)
    body = "\n".join(line[2:] if line.startswith("  ") else line for line in train[a:b].split("\n"))   # the cell body is indented
    exec(compile(body, NB, "exec"), g)
    combos = np.asarray(g["all_on_off_combos"], dtype=np.int8)
    mis = np.asarray(g["all_mis"], dtype=np.float64)
    assert combos.shape == (1024, 10) and mis.shape == (1024,)
    np.savez(os.path.join(OUT, "subset_mi.npz"), all_on_off_combos=combos, all_mis_bits=mis,
             entropy_y_bits=np.float64(g["entropy_y"]), truth_table=np.asarray(g["truth_table"], dtype=np.int8))
    print("H(Y)", g["entropy_y"], "I(all;Y)", mis[-1], "max single", mis[combos.sum(1) == 1].max())


if __name__ == "__main__":
    main()
