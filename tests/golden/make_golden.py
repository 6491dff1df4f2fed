"""Generate golden fixtures by EXECUTING reference code (run in the build container only).

The reference's modules import tensorflow at module scope (absent here), so the numpy-only
functions are lifted out of the reference sources by AST at generation time and executed with a
stub `tf` namespace.  Nothing from the reference is copied into the repo - only the numeric
outputs are stored (tests/golden/*.npz).  /root/reference does not exist on the GPU box, so tests
read only the committed .npz files.

    python tests/golden/make_golden.py
"""
import ast
import os
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def lift(path, names, extra_globals=None):
    src = open(path).read()
    tree = ast.parse(src)
    g = {"np": np}
    g.update(extra_globals or {})
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, g)
    return g


def main():
    # ---- utils.py:177-246 Bhattacharyya / KL matrices ---------------------------------------
    g = lift(os.path.join(REF, "utils.py"), {"bhattacharyya_dist_mat", "kl_divergence_mat",
                                             "compute_entropy_bits", "compute_entropy"})
    rng = np.random.default_rng(7)
    mus1 = rng.standard_normal((9, 6))
    lvs1 = 0.7 * rng.standard_normal((9, 6))
    mus2 = rng.standard_normal((5, 6))
    lvs2 = 0.7 * rng.standard_normal((5, 6))
    bh = g["bhattacharyya_dist_mat"](mus1, lvs1, mus2, lvs2)
    bh_self = g["bhattacharyya_dist_mat"](mus1, lvs1, mus1, lvs1)
    kl = g["kl_divergence_mat"](mus1, lvs1, mus2, lvs2)
    probs = rng.dirichlet(np.ones(7))
    ent = g["compute_entropy_bits"](probs)
    np.savez(os.path.join(OUT, "utils_distance_mats.npz"), mus1=mus1, lvs1=lvs1, mus2=mus2, lvs2=lvs2,
             bhattacharyya=bh, bhattacharyya_self=bh_self, kl=kl, probs=probs, entropy_bits=ent)

    # ---- data.py:21-81 Boolean circuit --------------------------------------------------------
    class _Loss:
        def __init__(self, *a, **k):
            self.kw = k
    tf = types.SimpleNamespace(keras=types.SimpleNamespace(losses=types.SimpleNamespace(
        BinaryCrossentropy=_Loss, SparseCategoricalCrossentropy=_Loss)))
    g = lift(os.path.join(REF, "data.py"), {"fetch_boolean_circuit"}, {"tf": tf})
    d = g["fetch_boolean_circuit"]()
    p1 = float(np.mean(d["y_train"]))
    hy = -(p1 * np.log2(p1) + (1 - p1) * np.log2(1 - p1))
    np.savez(os.path.join(OUT, "boolean_circuit.npz"), x_train=d["x_train"], y_train=d["y_train"],
             feature_dimensionalities=np.array(d["feature_dimensionalities"]),
             output_dimensionality=d["output_dimensionality"], entropy_y_bits=hy,
             loss_from_logits=bool(d["loss"].kw.get("from_logits", False)),
             loss_is_info_based=bool(d["loss_is_info_based"]))
    print("H(Y) bits", hy, "P(y=1)", p1, "rows", d["x_train"].shape)


if __name__ == "__main__":
    main()
