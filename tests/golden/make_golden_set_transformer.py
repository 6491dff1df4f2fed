"""Fixtures for the per-particle set-transformer DIB from the reference NOTEBOOK's own code, executed on
the NumPy stand-in for TensorFlow (tests/golden/tf_numpy_shim.py).  Run here only:
    python tests/golden/make_golden_set_transformer.py
The model-building region of code cell 8 (particle encoder, 6 attention blocks, pooling, head) and the forward statements of
its `train_step` are taken from the .ipynb verbatim (dedented / AST-lifted, nothing is copied into this repository) with
`number_particles_to_use = 7`; weights and noise come from oracle/set_transformer_oracle.py.  Writes
tests/golden/set_transformer_forward.npz.  This pins the WIRING (residual order, LayerNorm placement, pooling axis, KL axes,
the -3 logvar offset, loss composition); the primitives (Dense, MultiHeadAttention, LayerNormalization, BCE) are the stand-in's
own independent NumPy implementations of the Keras semantics."""
import ast
import json
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tf_numpy_shim as tf  # noqa: E402
import set_transformer_oracle as sto  # noqa: E402
import torch  # noqa: E402

NB = "/root/reference/complex_systems/InfoDecomp_Amorphous_plasticity_per_particle_measurements_and_set_transformer.ipynb"


def main():
    nb = json.load(open(NB))
    cells = ["".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code"]
    posenc_src = next(c for c in cells if c.startswith("#@title Positional encoding definition"))
    feat_src = next(c for c in cells if "def convert_to_per_particle_feature_set" in c)
    big = next(c for c in cells if "set_transformer = tf.keras.Model(inp, x)" in c)
    start = big.index("  particle_encoder_arch_spec = [128]*2")
    end = big.index("  ##############################################################################\n  ### Setup for displaying")
    region = textwrap.dedent(big[start:end])

    g = {"tf": tf, "np": np, "number_particles_to_use": 7, "SAFETY_EPS": 1e-10}
    exec(compile(posenc_src, "nb:cell4", "exec"), g)
    n_before = len(tf.ALL_LAYERS)
    exec(compile(region, "nb:cell8[model + train_step]", "exec"), g)
    layers = tf.ALL_LAYERS[n_before:]

    # ---- inject the oracle's parameters in Keras creation order ----
    spec = sto.SetTransformerSpec()
    p = sto.init_params(spec, seed=7)
    rng = np.random.default_rng(11)
    for k in p:                                   # non-trivial biases / LayerNorm parameters
        if k.endswith("_b") or k.endswith("_g"):
            p[k] = p[k] + torch.tensor(0.1 * rng.standard_normal(tuple(p[k].shape)))
    names = list(sto.param_shapes(spec))
    it = iter(names)

    def take(n):
        return [p[next(it)].numpy() for _ in range(n)]
    for L in layers:
        if isinstance(L, tf.Dense):
            L.kernel, L.bias = take(2)
        elif isinstance(L, tf.MultiHeadAttention):
            L.wq, L.bq, L.wk, L.bk, L.wv, L.bv, L.wo, L.bo = take(8)
        elif isinstance(L, tf.LayerNormalization):
            L.gamma, L.beta = take(2)
    assert next(it, None) is None, "parameter order mismatch"

    # ---- lift the forward statements of train_step (the body of `with tf.GradientTape() as tape:`) ----
    fn = next(n for n in ast.parse(region).body if isinstance(n, ast.FunctionDef) and n.name == "train_step")
    with_body = next(n for n in fn.body if isinstance(n, ast.With)).body
    ret = ast.parse("return dict(embs_mus=embs_mus, embs_logvars=embs_logvars, embs_reparam=embs_reparam, kl=kl, "
                    "loci_prediction=loci_prediction, bce_losses=bce_losses, loss=loss)").body[0]
    lifted = ast.FunctionDef(name="train_step_forward", args=fn.args, body=with_body + [ret], decorator_list=[], lineno=1)
    mod = ast.fix_missing_locations(ast.Module(body=[lifted], type_ignores=[]))
    exec(compile(mod, "nb:cell8[train_step forward]", "exec"), g)

    B, P = 5, 7
    feats = np.stack([sto.convert_to_per_particle_feature_set(rng.standard_normal((P + 3, 2)) * 1.5,
                                                              rng.integers(1, 3, P + 3), number_particles_to_use=P)
                      for _ in range(B)]).astype(np.float64)
    is_loci = (rng.random((B, 1)) > 0.5).astype(np.float64)
    eps = rng.standard_normal((B, P, spec.bottleneck_dimension))
    tf.push_eps([eps])
    beta = 0.013
    g["beta_var"].assign(np.float64(np.float32(beta)))            # TF holds beta in float32
    out = g["train_step_forward"](feats, is_loci, False)

    # the notebook's feature function (cell 6) on the same raw positions, executed from its own source
    exec(compile(feat_src, "nb:cell6", "exec"), g)
    raw_pos = (rng.standard_normal((9, 2)) * 1.5).astype(np.float32)
    raw_types = rng.integers(1, 3, 9).astype(np.float32)
    ref_feats = g["convert_to_per_particle_feature_set"](raw_pos, raw_types, number_particles_to_use=6)

    np.savez_compressed(
        os.path.join(HERE, "set_transformer_forward.npz"),
        flat=np.concatenate([p[k].numpy().ravel() for k in names]), feats=feats, is_loci=is_loci, eps=eps,
        beta=np.float64(np.float32(beta)), mu=out["embs_mus"], logvar=out["embs_logvars"], u=out["embs_reparam"],
        kl=np.float64(out["kl"]), pred=out["loci_prediction"], bce=np.float64(out["bce_losses"]), loss=np.float64(out["loss"]),
        raw_pos=raw_pos, raw_types=raw_types, ref_feats=ref_feats)
    print("layers", len(layers), "params", sum(int(np.prod(s)) for s in sto.param_shapes(spec).values()),
          "kl", float(out["kl"]), "bce", float(out["bce_losses"]), "pred", np.ravel(out["loci_prediction"])[:3])


if __name__ == "__main__":
    main()
