"""Fixture pinning the CONTENT of the compression-matrix figure (reference visualization.py:14-81,
models.py:175-186): the reference's own `save_compression_matrices` is lifted from its source by AST and EXECUTED with a
recording stand-in for matplotlib (tests/golden/plt_recorder.py), a two-function stand-in for the two TensorFlow calls it
makes (tf.gather, tf.split) and the reference's own utils.bhattacharyya_dist_mat (also lifted).  Run here only:
    python tests/golden/make_golden_compression.py
Two cases: (a) < 10 unique raw values (histogram mode, fully deterministic); (b) a continuous feature (random selection of
128 rows under np.random.seed).  Reference defects handled explicitly (SURVEY App. A style, follow intent):
  A2  `n` undefined in the continuous branch  -> n = len(sorted_features_raw) supplied as a global;
  A14 (found here) the continuous branch gathers `inp_features` with np.argsort(raw[random_selection_inds]) - indices INTO
      THE SELECTION used as dataset row numbers, so the matrix it draws belongs to the first 128 dataset rows, not to the
      sorted selection shown beside it.  The fixture stores both: `b_matrix_literal` (what the reference literally draws)
      and `b_matrix_intent` (the same statements executed with the one-line in-memory fix
      feature_inp_inds = random_selection_inds[np.argsort(...)]), which is what this project draws.
Nothing from the reference is copied into the repository; only the recorded arrays are stored."""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from plt_recorder import Recorder  # noqa: E402

REF = "/root/reference"


def fake_encoder(x):
    """deterministic stand-in for model.feature_encoders[f]: [n, d] -> [n, 2E] (mu | logvar), E = 5"""
    x = np.asarray(x, dtype=np.float64).reshape(len(x), -1)
    k = np.arange(1, 6, dtype=np.float64)
    return np.concatenate([np.sin(x[:, :1] * k), -1.0 + 0.5 * np.cos(x[:, :1] * k[::-1])], -1)


def lift(path, name, g, patch=None):
    src = open(path).read()
    if patch:
        assert patch[0] in src
        src = src.replace(patch[0], patch[1])
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), g)
    return g[name]


def run(raw, inp, seed=None, patch=None, n=None):
    rec = Recorder()
    gu = {"np": np}
    bhat = lift(os.path.join(REF, "utils.py"), "bhattacharyya_dist_mat", gu)
    tf = types.SimpleNamespace(gather=lambda x, idx, axis=0: np.take(np.asarray(x), np.asarray(idx), axis=axis),
                               split=lambda v, k, axis=-1: np.split(np.asarray(v), k, axis=axis))
    g = {"np": np, "plt": rec, "tf": tf, "utils": types.SimpleNamespace(bhattacharyya_dist_mat=bhat), "n": n}
    fn = lift(os.path.join(REF, "visualization.py"), "save_compression_matrices", g, patch)
    if seed is not None:
        np.random.seed(seed)
    fn(fake_encoder, inp, "unused.png", inp_features_raw=raw, feature_label="Feature 3")
    return rec.content()


def main():
    rng = np.random.default_rng(3)
    # (a) histogram mode: 4 distinct raw values, processed inputs = a monotone transform of them
    raw_a = rng.choice(np.array([-1.5, 0.0, 0.5, 2.0]), size=300)
    inp_a = (raw_a / 2.0)[:, None]
    ca = run(raw_a, inp_a)
    # (b) continuous feature
    raw_b = rng.standard_normal(1000)
    inp_b = np.tanh(raw_b)[:, None]
    cb_lit = run(raw_b, inp_b, seed=123, n=128)
    cb_int = run(raw_b, inp_b, seed=123, n=128,
                 patch=("feature_inp_inds = np.argsort(inp_features_raw[random_selection_inds])",
                        "feature_inp_inds = random_selection_inds[np.argsort(inp_features_raw[random_selection_inds])]"))
    k_im, k_l, k_t = ((1, 1), "imshow"), ((1, 0), None), ((0, 1), None)
    np.savez_compressed(
        os.path.join(HERE, "compression_matrix_figure.npz"), raw_a=raw_a, inp_a=inp_a, raw_b=raw_b, inp_b=inp_b,
        a_matrix=ca[k_im][0], a_barh_y=ca[((1, 0), "barh")][0], a_barh_w=ca[((1, 0), "barh")][1],
        a_bar_x=ca[((0, 1), "bar")][0], a_bar_h=ca[((0, 1), "bar")][1],
        b_matrix_literal=cb_lit[k_im][0], b_matrix_intent=cb_int[k_im][0],
        b_left_x=cb_int[((1, 0), "plot")][0], b_left_y=cb_int[((1, 0), "plot")][1],
        b_top_x=cb_int[((0, 1), "plot")][0], b_top_y=cb_int[((0, 1), "plot")][1])
    print("a matrix", ca[k_im][0].shape, "b matrix", cb_int[k_im][0].shape, "literal == intent:",
          np.allclose(cb_lit[k_im][0], cb_int[k_im][0]))


if __name__ == "__main__":
    main()
