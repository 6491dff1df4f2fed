"""Diagnostic: per-gradient-block error of the set-transformer step against the float64 oracle for both attention
implementations (8 neighbourhoods x 300 particles, 1 block, 12 heads) - relative to each block's own max and to the overall max."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import test_gpu_set_transformer as t  # noqa: E402

sto = t.sto
spec = sto.SetTransformerSpec(number_attention_blocks=1)
B, P = 8, 300
rng = np.random.default_rng(2)
feats = rng.standard_normal((B, P, 12)).astype(np.float32)
y = (rng.random((B, 1)) > 0.5).astype(np.float32)
eps = t._eps(5, 0, B * P, 32).reshape(B, P, 32)
ref = None
for att in ("gemm", "flash"):
    m, p = t._model(spec, seed=4, attention=att)
    if ref is None:
        vals, ref = sto.loss_and_grads(spec, p, feats.astype(np.float64), eps, y.astype(np.float64), 0.01)
        gmax = max(float(r.abs().max()) for r in ref.values())
    m.beta_dev.fill_(0.01)
    m.forward(feats, step=0)
    m.loss_and_backward(y)
    got = m.get_grads()
    print(f"== attention={att}  (overall gradient max {gmax:.3g})")
    for k, r in ref.items():
        r = r.numpy()
        err = np.abs(got[k] - r).max()
        print(f"  {k:14s} max|g| {np.abs(r).max():9.3g}  err {err:9.3g}  err/blockmax {err / max(np.abs(r).max(), 1e-30):8.2e}  err/gmax {err / gmax:8.2e}")
