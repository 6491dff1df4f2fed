"""Fixtures from the reference's OWN models.py source, executed on the NumPy stand-in for TensorFlow
(tests/golden/tf_numpy_shim.py - read its header for what that does and does not pin).  Run here only:
    python tests/golden/make_golden_models.py
Writes tests/golden/models_forward.npz: for several architectures the inputs, parameters and noise, and the outputs of
reference models.py:96-123 (prediction, per-feature KL metrics, beta * sum KL loss term), of PositionalEncoding.call
(models.py:22-23) and of InfoBottleneckAnnealingCallback.on_epoch_begin (models.py:147-149)."""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tf_numpy_shim as tf  # noqa: E402
import dib_oracle as orc  # noqa: E402

sys.modules["tensorflow"] = tf
sys.modules["utils"] = types.ModuleType("utils")        # models.py imports utils only for InfoPerFeatureCallback
spec = importlib.util.spec_from_file_location("ref_models", "/root/reference/models.py")
ref_models = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_models)

CASES = [
    dict(feature_dimensionalities=[1, 1, 1], feature_encoder_architecture=[8, 8], integration_network_architecture=[8],
         output_dimensionality=1, feature_embedding_dimension=4),
    dict(feature_dimensionalities=[2, 1, 3], feature_encoder_architecture=[6], integration_network_architecture=[5, 7],
         output_dimensionality=3, use_positional_encoding=False, activation_fn="tanh", feature_embedding_dimension=3),
    dict(feature_dimensionalities=[1, 4], feature_encoder_architecture=[5, 4, 3], integration_network_architecture=[],
         output_dimensionality=2, number_positional_encoding_frequencies=3, activation_fn="leaky_relu",
         feature_embedding_dimension=2, output_activation_fn="sigmoid"),
]


def main():
    out = {}
    for ci, kw in enumerate(CASES):
        model = ref_models.DistributedIBNet(**kw)                       # reference models.py:56-86
        ospec = orc.DIBSpec(**kw)
        p = orc.glorot_uniform_init(ospec, seed=100 + ci, dtype=np.float64)
        rng = np.random.default_rng(200 + ci)
        for bs in p.enc_b:                                              # non-zero biases: exercise the bias path
            for v in bs:
                v[...] = 0.1 * rng.standard_normal(v.shape)
        for v in p.int_b:
            v[...] = 0.1 * rng.standard_normal(v.shape)
        for f, seq in enumerate(model.feature_encoders):
            dense = [l for l in seq.layers if isinstance(l, tf.Dense)]
            for l, layer in enumerate(dense):
                layer.kernel, layer.bias = p.enc_W[f][l], p.enc_b[f][l]
        for l, layer in enumerate(model.integration_network.layers):
            layer.kernel, layer.bias = p.int_W[l], p.int_b[l]
        B, F, E = 11, len(kw["feature_dimensionalities"]), kw["feature_embedding_dimension"]
        x = rng.standard_normal((B, sum(kw["feature_dimensionalities"])))
        eps = rng.standard_normal((B, F, E))
        tf.push_eps([eps[:, f, :] for f in range(F)])                    # consumed in feature order (models.py:105-108)
        beta = 0.37
        model.beta.assign(beta)
        pred = model(x)                                                  # reference models.py:96-123
        out[f"c{ci}_x"], out[f"c{ci}_eps"], out[f"c{ci}_beta"] = x, eps, np.float64(np.float32(beta))
        out[f"c{ci}_pred"] = pred
        out[f"c{ci}_kl"] = np.array([model.metrics_log[f"KL{f}"] for f in range(F)])
        out[f"c{ci}_kl_loss"] = np.float64(model.losses[0])
        out[f"c{ci}_beta_metric"] = np.float64(model.metrics_log["beta"])
        out[f"c{ci}_flat"] = np.concatenate([t.ravel() for t in p.tensors()])
    # PositionalEncoding.call, models.py:22-23, with the frequency list DistributedIBNet builds (models.py:70)
    xs = np.random.default_rng(5).standard_normal((7, 2))
    out["posenc_x"] = xs
    out["posenc_out"] = ref_models.PositionalEncoding(2 ** np.arange(1, 5))(xs)
    # InfoBottleneckAnnealingCallback.on_epoch_begin, models.py:147-149 (float32 like TF)
    cb = ref_models.InfoBottleneckAnnealingCallback(1e-4, 3.0, 5, 20)
    cb.model = types.SimpleNamespace(beta=tf.Variable(1.0, dtype=tf.float32, trainable=False))
    betas = []
    for epoch in range(30):
        cb.on_epoch_begin(epoch)
        betas.append(cb.model.beta.value())
    out["anneal_args"] = np.array([1e-4, 3.0, 5, 20])
    out["anneal_betas"] = np.array(betas, dtype=np.float32)
    # ---- reference utils.py executed from its own source on the same stand-in ----
    uspec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/utils.py")
    ref_utils = importlib.util.module_from_spec(uspec)
    uspec.loader.exec_module(ref_utils)
    rng = np.random.default_rng(9)
    e1, e2 = rng.standard_normal((6, 5)), rng.standard_normal((4, 5))
    out["sim_e1"], out["sim_e2"] = e1, e2
    for kind in ("l2sq", "l2", "l1", "linf", "cosine"):                      # utils.py:131-175
        out[f"sim_{kind}"] = ref_utils.get_scaled_similarity(e1, e2, kind, 0.7)
    # utils.py:10-73: two evaluation batches of 16 points, 3-dimensional embeddings, noise fixed by the fixture
    nb, bs, ed = 2, 16, 3
    mus = rng.standard_normal((nb, bs, ed))
    logvars = 0.8 * rng.standard_normal((nb, bs, ed)) - 1.0
    eps_mi = rng.standard_normal((nb, bs, ed))
    tf.push_eps([eps_mi[i] for i in range(nb)])
    encoder = lambda batch: batch                                            # the "dataset" already holds (mu | logvar)
    data = tf.FixedBatches([np.concatenate([mus[i], logvars[i]], -1).astype(np.float32) for i in range(nb)])
    out["mi_mus"], out["mi_logvars"], out["mi_eps"] = mus.astype(np.float32), logvars.astype(np.float32), eps_mi
    out["mi_bounds"] = ref_utils.estimate_mi_sandwich_bounds(encoder, data, evaluation_batch_size=bs,
                                                             number_evaluation_batches=nb)
    np.savez_compressed(os.path.join(HERE, "models_forward.npz"), **out)
    print("wrote", len(out), "arrays; case 0 KL", out["c0_kl"], "pred[0]", out["c0_pred"][0])


if __name__ == "__main__":
    main()
