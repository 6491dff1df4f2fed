"""GPU parity of the per-particle Distributed-IB set transformer (BASELINE config 5) against the
float64 CPU oracle (oracle/set_transformer_oracle.py) and against the golden fixture produced by executing the reference
notebook's own model-building / train_step code (tests/golden/set_transformer_forward.npz).
Tolerances: activations 2e-4 (abs + rel), KL 1e-3 nats, gradients 3e-4 of each block's max-abs."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dib_oracle as orc  # noqa: E402
import set_transformer_oracle as sto  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "set_transformer_forward.npz")


def _model(spec: sto.SetTransformerSpec, seed=0, noise_seed=5, bias_scale=0.05):
    import dib_amd
    m = dib_amd.SetTransformerDIB(spec.particle_feature_dimensions, spec.number_positional_encoding_frequencies,
                                  spec.particle_encoder_arch_spec, spec.bottleneck_dimension, spec.key_dim,
                                  spec.number_heads_per_mha, spec.number_attention_blocks, spec.ff_arch_per_block,
                                  spec.final_processing_arch, spec.output_dimensionality, spec.logvar_initialization,
                                  spec.layer_norm_epsilon, init_seed=seed, noise_seed=noise_seed)
    p = m.get_params()
    rng = np.random.default_rng(seed + 100)
    for k in p:  # non-trivial biases / LayerNorm parameters (Keras initialises them to 0 / 1, which hides mistakes)
        if k.endswith("_b"):
            p[k] = (bias_scale * rng.standard_normal(p[k].shape)).astype(np.float32)
        if k.endswith("_g"):
            p[k] = (1.0 + bias_scale * rng.standard_normal(p[k].shape)).astype(np.float32)
    m.set_params(p)
    return m, {k: torch.tensor(v, dtype=torch.float64) for k, v in m.get_params().items()}


def _eps(seed, step, T, E):
    return orc.philox_normal_all(seed, step, np.arange(T, dtype=np.uint32), 1, E)[:, 0, :]


def test_parameter_layout_matches_keras_creation_order():
    spec = sto.SetTransformerSpec()
    m, _ = _model(spec)
    assert list(m.shapes) == list(sto.param_shapes(spec)) and m.n_params == 1299649
    assert all(tuple(m.shapes[k]) == tuple(v) for k, v in sto.param_shapes(spec).items())


def test_forward_replays_the_notebook_fixture():
    """The notebook's own model code executed on the NumPy stand-in (golden) -> mu, logvar (with the -3 offset), KL and,
    with the fixture's sampled embeddings injected, the six attention blocks + head."""
    g = np.load(GOLD)
    spec = sto.SetTransformerSpec()
    m, _ = _model(spec)
    flat, p, o = g["flat"], {}, 0
    for name, shp in sto.param_shapes(spec).items():
        n = int(np.prod(shp))
        p[name] = flat[o: o + n].reshape(shp)
        o += n
    m.set_params(p)
    pred = m.forward(g["feats"], embs_reparam=g["u"]).cpu().numpy()
    B, P = g["feats"].shape[:2]
    enc = m.particle_encoder(g["feats"]).cpu().numpy()
    assert np.abs(enc[..., :32] - g["mu"]).max() < 2e-4 * (1 + np.abs(g["mu"]).max())
    assert np.abs(enc[..., 32:] - 3.0 - g["logvar"]).max() < 2e-4 * (1 + np.abs(g["logvar"]).max())
    assert abs(float(m.last["kl"].item()) - float(g["kl"])) < 1e-3
    assert np.abs(pred - g["pred"]).max() < 2e-4 * (1 + np.abs(g["pred"]).max()), (pred.ravel()[:4], g["pred"].ravel()[:4])


SPECS = {
    "tiny": (sto.SetTransformerSpec(particle_encoder_arch_spec=[8], bottleneck_dimension=4, key_dim=3, number_heads_per_mha=2,
                                    number_attention_blocks=2, ff_arch_per_block=[5, 4], final_processing_arch=[6]), 3, 5),
    "odd_particles": (sto.SetTransformerSpec(number_attention_blocks=2), 5, 13),
    "reference_size": (sto.SetTransformerSpec(), 32, 50),      # the notebook: 32 neighbourhoods x 50 particles x 12 features
}


@pytest.mark.parametrize("name", list(SPECS))
def test_forward_backward_parity(name):
    spec, B, P = SPECS[name]
    m, p = _model(spec, seed=hash(name) % 97)
    rng = np.random.default_rng(B * 100 + P)
    feats = rng.standard_normal((B, P, spec.particle_feature_dimensions)).astype(np.float32)
    y = (rng.random((B, 1)) > 0.5).astype(np.float32)
    beta, step = 0.07, 11
    m.beta_dev.fill_(beta)
    pred = m.forward(feats, step=step).cpu().numpy()
    m.loss_and_backward(y)
    torch.cuda.synchronize()
    E = spec.bottleneck_dimension
    eps = _eps(5, step, B * P, E).reshape(B, P, E)
    vals, grads = sto.loss_and_grads(spec, p, feats.astype(np.float64), eps, y.astype(np.float64), beta)
    out = sto.forward(spec, p, feats.astype(np.float64), eps, y.astype(np.float64), beta)
    pl = m.last["plan"]
    u = m._view(pl, "x0", B, P, E).cpu().numpy()
    assert np.abs(u - out["u"].numpy()).max() < 2e-4 * (1 + np.abs(out["u"].numpy()).max()), "sampled embeddings"
    assert np.abs(pred - out["pred"].numpy()).max() < 2e-4 * (1 + np.abs(out["pred"].numpy()).max()), "logits"
    assert abs(float(m.last["kl"].item()) - vals["kl"]) < 1e-3 * max(1.0, vals["kl"] / 50), ("KL", float(m.last["kl"].item()), vals["kl"])
    assert abs(float(m.last["bce"].item()) - vals["bce"]) < 2e-4 * (1 + abs(vals["bce"])), "bce"
    got = m.get_grads()
    for k, r in grads.items():
        r = r.numpy()
        err = np.abs(got[k] - r).max()
        assert err <= 3e-4 * (np.abs(r).max() + 1e-6), (k, err, np.abs(r).max())


def test_train_steps_match_oracle_adam():
    """Three notebook train steps (lr warm-up value, per-step beta, Keras Adam) against the oracle."""
    spec = sto.SetTransformerSpec(number_attention_blocks=2)
    B, P = 8, 20
    m, p = _model(spec, seed=3)
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((B, P, 12)).astype(np.float32)
    y = (rng.random((B, 1)) > 0.5).astype(np.float32)
    mom = {k: torch.zeros_like(v) for k, v in p.items()}
    vel = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in range(3):
        lr = sto.learning_rate_schedule(step + 5, 1e-3, 100)
        beta = sto.beta_schedule(step, 2e-3, 2e-1, 10)
        m.lr_dev.fill_(lr)
        m.beta_dev.fill_(beta)
        bce = float(m.train_step(feats, y).item())
        eps = _eps(5, step, B * P, 32).reshape(B, P, 32)
        vals, grads = sto.loss_and_grads(spec, p, feats.astype(np.float64), eps, y.astype(np.float64), beta)
        assert abs(bce - vals["bce"]) < 2e-4 * (1 + abs(vals["bce"])), (step, bce, vals["bce"])
        t = step + 1
        lr_t = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        for k in p:
            mom[k] += 0.1 * (grads[k] - mom[k])
            vel[k] += 0.001 * (grads[k] ** 2 - vel[k])
            p[k] = p[k] - lr_t * mom[k] / (torch.sqrt(vel[k]) + 1e-7)
    got = m.get_params()
    for k in p:
        assert np.abs(got[k] - p[k].numpy()).max() < 2e-4, k
    assert int(m.t_dev.item()) == 3


def test_large_token_count_uses_split_weight_gradients():
    """T = batch * particles >= 2048 tokens: weight gradients are accumulated in batch slabs + fixed-order reduce."""
    spec = sto.SetTransformerSpec(number_attention_blocks=1)
    B, P = 8, 300
    m, p = _model(spec, seed=4)
    rng = np.random.default_rng(2)
    feats = rng.standard_normal((B, P, 12)).astype(np.float32)
    y = (rng.random((B, 1)) > 0.5).astype(np.float32)
    m.beta_dev.fill_(0.01)
    m.forward(feats, step=0)
    assert m.last["plan"]["nsplit"] > 1
    m.loss_and_backward(y)
    g1 = m.grads.clone()
    m.forward(feats, step=0)
    m.loss_and_backward(y)
    assert torch.equal(g1, m.grads), "deterministic replay"
    eps = _eps(5, 0, B * P, 32).reshape(B, P, 32)
    vals, grads = sto.loss_and_grads(spec, p, feats.astype(np.float64), eps, y.astype(np.float64), 0.01)
    got = m.get_grads()
    for k, r in grads.items():
        r = r.numpy()
        assert np.abs(got[k] - r).max() <= 3e-4 * (np.abs(r).max() + 1e-6), k
