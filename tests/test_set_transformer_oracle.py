"""SURVEY 8(f) rank 3 (per-particle set-transformer DIB, BASELINE config 5): the CPU oracle
(oracle/set_transformer_oracle.py) against fixtures produced by executing the reference notebook's own model-building,
train_step and probe-grid code on the NumPy stand-in for TensorFlow (tests/golden/make_golden_set_transformer.py,
make_golden_probe_grid.py).  These tests pin the checker that tests/test_gpu_set_transformer.py holds the HIP path to."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import set_transformer_oracle as sto  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "set_transformer_forward.npz")


def _params_from_flat(spec, flat):
    p, off = {}, 0
    for name, shp in sto.param_shapes(spec).items():
        n = int(np.prod(shp))
        p[name] = torch.tensor(flat[off: off + n].reshape(shp), dtype=torch.float64)
        off += n
    assert off == flat.size
    return p


def test_parameter_count_matches_the_notebook_architecture():
    spec = sto.SetTransformerSpec()
    n = sum(int(np.prod(s)) for s in sto.param_shapes(spec).values())
    enc = 60 * 128 + 128 + 128 * 128 + 128 + 128 * 64 + 64
    blk = 3 * (32 * 12 * 128 + 12 * 128) + (12 * 128 * 32 + 32) + 2 * 64 + (32 * 128 + 128) + (128 * 32 + 32)
    head = 32 * 256 + 256 + 256 + 1
    assert n == enc + 6 * blk + head == 1299649
    assert sto.flops_per_neighbourhood(spec, 50) > 0


def test_forward_matches_the_notebook_code_executed_on_the_numpy_backend():
    g = np.load(GOLD)
    spec = sto.SetTransformerSpec()
    p = _params_from_flat(spec, g["flat"])
    out = sto.forward(spec, p, g["feats"], g["eps"], g["is_loci"], float(g["beta"]))
    assert np.abs(out["mu"].numpy() - g["mu"]).max() < 1e-12
    assert np.abs(out["logvar"].numpy() - g["logvar"]).max() < 1e-12        # includes the -3 offset
    assert np.abs(out["u"].numpy() - g["u"]).max() < 1e-12
    assert abs(float(out["kl"]) - float(g["kl"])) < 1e-10 * abs(float(g["kl"]))
    assert np.abs(out["pred"].numpy() - g["pred"]).max() < 1e-11            # 6 attention blocks, two implementations
    assert abs(float(out["bce"]) - float(g["bce"])) < 1e-12
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-10


def test_feature_preprocessing_matches_the_notebook_function():
    g = np.load(GOLD)
    got = sto.convert_to_per_particle_feature_set(g["raw_pos"], g["raw_types"], number_particles_to_use=6)
    assert got.shape == (6, 12) and np.abs(got - g["ref_feats"]).max() < 1e-6
    radii = got[:, 4]
    assert np.all(np.diff(radii) >= 0)                                     # nearest particles first


def test_autograd_gradients_agree_with_central_differences():
    spec = sto.SetTransformerSpec(particle_encoder_arch_spec=[8], bottleneck_dimension=4, key_dim=3, number_heads_per_mha=2,
                                  number_attention_blocks=2, ff_arch_per_block=[5, 4], final_processing_arch=[6])
    p = sto.init_params(spec, 1)
    rng = np.random.default_rng(2)
    for k in p:
        if k.endswith("_b"):
            p[k] = p[k] + torch.tensor(0.05 * rng.standard_normal(tuple(p[k].shape)))
    B, P = 3, 5
    feats = rng.standard_normal((B, P, 12))
    eps = rng.standard_normal((B, P, 4))
    y = (rng.random((B, 1)) > 0.5).astype(np.float64)
    vals, grads = sto.loss_and_grads(spec, p, feats, eps, y, beta=0.3)
    assert np.isfinite(vals["loss"])
    h = 1e-6
    for name in ("enc0_w", "blk0_q_w", "blk0_o_b", "blk1_ln1_g", "blk1_ff1_w", "fin0_b", "out_w"):
        idx = tuple(rng.integers(0, s) for s in p[name].shape)
        q = {k: v.clone() for k, v in p.items()}
        q[name][idx] += h
        up = float(sto.forward(spec, q, feats, eps, y, 0.3)["loss"])
        q[name][idx] -= 2 * h
        dn = float(sto.forward(spec, q, feats, eps, y, 0.3)["loss"])
        num = (up - dn) / (2 * h)
        assert abs(num - float(grads[name][idx])) < 1e-6 * (1 + abs(num)), name


def test_permutation_invariance_of_the_set_transformer():
    """Mean pooling over self-attention blocks: the prediction does not depend on the particle order."""
    spec = sto.SetTransformerSpec(number_attention_blocks=2)
    p = sto.init_params(spec, 3)
    rng = np.random.default_rng(4)
    u = torch.tensor(rng.standard_normal((2, 9, 32)))
    perm = torch.tensor(rng.permutation(9))
    a = sto.set_transformer(spec, p, u)
    b = sto.set_transformer(spec, p, u[:, perm])
    assert torch.allclose(a, b, rtol=0, atol=1e-12)


def test_schedules():
    assert sto.learning_rate_schedule(0, 1e-4, 25000) == 0.0
    assert sto.learning_rate_schedule(1250, 1e-4, 25000) == pytest.approx(5e-5)
    assert sto.learning_rate_schedule(2500, 1e-4, 25000) == 1e-4 == sto.learning_rate_schedule(20000, 1e-4, 25000)
    assert sto.beta_schedule(0, 2e-6, 2e-1, 25000) == pytest.approx(2e-6)
    assert sto.beta_schedule(12500, 2e-6, 2e-1, 25000) == pytest.approx(np.sqrt(2e-6 * 2e-1))


def test_probe_grid_bounds_match_the_notebook_statements_executed():
    """oracle.probe_info_bounds against the notebook's own inner-loop statements run on the NumPy TF stand-in
    (tests/golden/make_golden_probe_grid.py)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "probe_grid_bounds.npz"))
    u = g["mus_probes"] + np.exp(g["logvars_probes"] / 2.0) * g["eps"]
    assert np.abs(u - g["sampled_u_probes"]).max() < 1e-14
    lo, up = sto.probe_info_bounds(g["mus_probes"], g["logvars_probes"], u, g["mus_data"], g["logvars_data"])
    assert np.abs(lo - g["infonce_per"]).max() < 1e-11 and np.abs(up - g["loo_per"]).max() < 1e-10
    assert (lo <= np.log(41) + 1e-12).all() and (lo <= up + 1e-12).all()
