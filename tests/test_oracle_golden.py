"""CPU: pin the oracle against every golden vector the reference offers for this path
(tests/golden/*.npz were produced by EXECUTING reference code, see tests/golden/make_golden.py) and
against the numbers printed in the reference's notebooks."""
import os

import numpy as np
import pytest

import dib_oracle as orc
from _helpers import SPECS, random_params

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_boolean_circuit_truth_table_matches_reference_data_py():
    g = np.load(os.path.join(GOLD, "boolean_circuit.npz"))
    x, y = orc.boolean_circuit_truth_table()
    assert np.array_equal(x, g["x_train"]) and np.array_equal(y, g["y_train"])
    p1 = y.mean()
    hy = orc.entropy_bits([p1, 1 - p1])
    assert abs(hy - float(g["entropy_y_bits"])) < 1e-12
    assert abs(hy - 0.758) < 5e-4  # Boolean_circuits.ipynb:238
    assert bool(g["loss_from_logits"]) and bool(g["loss_is_info_based"])
    import dib_amd
    d = dib_amd.data.fetch_boolean_circuit()
    assert np.array_equal(d["x_train"], g["x_train"]) and np.array_equal(d["y_train"], g["y_train"])
    assert d["loss"].kind == "bce_logits" and d["feature_dimensionalities"] == [1] * 10


def test_distance_matrices_match_reference_utils_py():
    g = np.load(os.path.join(GOLD, "utils_distance_mats.npz"))
    import dib_amd
    for mod in (orc, dib_amd.utils):
        assert np.allclose(mod.bhattacharyya_dist_mat(g["mus1"], g["lvs1"], g["mus2"], g["lvs2"]), g["bhattacharyya"],
                           rtol=1e-10, atol=1e-12)
        assert np.allclose(mod.bhattacharyya_dist_mat(g["mus1"], g["lvs1"], g["mus1"], g["lvs1"]),
                           g["bhattacharyya_self"], rtol=1e-10, atol=1e-12)
        assert np.allclose(mod.kl_divergence_mat(g["mus1"], g["lvs1"], g["mus2"], g["lvs2"]), g["kl"], rtol=1e-10)
    assert abs(dib_amd.utils.compute_entropy_bits(g["probs"]) - float(g["entropy_bits"])) < 1e-12
    assert abs(orc.entropy_bits(g["probs"]) - float(g["entropy_bits"])) < 1e-12


@pytest.mark.parametrize("circuit,hy", [
    ([0, 1, 2, 3, [0, 2, 0], [2, 4, 3], [0, 5, 1]], 0.811),   # SI circuit (c), Boolean_circuits.ipynb:987-992
    ([0, 1, 2, 3, [1, 1, 3], [0, 4, 0], [2, 2, 5]], 1.000),   # SI circuit (d), :1058-1063
])
def test_si_circuit_entropies_from_notebook(circuit, hy):
    x, y = orc.boolean_circuit_truth_table(circuit, 4)
    p1 = y.mean()
    assert abs(orc.entropy_bits([p1, 1 - p1]) - hy) < 1e-3
    full = orc.subset_mutual_information_bits(x, y, [0, 1, 2, 3])
    assert abs(full - orc.entropy_bits([p1, 1 - p1])) < 1e-12  # deterministic circuit: I(X;Y) = H(Y)
    for s in ([0], [1], [0, 1], [2, 3]):
        assert -1e-12 <= orc.subset_mutual_information_bits(x, y, s) <= full + 1e-12


def test_positional_encoding_layout_is_blockwise():
    x = np.array([[0.1, 0.2]])
    out = orc.positional_encoding(x, [2, 4])
    assert np.allclose(out, [[0.1, 0.2, np.sin(0.2), np.sin(0.4), np.sin(0.4), np.sin(0.8)]])
    spec = orc.DIBSpec([1], [4], [4], 1)
    assert list(spec.frequencies) == [2, 4, 8, 16] and spec.encoder_input_dim(0) == 5  # models.py:70


def test_kl_known_answers():
    spec = orc.DIBSpec([1, 1], [], [], 1, use_positional_encoding=False, feature_embedding_dimension=4)
    p = orc.glorot_uniform_init(spec)
    for f in range(2):
        p.enc_W[f][0][:] = 0
    c = 0.7
    p.enc_b[0][0][:4] = c  # mu = c, logvar = 0  => KL = E*c^2/2
    x = np.zeros((5, 2))
    out = orc.forward(spec, p, x, np.zeros((5, 2, 4)))
    assert np.allclose(out.kl, [4 * c * c / 2, 0.0])


def test_beta_schedule_endpoints_and_geometric_ramp():
    assert orc.beta_schedule(0, 1e-4, 3.0, 10, 100) == np.float32(1e-4) or abs(orc.beta_schedule(0, 1e-4, 3.0, 10, 100) - 1e-4) < 1e-10
    assert abs(orc.beta_schedule(10, 1e-4, 3.0, 10, 100) - 1e-4) < 1e-10
    assert abs(orc.beta_schedule(110, 1e-4, 3.0, 10, 100) - 3.0) < 1e-5
    r1 = orc.beta_schedule(61, 1e-4, 3.0, 10, 100) / orc.beta_schedule(60, 1e-4, 3.0, 10, 100)
    r2 = orc.beta_schedule(31, 1e-4, 3.0, 10, 100) / orc.beta_schedule(30, 1e-4, 3.0, 10, 100)
    assert abs(r1 - r2) < 1e-4
    import dib_amd
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-4, 3.0, 10, 100)
    for e in (0, 5, 10, 11, 60, 110):
        assert cb.beta_at(e) == orc.beta_schedule(e, 1e-4, 3.0, 10, 100)


@pytest.mark.parametrize("name", ["odd_shapes_tanh", "no_posenc_leaky", "sigmoid_out_elu", "boolean4_32x32"])
def test_oracle_backward_gradcheck(name):
    spec = SPECS[name]
    p = random_params(spec, 1)
    rng = np.random.default_rng(0)
    B = 6
    x = rng.standard_normal((B, sum(spec.feature_dimensionalities)))
    if spec.output_dimensionality == 1:
        kind = "bce" if spec.output_activation_fn == "sigmoid" else "bce_logits"
        y = rng.integers(0, 2, (B, 1))
    else:
        kind = "sparse_cce_logits"
        y = rng.integers(0, spec.output_dimensionality, (B, 1))
    eps = orc.philox_normal_all(5, 3, np.arange(B), spec.number_features, spec.feature_embedding_dimension)
    beta = 0.3

    def L():
        c = orc.forward(spec, p, x, eps)
        return orc.loss_and_grad(kind, y, c.pred)[0] + beta * c.kl.sum()

    c = orc.forward(spec, p, x, eps)
    _, g, _ = orc.backward(spec, p, x, y, c, beta, kind)
    for t, gt in zip(p.tensors(), g.tensors()):
        for _ in range(4):
            idx = tuple(rng.integers(0, s) for s in t.shape)
            old, h = t[idx], 1e-6
            t[idx] = old + h; lp = L()
            t[idx] = old - h; lm = L()
            t[idx] = old
            assert abs((lp - lm) / (2 * h) - gt[idx]) < 1e-6


def test_param_count_and_flops_match_survey():
    s3 = orc.DIBSpec([1] * 64, [128, 128], [256, 256], 1)
    assert s3.num_params() == 2224897 and s3.gemm_flops_per_sample() == 13141504
    s1 = orc.DIBSpec([1] * 4, [32, 32], [256, 256], 1)
    assert s1.num_params() == 112513 and s1.gemm_flops_per_sample() == 667648


def test_philox_known_answer_and_moments():
    # Random123 known-answer test vector for philox4x32-10: counter=key=0
    out = orc.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(v) for v in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    out = orc.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(v) for v in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    e = orc.philox_normal_all(1, 0, np.arange(20000), 2, 32)
    assert abs(e.mean()) < 5e-3 and abs(e.std() - 1) < 5e-3 and abs((e ** 4).mean() - 3) < 0.05


def test_mi_sandwich_bounds_analytic_limits():
    """SURVEY section 4 pins for utils.estimate_mi_sandwich_bounds: X ~ U({+-1}^k) through a Gaussian channel:
    zero separation -> 0 nats; large separation -> k bits (as long as log N allows); lower <= upper."""
    rng = np.random.default_rng(0)
    n, k = 512, 2
    bits = rng.integers(0, 2, (n, k)) * 2.0 - 1.0
    lv = np.zeros((n, k))
    for sep, want in ((0.0, 0.0), (6.0, k * np.log(2))):
        mus = sep * bits
        u = orc.mi_sandwich_sample_u(mus, lv, 1, 0, 0)
        lo, up = orc.mi_sandwich_bounds_batch(mus, lv, u)
        assert lo <= up + 1e-9
        assert abs(lo - want) < 0.08 and abs(up - want) < 0.08, (sep, lo, up, want)


@pytest.mark.parametrize("kind", ["l2sq", "l2", "l1", "linf", "cosine"])
def test_infonce_oracle_similarity_properties(kind):
    """utils.get_scaled_similarity semantics (utils.py:131-175): self-similarity is maximal, temperature divides,
    and the symmetric InfoNCE of perfectly matched, well separated embeddings tends to 0."""
    rng = np.random.default_rng(0)
    a = rng.standard_normal((6, 5))
    S = orc.scaled_similarity(a, a, kind, 2.0)
    assert np.allclose(np.diag(S), S.max(axis=1), atol=1e-4)
    assert np.allclose(orc.scaled_similarity(a, a, kind, 1.0), 2.0 * S)
    if kind != "cosine":
        assert orc.infonce_loss(20 * a, 20 * a, kind, 1.0) < 1e-3
    b = rng.standard_normal((6, 5))
    l0 = orc.infonce_loss(a, b, kind, 1.0)
    assert l0 > 0 and abs(l0 - orc.infonce_loss(b, a, kind, 1.0)) < 1e-12  # symmetric in (X, Y)


@pytest.mark.parametrize("kind", ["l2sq", "l2", "l1", "linf", "cosine"])
def test_infonce_autograd_oracle_equals_numpy_restatement(kind):
    """The float64 autograd checker used at working batch sizes (oracle/dib_torch_cpu.infonce_loss_and_grads) against the
    numpy restatement of utils.py:131-175 / train.py:203-215 and its central-difference gradients."""
    import dib_torch_cpu as tc
    rng = np.random.default_rng(2)
    a = rng.standard_normal((9, 6))
    b = a + 0.7 * rng.standard_normal((9, 6))
    loss, ga, gb = tc.infonce_loss_and_grads(a, b, kind, 0.7)
    assert abs(loss - orc.infonce_loss(a, b, kind, 0.7)) < 1e-12
    n1, n2 = orc.infonce_grads_numeric(a, b, kind, 0.7)
    assert np.abs(ga - n1).max() < 1e-7 and np.abs(gb - n2).max() < 1e-7


# ---- fixtures produced by executing the reference's own models.py source on a NumPy stand-in for TensorFlow
#      (tests/golden/make_golden_models.py + tf_numpy_shim.py): pins the graph the reference builds ----
_MODEL_CASES = [
    dict(feature_dimensionalities=[1, 1, 1], feature_encoder_architecture=[8, 8], integration_network_architecture=[8],
         output_dimensionality=1, feature_embedding_dimension=4),
    dict(feature_dimensionalities=[2, 1, 3], feature_encoder_architecture=[6], integration_network_architecture=[5, 7],
         output_dimensionality=3, use_positional_encoding=False, activation_fn="tanh", feature_embedding_dimension=3),
    dict(feature_dimensionalities=[1, 4], feature_encoder_architecture=[5, 4, 3], integration_network_architecture=[],
         output_dimensionality=2, number_positional_encoding_frequencies=3, activation_fn="leaky_relu",
         feature_embedding_dimension=2, output_activation_fn="sigmoid"),
]


def _params_from_flat(spec, flat):
    p = orc.glorot_uniform_init(spec, 0, dtype=np.float64)
    off = 0
    for t in p.tensors():
        t[...] = flat[off: off + t.size].reshape(t.shape)
        off += t.size
    assert off == flat.size
    return p


@pytest.mark.parametrize("ci", range(len(_MODEL_CASES)))
def test_forward_matches_reference_models_py_executed_on_numpy_backend(ci):
    """oracle.forward vs reference models.py:96-123 (DistributedIBNet.call) run from its own source."""
    g = np.load(os.path.join(GOLD, "models_forward.npz"))
    spec = orc.DIBSpec(**_MODEL_CASES[ci])
    p = _params_from_flat(spec, g[f"c{ci}_flat"])
    c = orc.forward(spec, p, g[f"c{ci}_x"], g[f"c{ci}_eps"])
    assert np.abs(c.pred - g[f"c{ci}_pred"]).max() < 1e-12
    assert np.abs(c.kl - g[f"c{ci}_kl"]).max() < 1e-12                      # add_metric(KL{f}), models.py:115
    beta = float(g[f"c{ci}_beta"])
    assert abs(beta * c.kl.sum() - float(g[f"c{ci}_kl_loss"])) < 1e-12      # add_loss(beta * sum KL), models.py:118
    assert float(g[f"c{ci}_beta_metric"]) == beta                           # add_metric(beta), models.py:121


def test_positional_encoding_and_beta_ramp_match_reference_source():
    g = np.load(os.path.join(GOLD, "models_forward.npz"))
    got = orc.positional_encoding(g["posenc_x"], [2, 4, 8, 16])               # models.py:22-23 with models.py:70
    assert np.abs(got - g["posenc_out"]).max() < 1e-15
    b0, b1, n_pre, n_ann = g["anneal_args"]
    ours = np.array([orc.beta_schedule(e, b0, b1, int(n_pre), int(n_ann)) for e in range(30)], dtype=np.float32)
    # both sides evaluate the ramp in float32 (models.py:147-149); allow one float32 ulp for exp/log rounding order
    assert np.all(np.abs(ours - g["anneal_betas"]) <= 2e-7 * np.abs(g["anneal_betas"]))
    assert ours[0] == np.float32(1e-4) or abs(ours[0] - 1e-4) < 1e-10


def test_similarities_and_mi_bounds_match_reference_utils_py_executed_on_numpy_backend():
    """oracle.scaled_similarity vs utils.py:131-175 and oracle.mi_sandwich_bounds_batch vs utils.py:36-73, both run
    from the reference's own source (tests/golden/make_golden_models.py)."""
    g = np.load(os.path.join(GOLD, "models_forward.npz"))
    for kind in ("l2sq", "l2", "l1", "linf", "cosine"):
        got = orc.scaled_similarity(g["sim_e1"], g["sim_e2"], kind, 0.7)
        assert np.abs(got - g[f"sim_{kind}"]).max() < 1e-12, kind
    bounds = []
    for i in range(g["mi_mus"].shape[0]):
        mus, lvs = g["mi_mus"][i].astype(np.float64), g["mi_logvars"][i].astype(np.float64)   # utils.py:40-41 casts
        u = mus + np.exp(lvs / 2.0) * g["mi_eps"][i]                                           # utils.py:44-45
        bounds.append(orc.mi_sandwich_bounds_batch(mus, lvs, u))
    got = np.mean(np.array(bounds), 0)                                                         # utils.py:73
    assert np.abs(got - g["mi_bounds"]).max() < 1e-10, (got, g["mi_bounds"])


# ---- custom-loop accounting (train.py:222-285) pinned on the reference's own statements executed -------------------------
def _loop_stub_run(loop_fn, case):
    """Drive a `run_loop`-shaped function with the stubs the golden generator gave the reference's loop."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("make_golden_infonce_loop", os.path.join(GOLD, "make_golden_infonce_loop.py"))
    gen = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(gen)
    n, nv, bs, n_pre, n_ann, b0, b1, width = gen.CASES[case]
    state = {"beta": np.float32(1.0), "call": 0, "log": []}

    def step(training):
        l, k = gen.stub_values(state["call"], training, state["beta"], width)
        state["log"].append((state["call"], int(training), float(state["beta"])))
        state["call"] += 1
        return l, k

    out = loop_fn(dataset_length=n, validation_set_length=nv, batch_size=bs, number_pretraining_epochs=n_pre,
                  number_annealing_epochs=n_ann, beta_start=b0, beta_end=b1, train_step=lambda s: step(True),
                  validation_step=lambda e, vb: step(False),
                  assign_beta=lambda v: state.__setitem__("beta", np.float32(v)))
    return out, np.array(state["log"], dtype=np.float64)


@pytest.mark.parametrize("case", ["fractional", "bankers_half", "repeated_boundaries", "exact"])
@pytest.mark.parametrize("which", ["oracle", "product"])
def test_custom_loop_accounting_matches_reference_statements(case, which):
    """oracle run_loop AND the product's host loop (dib_amd.infonce.run_custom_loop) reproduce, from the same step stubs, the
    series that train.py:224-285 itself produced: beta (float32), epoch-mean InfoNCE losses (float32), KL means, the order
    and beta of every train / validation call (incl. the first step at beta = 1), E - 1 recorded epochs."""
    g = np.load(os.path.join(GOLD, "infonce_loop.npz"))
    if which == "oracle":
        import infonce_loop_oracle as ilo
        fn = ilo.run_loop
    else:
        from dib_amd.infonce import run_custom_loop as fn
    out, log = _loop_stub_run(fn, case)
    assert np.array_equal(log, g[f"{case}_calls"])
    assert np.array_equal(np.float32(out["beta"]), g[f"{case}_beta"]) and np.float32(out["beta"]).dtype == g[f"{case}_beta"].dtype
    assert np.array_equal(np.float32(out["loss_infonce"]), g[f"{case}_loss"])
    assert np.array_equal(np.float32(out["loss_infonce_validation"]), g[f"{case}_loss_validation"])
    assert np.allclose(np.asarray(out["kl"], dtype=np.float64) / np.log(2), g[f"{case}_kl_bits"], rtol=1e-14, atol=0)
    assert np.allclose(np.asarray(out["kl_validation"], dtype=np.float64) / np.log(2), g[f"{case}_kl_bits_validation"], rtol=1e-14, atol=0)


def test_infonce_loop_oracle_runs_the_composed_step():
    """float64 loop oracle (train.py:196-279) end to end on a toy problem: one Adam over X model + Y encoder (both networks move),
    first step at beta = 1, loss falls, kl_total is the row sum (the reference's own series)."""
    import torch
    import infonce_loop_oracle as ilo
    spec = orc.DIBSpec([2, 1], [16], [16], 8, feature_embedding_dimension=4)
    p = orc.glorot_uniform_init(spec, 1)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((96, 3)).astype(np.float32)
    y = np.stack([np.sin(x[:, 0]) + x[:, 2], x[:, 1] * x[:, 0]], -1).astype(np.float32)
    dims = [2 * 3, 16, 8]
    Ws = [rng.uniform(-1, 1, (i, o)) * np.sqrt(6 / (i + o)) for i, o in zip(dims[:-1], dims[1:])]
    ye = ilo.YEncoder(Ws, [np.zeros(o) for o in dims[1:]], "relu", True, 3)
    w0 = [t.detach().clone() for t in ye.tensors()]
    o = ilo.InfoNCELoopOracle(spec, p, ye, "l2", 1.0, 3e-3, noise_seed=2)
    betas = []
    orig = o.eval_batch
    o.eval_batch = lambda *a, **k: (betas.append(float(o.beta)), orig(*a, **k))[1]
    out = o.fit(x, y, x[:40], y[:40], batch_size=32, number_pretraining_epochs=4, number_annealing_epochs=6, beta_start=1e-3,
                beta_end=1.0, seed=0)
    assert betas[0] == 1.0 and abs(betas[1] - 1e-3) < 1e-9           # models.py:86, then train.py:248 after step 0
    assert o.t == 27 and out["kl"].shape == (9, 2) and out["beta"].dtype == np.float32
    assert np.allclose(out["kl_total"], out["kl"].sum(-1))
    assert out["loss_infonce"][-1] < out["loss_infonce"][0]
    assert all(float((a - b.detach()).abs().max()) > 0 for a, b in zip(w0, ye.tensors()))
    assert all(float(v.abs().max()) > 0 for v in o.m)


def test_subset_informations_match_the_notebook_statements_executed():
    """Boolean_circuits.ipynb:425-434 - the exhaustive I(X_S;Y) over all 2^10 subsets of the paper circuit's input gates, the
    notebook's own compute_info / meshgrid statements executed (tests/golden/make_golden_subset_mi.py) - against the oracle's
    restatement on the product's truth table (dib_amd.data.fetch_boolean_circuit) and against the table the GPU KAT's ceiling
    uses (tools/paper_circuit_run.subset_information_bits)."""
    import importlib.util
    import dib_amd
    fx = np.load(os.path.join(GOLD, "subset_mi.npz"))
    d = dib_amd.data.fetch_boolean_circuit()
    x, y = np.asarray(d["x_train"]), np.asarray(d["y_train"]).reshape(-1).astype(np.int64)
    tt = fx["truth_table"]
    assert np.array_equal((x > 0).astype(np.int8), tt[:, :10]) and np.array_equal(y, tt[:, 10])
    assert abs(float(fx["entropy_y_bits"]) - 0.7578784625) < 1e-9
    combos, want = fx["all_on_off_combos"], fx["all_mis_bits"]
    spec_ = importlib.util.spec_from_file_location("paper_circuit_run", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "paper_circuit_run.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    table = mod.subset_information_bits(x, y)
    for k in range(0, 1024, 7):
        s = [int(i) for i in np.where(combos[k])[0]]
        assert abs(orc.subset_mutual_information_bits(x, y, s) - want[k]) < 1e-12, s
    for c, w in zip(combos, want):
        assert abs(table[int((c.astype(np.int64) << np.arange(10)).sum())] - w) < 1e-12
    # monotone in the subset (more gates never carry less) and the full set carries all of H(Y)
    assert abs(want[-1] - float(fx["entropy_y_bits"])) < 1e-12 and want[0] == 0


# The numbers the reference's own run PRINTED for the six SI circuits (InfoDecomp_Boolean_circuits.ipynb, outputs of cell 10:
# "H(Y)=..., Sum of Shapley values = ...", "Shapley values: [...]", "Logistic regression accuracy: ..."; raw-JSON lines 987-992
# for circuit (c), 1058-1063 for (d)) - deterministic functions of the truth tables, so they are known answers.
_SI_PRINTED = [
    (1.000, [0.3333333333333333, 0.3333333333333333, 0.3333333333333333], 0.500),
    (1.000, [0.21854635407652215, 0.21854635407652215, 0.5629072918469556], 0.750),
    (0.811, [0.0962198571076538, 0.37685891933722027, 0.0962198571076538, 0.24197949090660467], 0.875),
    (1.000, [0.3425060887628776, 0.09250608876287761, 0.47248173371136726, 0.09250608876287761], 0.625),
    (0.954, [0.6361993267150806, 0.04121479843161109, 0.13619932671508067, 0.09960575263158164, 0.041214798431611056], 0.938),
    (1.000, [0.20314536459234778, 0.0937092708153044, 0.0937092708153044, 0.2031453645923478, 0.2031453645923478,
             0.2031453645923478], 0.500),
]


@pytest.mark.parametrize("circuit", range(6))
def test_si_circuit_printed_known_answers(circuit):
    """H(Y), the Shapley values (to the 16 digits the notebook printed) and the logistic-regression accuracy of every SI circuit,
    from the oracle's restatements on the PRODUCT's truth table (dib_amd.data.truth_table, the data path `train.py --dataset
    boolean_circuit` uses)."""
    import dib_amd
    spec = orc.SI_CIRCUITS[circuit]
    n = sum(1 for v in spec if isinstance(v, int))
    table = dib_amd.data.truth_table(spec, n)
    xo, yo = orc.boolean_circuit_truth_table(spec, n)
    x, y = 2 * table[:, :n] - 1, table[:, -1].astype(np.int64)
    assert np.array_equal(x, xo) and np.array_equal(y, yo)
    hy, shap, acc = _SI_PRINTED[circuit]
    entropy_y = orc.entropy_bits(np.bincount(y, minlength=2) / len(y))
    assert f"{entropy_y:.3f}" == f"{hy:.3f}"
    got = orc.shapley_values_bits(x, y)
    assert np.abs(got - np.array(shap)).max() < 1e-14, (got, shap)
    assert abs(got.sum() - entropy_y) < 1e-12                     # efficiency: the values share out all of H(Y)
    # "Logistic regression accuracy" (the notebook: LogisticRegression(random_state=0, penalty='none') on the 0/1 truth table)
    from sklearn.linear_model import LogisticRegression
    clf = LogisticRegression(random_state=0, penalty=None, max_iter=10000).fit(table[:, :n], y)
    assert f"{np.average(clf.predict(table[:, :n]) == y):.3f}" == f"{acc:.3f}"
