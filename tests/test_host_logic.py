"""CPU: host logic of the drop-in surface (fit loop, callbacks, History accounting, compression-matrix
callback, Keras-shaped objects) exercised with the TEST-ONLY oracle engine (tests/_oracle_engine.py)."""
import os

import numpy as np
import pytest

import dib_oracle as orc
from _helpers import spec_kwargs
from _oracle_engine import OracleEngine


def _model(spec, **kw):
    import dib_amd
    m = dib_amd.DistributedIBNet(**spec_kwargs(spec), **kw)
    m._make_engine = lambda: OracleEngine(**m._spec_kwargs(), init_seed=m.init_seed)
    return m


def _si_circuit(copies=4):
    x, y = orc.boolean_circuit_truth_table([0, 1, 2, 3, [0, 2, 0], [2, 4, 3], [0, 5, 1]], 4)
    return np.tile(x, (copies, 1)).astype(np.float32), np.tile(y, copies).astype(np.float32)


def test_fit_history_contract_matches_oracle_fit():
    """History keys/units of reference train.py:169-172 and bit-level agreement of the host loop
    (shuffle order, partial last batch, beta schedule, validation pass) with oracle.fit."""
    import dib_amd
    spec = orc.DIBSpec([1, 1, 1, 1], [8], [8], 1, feature_embedding_dimension=4)
    x, y = _si_circuit()
    model = _model(spec, noise_seed=3, shuffle_seed=5, init_seed=1)
    opt = dib_amd.optimizers.get("adam")
    opt.learning_rate = 1e-2
    model.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-3, 1.0, 1, 3)
    hist = model.fit(x, y, epochs=4, shuffle=True, batch_size=24, callbacks=[cb], verbose=False,
                     validation_data=(x[:20], y[:20]))
    p = orc.glorot_uniform_init(spec, 1)
    ref = orc.fit(spec, p, x, y, epochs=4, batch_size=24, loss_kind="bce_logits",
                  beta_fn=lambda e: orc.beta_schedule(e, 1e-3, 1.0, 1, 3), lr=1e-2, validation_data=(x[:20], y[:20]),
                  noise_seed=3, shuffle_seed=5, metrics=["accuracy"])
    assert set(hist.history) == set(ref) == {"loss", "val_loss", "beta", "val_beta", "accuracy", "val_accuracy",
                                             *[f"KL{f}" for f in range(4)], *[f"val_KL{f}" for f in range(4)]}
    for k in ref:
        assert np.allclose(hist.history[k], ref[k], rtol=1e-9, atol=1e-12), k
    assert hist.epoch == [0, 1, 2, 3]
    # train.py:169-178 post-processing
    beta_series, kl_bits, loss_bits = orc.postprocess_history(hist.history, 4, True)
    assert kl_bits.shape == (4, 4) and np.all(kl_bits >= 0)
    assert np.allclose(loss_bits * np.log(2), np.array(hist.history["loss"]) - beta_series * kl_bits.sum(-1) * np.log(2),
                       atol=1e-6)


def test_fit_evaluates_full_validation_batches_together():
    """fit's validation pass (reference train.py:157-166 validation_data=): full batches are independent and every History
    quantity is linear in per-row sums, so k of them run as one launch set of k * batch_size rows scaled by 1 / batch_size and
    counted as k steps (model.validation_merge_rows); a ragged last batch stays on its own.  Same History as the oracle's
    batch-by-batch pass, fewer engine calls."""
    import dib_amd
    spec = orc.DIBSpec([1, 1, 1, 1], [8], [8], 1, feature_embedding_dimension=4)
    x, y = _si_circuit()
    hists, calls = [], []
    for merge in (1024, 0):
        model = _model(spec, noise_seed=3, shuffle_seed=5, init_seed=1)
        model.validation_merge_rows = merge
        opt = dib_amd.optimizers.get("adam")
        opt.learning_rate = 1e-2
        model.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
        eng = model._ensure_engine()
        seen = []
        inner = eng.eval_step
        eng.eval_step = lambda *a, _inner=inner, _seen=seen, **k: (_seen.append((a[3], a[4], k.get("inv_global_batch"))), _inner(*a, **k))[1]
        hists.append(model.fit(x, y, epochs=2, shuffle=True, batch_size=24, verbose=False, validation_data=(x, y)).history)
        calls.append(seen)
    p = orc.glorot_uniform_init(spec, 1)
    ref = orc.fit(spec, p, x, y, epochs=2, batch_size=24, loss_kind="bce_logits", beta_fn=lambda e: 1.0, lr=1e-2,
                  validation_data=(x, y), noise_seed=3, shuffle_seed=5, metrics=["accuracy"])
    assert len(x) == 64
    assert calls[0] == [(0, 48, 1.0 / 24), (48, 16, 1.0 / 16)] * 2                       # 2 full batches together, then the ragged one
    assert calls[1] == [(0, 24, 1.0 / 24), (24, 24, 1.0 / 24), (48, 16, 1.0 / 16)] * 2
    for k in ref:
        if k in ("beta", "val_beta"):
            continue
        assert np.allclose(hists[0][k], ref[k], rtol=1e-9, atol=1e-12), k
        assert np.allclose(hists[1][k], ref[k], rtol=1e-9, atol=1e-12), k


def test_evaluate_merges_full_batches_like_fit():
    """model.evaluate (Keras surface of the reference's `model.evaluate`-style checks): the same batching rule as fit's validation
    pass - identical logs with validation_merge_rows = 0 and 1024, fewer engine calls."""
    import dib_amd
    spec = orc.DIBSpec([1, 1, 1, 1], [8], [8], 1, feature_embedding_dimension=4)
    x, y = _si_circuit()
    logs, calls = [], []
    for merge in (1024, 0):
        model = _model(spec, noise_seed=3, shuffle_seed=5, init_seed=1)
        model.validation_merge_rows = merge
        model.compile(optimizer=dib_amd.optimizers.get("adam"), loss=dib_amd.losses.BinaryCrossentropy(from_logits=True),
                      metrics=["accuracy"])
        eng = model._ensure_engine()
        seen = []
        inner = eng.eval_step
        eng.eval_step = lambda *a, _inner=inner, _seen=seen, **k: (_seen.append((a[3], a[4])), _inner(*a, **k))[1]
        logs.append(model.evaluate(x, y, batch_size=24))
        calls.append(seen)
    assert calls[0] == [(0, 48), (48, 16)] and calls[1] == [(0, 24), (24, 24), (48, 16)]
    assert set(logs[0]) == set(logs[1])
    for k in logs[0]:
        assert np.isclose(logs[0][k], logs[1][k], rtol=1e-9, atol=1e-12), k


def test_beta_variable_and_annealing_callback():
    import dib_amd
    spec = orc.DIBSpec([1, 1], [4], [4], 1, feature_embedding_dimension=4)
    model = _model(spec)
    assert float(model.beta.value()) == 1.0  # reference models.py:86
    model.beta.assign(0.25)
    assert model._ensure_engine().get_beta() == 0.25
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-4, 3.0, 10, 100)
    cb.set_model(model)
    cb.on_epoch_begin(0)
    assert float(model.beta.value()) == pytest.approx(1e-4, rel=1e-6)
    cb.on_epoch_begin(110)
    assert float(model.beta.value()) == pytest.approx(3.0, rel=1e-5)
    assert model.feature_dimensionalities == [1, 1] and model.number_features == 2


def test_compile_accepts_reference_style_arguments():
    import dib_amd
    spec = orc.DIBSpec([1, 1], [4], [4], 3, feature_embedding_dimension=4)
    m = _model(spec)
    m.compile(optimizer="adam", loss=dib_amd.losses.SparseCategoricalCrossentropy(from_logits=True), metrics=["accuracy"])
    assert m.optimizer.learning_rate == 1e-3 and m.optimizer.epsilon == 1e-7 and m.loss.kind == "sparse_cce_logits"
    m.compile(optimizer=dib_amd.optimizers.SGD(0.1), loss="mse")
    assert m.loss.kind == "mse"
    with pytest.raises(NotImplementedError):
        m.compile(optimizer="adam", loss="infonce")
    with pytest.raises(ValueError):
        m.compile(optimizer="lion", loss="mse")
    with pytest.raises(RuntimeError):
        _model(spec).fit(np.zeros((2, 2)), np.zeros(2))


def test_save_compression_matrices_callback(tmp_path):
    import dib_amd
    spec = orc.DIBSpec([1, 1, 1, 1], [8], [8], 1, feature_embedding_dimension=4)
    x, y = _si_circuit()
    model = _model(spec, init_seed=2)
    model.compile(optimizer="adam", loss=dib_amd.losses.BinaryCrossentropy(from_logits=True))
    cb = dib_amd.SaveCompressionMatricesCallback(2, x, x, str(tmp_path))
    model.fit(x, y, epochs=3, batch_size=32, callbacks=[cb], verbose=False)
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 4 and all(f.startswith("feature_") and "log10beta_0.000" in f for f in files)  # epochs 0, 2 same beta
    mat = cb.matrices[(2, 0)]
    # Boolean inputs have 2 unique values -> 2x2 matrix, unit diagonal, symmetric (visualization.py:17-22)
    assert mat.shape == (2, 2) and np.allclose(np.diag(mat), 1.0) and np.allclose(mat, mat.T)
    p = model._engine.p
    ref = orc.compression_matrix(spec, p, 0, np.array([[-1.0], [1.0]]))
    assert np.allclose(mat, ref, atol=1e-9)


def test_info_plane_plot_and_utils(tmp_path):
    import dib_amd
    kl = np.abs(np.random.default_rng(0).standard_normal((40, 3)))
    out = dib_amd.visualization.save_distributed_info_plane(kl, np.linspace(1, 0, 40), str(tmp_path), entropy_y=0.8)
    assert os.path.exists(out)
    assert dib_amd.utils.compute_entropy([0, 0, 1, 1]) == pytest.approx(1.0)
    pe = dib_amd.PositionalEncoding([2, 4])
    assert np.allclose(pe(np.array([[0.5]])), [[0.5, np.sin(1.0), np.sin(2.0)]])


def test_synthetic_tabular_dataset_definition():
    import dib_amd
    d = dib_amd.data.DATASETS["synthetic_tabular"](synthetic_rows=4096, synthetic_features=64)
    assert d["x_train"].shape == (4096, 64) and d["x_train"].dtype == np.float32
    assert set(np.unique(d["y_train"])) == {0.0, 1.0} and 0.3 < d["y_train"].mean() < 0.7
    rng = np.random.default_rng(20241008)
    x = rng.standard_normal((4096, 64), dtype=np.float32)
    assert np.array_equal(x, d["x_train"])


def test_torch_cpu_restatement_agrees_with_numpy_oracle():
    """The cpu_baseline implementation (autograd backward) cross-checks the hand-derived oracle backward."""
    import torch
    from dib_torch_cpu import TorchCpuDIB
    from _helpers import random_params
    spec = orc.DIBSpec([2, 1, 1], [16, 8], [12], 1, feature_embedding_dimension=4)
    p = random_params(spec, 2)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((9, 4))
    y = rng.integers(0, 2, (9, 1)).astype(np.float64)
    eps = orc.philox_normal_all(1, 2, np.arange(9), 3, 4)
    c = orc.forward(spec, p, x, eps)
    task, g, _ = orc.backward(spec, p, x, y, c, 0.4, "bce_logits")
    m = TorchCpuDIB(spec, p, dtype=torch.float64)
    t_task, t_kl, t_grads = m.train_step(torch.tensor(x), torch.tensor(y), torch.tensor(eps), 0.4, "bce_logits")
    assert abs(t_task - task) < 1e-12 and np.allclose(t_kl.numpy(), c.kl, atol=1e-12)
    for a, b in zip(t_grads, g.tensors()):
        assert np.allclose(a.numpy(), b, atol=1e-12)
    st = orc.adam_init(p)
    orc.adam_keras_step(p, g, st)
    for a, b in zip(m.tensors(), p.tensors()):
        assert np.allclose(a.detach().numpy(), b, atol=1e-12)


def test_torch_cpu_batched_equals_loop():
    """The full-size GPU parity tests use the feature-batched (bmm) + row-chunked form of the float64 PyTorch-CPU checker;
    pin it on the reference-shaped per-feature loop form (models.py:105-122) and on the NumPy oracle."""
    import torch
    import dib_oracle as orc
    from dib_torch_cpu import TorchCpuDIB
    spec = orc.DIBSpec([1] * 6, [16, 12], [10], 1, feature_embedding_dimension=4)
    p = orc.glorot_uniform_init(spec, 3)
    rng = np.random.default_rng(0)
    for t in p.tensors():
        if t.ndim == 1:
            t[:] = 0.1 * rng.standard_normal(t.shape)
    B = 50
    x = rng.standard_normal((B, 6))
    y = (rng.random((B, 1)) > 0.5).astype(np.float64)
    eps = rng.standard_normal((B, 6, 4))
    xt, yt, et = torch.tensor(x), torch.tensor(y), torch.tensor(eps)
    a = TorchCpuDIB(spec, p, dtype=torch.float64)
    t0, k0, g0, p0 = a.loss_and_grads(xt, yt, et, 0.3, "bce_logits")
    t1, k1, g1, p1 = a.loss_and_grads(xt, yt, et, 0.3, "bce_logits", chunk=16, batched=True)
    assert abs(t0 - t1) < 1e-14 and torch.allclose(k0, k1, rtol=0, atol=1e-14) and torch.allclose(p0, p1, rtol=0, atol=1e-13)
    for u, v in zip(g0, g1):
        assert torch.allclose(u, v, rtol=0, atol=1e-14)
    c = orc.forward(spec, p, x, eps)
    task, grads, _ = orc.backward(spec, p, x, y, c, 0.3, "bce_logits")
    assert abs(task - t1) < 1e-12 and np.abs(c.kl - k1.numpy()).max() < 1e-12
    for u, v in zip(g1, [np.asarray(t) for f in range(6) for l in range(3) for t in (grads.enc_W[f][l], grads.enc_b[f][l])]
                    + [np.asarray(t) for l in range(2) for t in (grads.int_W[l], grads.int_b[l])]):
        assert np.abs(u.numpy() - v).max() < 1e-12
    # train_step = loss_and_grads + Keras Adam, unchanged behaviour
    b = TorchCpuDIB(spec, p, dtype=torch.float64)
    b.train_step(xt, yt, et, 0.3, "bce_logits", lr=1e-3)
    a.apply_adam(g1, 1e-3)
    for u, v in zip(a.tensors(), b.tensors()):
        assert torch.allclose(u, v, rtol=0, atol=1e-14)


def test_compression_matrix_figure_content_matches_reference_function(monkeypatch, tmp_path):
    """SaveCompressionMatricesCallback / visualization.save_compression_matrices: the CONTENT of the PNG (matrix handed to
    imshow, side-plot arrays) against what the reference's own function draws (visualization.py:14-81 executed with a
    recording matplotlib stand-in, tests/golden/make_golden_compression.py) - not just that a file appears."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from plt_recorder import Recorder
    from make_golden_compression import fake_encoder
    import dib_amd
    from dib_amd import visualization
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "compression_matrix_figure.npz"))
    # (a) fewer than 10 unique raw values: histogram mode, deterministic
    rec = Recorder()
    monkeypatch.setattr(visualization, "_plt", lambda: rec)
    m = visualization.save_compression_matrices(fake_encoder, g["inp_a"], str(tmp_path / "a.png"), inp_features_raw=g["raw_a"],
                                                feature_label="Feature 3")
    c = rec.content()
    assert np.abs(m - g["a_matrix"]).max() < 1e-12 and np.abs(c[((1, 1), "imshow")][0] - g["a_matrix"]).max() < 1e-12
    assert np.array_equal(c[((1, 0), "barh")][0], g["a_barh_y"]) and np.allclose(c[((1, 0), "barh")][1], g["a_barh_w"], atol=1e-15)
    assert np.array_equal(c[((0, 1), "bar")][0], g["a_bar_x"]) and np.allclose(c[((0, 1), "bar")][1], g["a_bar_h"], atol=1e-15)
    assert rec.saved == [str(tmp_path / "a.png")]
    # (b) continuous feature: 128 random rows sorted by raw value (same np.random stream as the reference)
    rec = Recorder()
    monkeypatch.setattr(visualization, "_plt", lambda: rec)
    np.random.seed(123)
    m = visualization.save_compression_matrices(fake_encoder, g["inp_b"], str(tmp_path / "b.png"), inp_features_raw=g["raw_b"])
    c = rec.content()
    assert np.abs(m - g["b_matrix_intent"]).max() < 1e-12, "matrix of the sorted random selection (reference intent, defect A14 fixed)"
    assert np.abs(m - g["b_matrix_literal"]).max() > 1e-3    # the reference literally draws the first 128 dataset rows
    assert np.array_equal(c[((1, 0), "plot")][0], g["b_left_x"]) and np.array_equal(c[((1, 0), "plot")][1], g["b_left_y"])
    assert np.array_equal(c[((0, 1), "plot")][0], g["b_top_x"]) and np.array_equal(c[((0, 1), "plot")][1], g["b_top_y"])


def test_double_pendulum_data_path_matches_the_reference_executed(tmp_path):
    """BASELINE config 2's data source.  tests/golden/pendulum.npz = the reference's simulate_pendulum.py executed with a
    seeded global NumPy stream, then its data.fetch_double_pendulum on the file it wrote (make_golden_pendulum.py).
    dib_amd.simulate_pendulum fed the same stream (a RandomState adapter: uniform() / randint(2), sign drawn before the validity
    test like the reference) reproduces the trajectories; dib_amd.data.fetch_double_pendulum builds the same (x, y) pairs -
    except that it holds out the FIRST tenth for validation where the reference's np.split trains on it."""
    import dib_amd
    from dib_amd import simulate_pendulum
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pendulum.npz"))

    class LegacyStream:  # the reference draws from the global np.random stream (simulate_pendulum.py:58,63)
        def __init__(self, seed):
            self.r = np.random.RandomState(seed)

        def uniform(self):
            return self.r.uniform()

        def integers(self, n):
            return self.r.randint(n)

    prm = {k[len("param_"):]: g[k].item() for k in g.files if k.startswith("param_")}
    prm["number_trajectories"] = int(prm["number_trajectories"])
    traj = simulate_pendulum.simulate_double_pendulum(str(tmp_path), prm, rng=LegacyStream(int(g["seed"])), save=True)
    assert traj.shape == g["trajectories"].shape
    # same stream, same initial conditions, same integrator: most trajectories agree to the last bit; where the right-hand side
    # rounds one operation differently (`z**2` vs `z*z`) the chaotic dynamics amplify the last bit to ~3e-9 over these 3 s
    assert np.abs(traj - g["trajectories"]).max() < 1e-6
    assert np.array_equal(traj[:, 0, [1, 3]] == 0, g["trajectories"][:, 0, [1, 3]] == 0)
    d = dib_amd.data.fetch_double_pendulum(data_path=str(tmp_path), pendulum_time_delta=float(g["time_delta"]))
    assert d["feature_dimensionalities"] == list(g["feature_dimensionalities"]) and d["loss"] == "infonce"
    # reference "train" = first tenth, "valid" = the rest; here the first tenth is the held-out set
    for mine, theirs in (("x_valid", "x_train"), ("y_valid", "y_train"), ("x_train", "x_valid"), ("y_train", "y_valid")):
        assert d[mine].shape == g[theirs].shape, (mine, d[mine].shape, g[theirs].shape)
        assert np.abs(d[mine] - g[theirs].astype(np.float32)).max() < 1e-6, mine
    # energy is conserved along every stored trajectory (the simulator's acceptance test)
    e = simulate_pendulum.total_energy(traj.reshape(-1, 4)).reshape(traj.shape[:2])
    assert np.abs(e / e[:, :1] - 1).max() < 1e-3


def test_info_plane_figure_content_matches_reference_function(monkeypatch, tmp_path):
    """visualization.save_distributed_info_plane (History post-processing -> figure, reference visualization.py:83-113 executed
    with the recording matplotlib stand-in, tests/golden/make_golden_misc.py): the sieve factor, the start index and every
    array handed to ax.plot on the main and the twin axis."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from plt_recorder import Recorder
    from make_golden_misc import info_plane_inputs
    from dib_amd import visualization
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "misc.npz"))
    kl, loss = info_plane_inputs()
    for tag, (k, l, hy) in {"long": (kl, loss, 0.758), "short": (kl[:40, :1], loss[:40], None)}.items():
        rec = Recorder()
        monkeypatch.setattr(visualization, "_plt", lambda: rec)
        saved = visualization.save_distributed_info_plane(k, l, str(tmp_path / tag), entropy_y=hy)
        main_plots, twin_plots = rec.plots("main"), rec.plots(("main", "twin"))
        assert len(main_plots) == int(g[f"ip_{tag}_n_main"]) and len(twin_plots) == int(g[f"ip_{tag}_n_twin"])
        for i, (x, y) in enumerate(main_plots):
            assert np.array_equal(x, g[f"ip_{tag}_main{i}_x"]) and np.array_equal(y, g[f"ip_{tag}_main{i}_y"]), (tag, "main", i)
        for i, (x, y) in enumerate(twin_plots):
            assert np.array_equal(x, g[f"ip_{tag}_twin{i}_x"]) and np.array_equal(y, g[f"ip_{tag}_twin{i}_y"]), (tag, "twin", i)
        assert os.path.basename(saved) == os.path.basename(str(g[f"ip_{tag}_saved"][0])) == "distributed_info_plane.png"


def test_chaos_maps_match_the_reference_executed():
    """dib_amd.chaos_data.generate_data with seed=None draws its initial condition from NumPy's global stream like the
    reference (chaos/chaos_data.py:3-55): under the same np.random.seed the trajectories are bit-identical to the reference's
    (executed by tests/golden/make_golden_misc.py) - 1900 iterations of a chaotic map leave no room for a different rounding."""
    from dib_amd import chaos_data
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "misc.npz"))
    for name, prm in (("logistic", {}), ("henon", {}), ("ikeda", {}), ("logistic_r4", {"r": 4.0})):
        np.random.seed(int(g["seed"]))
        got = chaos_data.generate_data(name.split("_")[0], number_iterations=400, number_skip_iterations=1500, **prm)
        assert got.shape == g[f"chaos_{name}"].shape
        assert np.array_equal(got, g[f"chaos_{name}"]), (name, np.abs(got - g[f"chaos_{name}"]).max())


def test_history_postprocessing_matches_the_reference_statements():
    """train.postprocess_history against reference train.py:168-178 - the statements themselves, cut out of the reference
    script and executed on a synthetic `history.history` (tests/golden/make_golden_misc.py): loss without beta * sum KL, KL
    and info-based losses in bits, float32 / float64 dtypes included.  The validation series are this project's addition
    (the reference leaves val_loss raw and never builds kl_series_validation): checked by construction."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_misc import history_inputs
    from dib_amd import train
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "misc.npz"))
    for tag, info_based in (("info", True), ("plain", False)):
        out = train.postprocess_history(history_inputs(), 3, info_based)
        assert out["loss"].dtype == g[f"hist_{tag}_loss"].dtype == np.float32
        assert np.array_equal(out["loss"], g[f"hist_{tag}_loss"]) and np.array_equal(out["beta"], g[f"hist_{tag}_beta"])
        assert np.array_equal(out["kl_bits"], g[f"hist_{tag}_kl_bits"])
        h = history_inputs()
        raw = g[f"hist_{tag}_loss_validation_raw"]
        want = raw - np.float32(h["val_beta"]) * sum(np.array(h[f"val_KL{f}"]) for f in range(3))
        want = want / np.log(2) if info_based else want
        assert np.allclose(out["loss_validation"], want, rtol=1e-6, atol=1e-6)


def test_train_script_end_to_end_on_the_checker_engine(monkeypatch, tmp_path):
    """dib_amd.train.main (the reference's `python train.py --dataset boolean_circuit ...`, train.py:118-178) from argument
    parsing to history.npz and the info-plane figure, with the TEST-ONLY oracle engine injected - the host logic of the script
    path without a GPU (the same call runs on the device in tests/test_gpu_parity.py)."""
    import dib_amd
    from dib_amd import models, train

    monkeypatch.setattr(models.DistributedIBNet, "_make_engine",
                        lambda self: OracleEngine(**self._spec_kwargs(), init_seed=self.init_seed))
    hist = train.main(["--dataset", "boolean_circuit", "--number_pretraining_epochs", "1", "--number_annealing_epochs", "2",
                       "--batch_size", "256", "--artifact_outdir", str(tmp_path), "--save_compression_matrices_frequency", "2",
                       "--feature_encoder_architecture", "8", "8", "--integration_network_architecture", "16",
                       "--feature_embedding_dimension", "4"])
    h = hist.history
    assert len(h["loss"]) == 3 and np.isfinite(h["loss"]).all() and np.isfinite(h["val_loss"]).all()
    z = np.load(os.path.join(str(tmp_path), "history.npz"))
    want = train.postprocess_history(h, 10, True)
    assert z["kl_bits"].shape == (3, 10) and np.array_equal(z["loss"], want["loss"]) and np.array_equal(z["beta"], want["beta"])
    assert np.allclose(z["beta"], np.float32(h["beta"])) and z["beta"][0] < z["beta"][-1]     # the annealing ramp ran
    assert os.path.exists(os.path.join(str(tmp_path), "distributed_info_plane.png"))
    assert any(f.startswith("feature_0_log10beta") for f in os.listdir(str(tmp_path)))


def test_per_particle_feature_set_matches_the_notebook_function_executed():
    """dib_amd.set_transformer.convert_to_per_particle_feature_set (the PRODUCT's copy of notebook cell 6) against the notebook's
    own function executed on the same raw positions / types (tests/golden/make_golden_set_transformer.py -> ref_feats)."""
    from dib_amd.set_transformer import convert_to_per_particle_feature_set
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "set_transformer_forward.npz"))
    got = convert_to_per_particle_feature_set(g["raw_pos"], g["raw_types"], number_particles_to_use=6)
    assert got.shape == g["ref_feats"].shape and np.abs(got - g["ref_feats"]).max() < 1e-12


def test_streaming_projection_launch_record_only_claims_shapes_the_kernel_covers():
    """`_SkinnyKGemm.fits` (the host-side rule in front of dib_gemm_skinny_k) must mirror the entry's own refusals
    (include/dib_st.h: K <= 32, K % 4 == 0, N % 32 == 0, uniform groups, no activation / act' mask, modes 0 / 1 only)."""
    from dib_amd._gemm_plan import _SkinnyKGemm, _d
    ok = [_d(0, 32, 0, 1536, 0, 1536, 4096, 1536, 32, bias_off=5) for _ in range(3)]
    assert _SkinnyKGemm.fits(0, ok) and _SkinnyKGemm.fits(1, ok[:1])
    assert not _SkinnyKGemm.fits(2, ok)                                   # weight gradients stay on the tiled kernel
    assert not _SkinnyKGemm.fits(0, ok, act=1)                            # no activation epilogue
    assert not _SkinnyKGemm.fits(1, ok, aux=object())                     # no act' mask
    for bad in (dict(K=36), dict(K=30), dict(N=1000), dict(K=0)):
        d = dict(ok[0], **bad)
        assert not _SkinnyKGemm.fits(0, [d]), bad
    assert not _SkinnyKGemm.fits(0, [ok[0], dict(ok[0], M=2048)])         # all groups share M, N, K
    rec = _SkinnyKGemm(0, ok, None, None, None)
    assert (rec.mode, rec.n, rec.max_m, rec.max_n) == (0, 3, 4096, 1536) and rec.host["K"].tolist() == [32, 32, 32]


def test_batch_stream_is_tf_data_shuffle_buffer_over_the_repeating_sequential_stream():
    """reference train.py:226-227 / 233-234: `from_tensor_slices(...).repeat().shuffle(min(n, 10_000)).batch(B)` - a 10 000-element
    shuffle BUFFER over the sequential, repeating stream, not a permutation of the dataset (VERDICT r04 missing item 2).  The
    pendulum rows are time-ordered, trajectory after trajectory (data.py:122-123): a batch's in-batch negatives come from a
    window of the sequential order.  Checked: (1) the product's vectorised stream = the oracle's one-element-at-a-time restatement
    of tf.data's algorithm, however the draws are grouped into calls; (2) the conservation law of a shuffle buffer - emitted
    elements + buffer content = a prefix of the source, nothing skipped, nothing twice; (3) draw t never runs ahead of source
    position t + buffer, and the time an element waits in the buffer is geometric with mean = buffer; (4) a dataset that fits
    the buffer is NOT reshuffled pass by pass either: repeat() comes first, so passes blend - the first draw is uniform over the
    whole dataset, a row can recur before every row was seen, but the counts stay balanced by the conservation law."""
    from collections import Counter
    from dib_amd.infonce import SHUFFLE_BUFFER, _BatchStream
    from infonce_loop_oracle import BatchStream
    assert SHUFFLE_BUFFER == 10_000
    # (1) product == literal restatement, any call grouping
    for n, bs, nbuf in ((50, 16, 10_000), (7, 5, 10_000), (1000, 128, 100), (30_000, 64, 10_000), (5, 64, 3)):
        lit = BatchStream(n, bs, 3, nbuf)
        want = np.stack([lit.next() for _ in range(40)])
        a = _BatchStream(n, bs, 3, nbuf)
        assert np.array_equal(np.stack([a.next() for _ in range(40)]), want)
        b = _BatchStream(n, bs, 3, nbuf)
        assert np.array_equal(np.concatenate([b.next_batches(7), b.next_batches(1), b.next_batches(32)]), want)
        assert sorted(b.buf.tolist()) == sorted(lit.buffer) and b.pos == lit.source_position
    # (2) + (3): the pendulum's size, ordered rows (row index = source position for the first pass)
    n, bs, nb = 240_000, 128, 500
    s = _BatchStream(n, bs, 11)
    assert s.nbuf == 10_000
    rows = s.next_batches(nb)
    T = nb * bs
    assert T + s.nbuf <= n                                         # still in the first pass: rows ARE source positions
    flat = rows.reshape(-1)
    assert np.array_equal(np.sort(np.concatenate([flat, s.buf])), np.arange(T + s.nbuf))
    t = np.arange(T)
    assert (flat < t + s.nbuf).all()                               # never ahead of the source
    lag = t + s.nbuf - 1 - flat                                    # draws an element waited beyond the minimum
    steady = lag[5 * s.nbuf:]
    assert abs(steady.mean() / s.nbuf - 1.0) < 0.06                # geometric residence, mean = buffer size
    # every batch lives in a window of the sequential order: >= 90 % of its rows within the newest 4 buffers
    in_window = (lag.reshape(nb, bs) < 4 * s.nbuf).mean(axis=1)
    assert in_window.min() >= 0.9, in_window.min()
    # a permutation of the dataset would spread a batch over all 240 000 rows: here its spread is a few buffers
    spread = np.median(rows.max(axis=1) - rows.min(axis=1))
    assert spread < 6 * s.nbuf, spread
    # (4) dataset smaller than the buffer: buffer = n, conservation law across many passes, passes blend
    n, bs = 300, 32
    s = _BatchStream(n, bs, 5)
    assert s.nbuf == n
    out = s.next_batches(200).reshape(-1)
    have = Counter(out.tolist()) + Counter(s.buf.tolist())
    want = Counter((np.arange(len(out) + n) % n).tolist())
    assert have == want
    firsts = {int(_BatchStream(n, bs, k).next()[0]) for k in range(200)}
    assert len(firsts) > 100 and max(firsts) > 250                 # first draw: anywhere in the dataset
    first_pass = out[:n]
    assert len(set(first_pass.tolist())) < n                       # not a permutation: repeat() precedes shuffle()
    # the validation pass of every boundary is a fresh iterator (train.py:262): same rows reachable, different order
    v0 = _BatchStream(5000, 128, [7, 0]).next_batches(40)
    v1 = _BatchStream(5000, 128, [7, 1]).next_batches(40)
    assert not np.array_equal(v0, v1) and v0.max() < 5000
