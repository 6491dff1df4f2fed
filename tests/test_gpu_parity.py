"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (float32 device arithmetic vs float64 oracle): activations/predictions 2e-4 abs+rel,
per-feature KL 1e-3 nats absolute (BASELINE.json north_star), gradients 2e-4 relative to the
gradient's max-abs.
"""
import ctypes
import os
import zlib

import numpy as np
import pytest
import torch

import dib_oracle as orc
from _helpers import DISPATCH_PATHS, FUSED_ELIGIBLE, SPECS, dispatch_path, flat_to_params, params_to_flat, random_params, spec_kwargs

pytestmark = pytest.mark.gpu


def _engine(spec, seed=0):
    from dib_amd.engine import HipEngine
    eng = HipEngine(**spec_kwargs(spec), init_seed=seed)
    p = random_params(spec, seed)
    eng.set_flat_params(params_to_flat(eng.blocks, p, eng.params.numel()))
    # oracle sees exactly the float32-rounded parameters
    return eng, flat_to_params(eng.blocks, eng.get_flat_params(), spec)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 70, 45), (1, 1, 1), (37, 129, 5), (300, 257, 130),
                                   (64, 5, 2048), (1000, 64, 128)])
def test_gemm_vs_numpy(mode, M, N, K):
    from dib_amd import _lib
    lib = _lib.load_library()
    rng = np.random.default_rng(M * 1000 + N * 10 + K + mode)
    dev = torch.device("cuda:0")
    if mode == 0:
        A, B = rng.standard_normal((M, K)), rng.standard_normal((K, N))
        bias = rng.standard_normal(N)
        ref = np.maximum(A @ B + bias, 0)
    elif mode == 1:
        A, B = rng.standard_normal((M, K)), rng.standard_normal((N, K))
        aux = rng.standard_normal((M, N))
        ref = (A @ B.T) * (aux > 0)
    else:
        A, B = rng.standard_normal((K, M)), rng.standard_normal((K, N))
        ref = A.T @ B
        ref_bias = B.sum(0)
    At = torch.tensor(A, dtype=torch.float32, device=dev)
    Bt = torch.tensor(B, dtype=torch.float32, device=dev)
    Ct = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    desc = torch.zeros(256, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if mode == 0:
        bt = torch.tensor(bias, dtype=torch.float32, device=dev)
        rc = lib.dib_gemm(0, M, N, K, _ptr(At), K, _ptr(Bt), N, _ptr(Ct), N, _ptr(bt), None, 0, 1, _ptr(desc), st)
    elif mode == 1:
        xt = torch.tensor(aux, dtype=torch.float32, device=dev)
        rc = lib.dib_gemm(1, M, N, K, _ptr(At), K, _ptr(Bt), K, _ptr(Ct), N, None, _ptr(xt), N, 1, _ptr(desc), st)
    else:
        bt = torch.full((N,), float("nan"), dtype=torch.float32, device=dev)
        rc = lib.dib_gemm(2, M, N, K, _ptr(At), M, _ptr(Bt), N, _ptr(Ct), N, _ptr(bt), None, 0, 0, _ptr(desc), st)
    assert rc == 0
    torch.cuda.synchronize()
    got = Ct.cpu().numpy()
    scale = np.abs(ref).max() + 1e-6
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / scale < 2e-5
    if mode == 2:
        gb = bt.cpu().numpy()
        assert np.abs(gb - ref_bias).max() / (np.abs(ref_bias).max() + 1e-6) < 2e-5


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("M,N,K,groups,pad", [(4096, 1536, 32, 3, 0), (1000, 96, 32, 1, 0), (77, 32, 12, 2, 0), (64, 128, 4, 1, 0),
                                              (513, 160, 20, 2, 3), (1, 32, 32, 1, 0), (300000, 256, 32, 1, 0)])
def test_skinny_k_gemm_vs_numpy(mode, M, N, K, groups, pad):
    """dib_gemm_skinny_k (the streaming kernel of the set transformer's q / k / v projections and of the context gradient):
    every group against float64 numpy - ragged row counts, N not a multiple of 128, K < 32, leading dimensions wider than
    the matrices (pad: unaligned rows take the same path), a 300 000-row launch whose output is stored non-temporally."""
    from dib_amd import _lib
    from dib_amd._gemm_plan import DESC
    lib = _lib.load_library()
    rng = np.random.default_rng(M + 7 * N + 13 * K + mode)
    dev = torch.device("cuda:0")
    lda, ldc = K + pad, N + pad
    ldb = (N if mode == 0 else K) + pad
    brows = K if mode == 0 else N
    A = rng.standard_normal((groups, M, lda)).astype(np.float32)
    B = rng.standard_normal((groups, brows, ldb)).astype(np.float32)
    bias = rng.standard_normal((groups, N)).astype(np.float32)
    At, Bt, bt = (torch.from_numpy(x).to(dev) for x in (A, B, bias))
    Ct = torch.full((groups, M, ldc), float("nan"), dtype=torch.float32, device=dev)
    arr = np.zeros(groups, dtype=DESC)
    for g in range(groups):
        arr[g]["a_off"], arr[g]["b_off"], arr[g]["c_off"] = g * M * lda, g * brows * ldb, g * M * ldc
        arr[g]["bias_off"] = g * N if (mode == 0 and g != 1) else -1        # group 1: no bias
        arr[g]["lda"], arr[g]["ldb"], arr[g]["ldc"] = lda, ldb, ldc
        arr[g]["M"], arr[g]["N"], arr[g]["K"] = M, N, K
    desc = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.dib_gemm_skinny_k(mode, groups, _ptr(desc), M, N, K, _ptr(At), _ptr(Bt), _ptr(Ct), _ptr(bt) if mode == 0 else None, st)
    assert rc == 0
    torch.cuda.synchronize()
    got = Ct.cpu().numpy()
    for g in range(groups):
        a64 = A[g, :, :K].astype(np.float64)
        if mode == 0:
            ref = a64 @ B[g, :, :N].astype(np.float64) + (bias[g].astype(np.float64) if g != 1 else 0.0)
        else:
            ref = a64 @ B[g, :, :K].astype(np.float64).T
        out = got[g, :, :N]
        assert np.isfinite(out).all()
        assert np.abs(out - ref).max() / (np.abs(ref).max() + 1e-6) < 2e-6, (g, np.abs(out - ref).max())
        if pad:
            assert np.isnan(got[g, :, N:]).all(), "wrote outside the matrix"


def test_skinny_k_gemm_refuses_what_it_does_not_cover():
    from dib_amd import _lib
    lib = _lib.load_library()
    t = torch.zeros(4096, device="cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (mode, M, N, K) in [(0, 64, 128, 36), (0, 64, 128, 30), (0, 64, 100, 32), (2, 64, 128, 32)]:
        assert lib.dib_gemm_skinny_k(mode, 1, _ptr(t), M, N, K, _ptr(t), _ptr(t), _ptr(t), None, st) < 0


def test_gemm_is_transpose_detecting():
    """A = I with ASYMMETRIC B (cdna guide: symmetric inputs hide a row/col swap in the C write)."""
    from dib_amd import _lib
    lib = _lib.load_library()
    dev = torch.device("cuda:0")
    n = 96
    A = torch.eye(n, dtype=torch.float32, device=dev)
    B = (torch.arange(n * n, dtype=torch.float32, device=dev).view(n, n) * 0.001).contiguous()
    C = torch.zeros((n, n), dtype=torch.float32, device=dev)
    desc = torch.zeros(256, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.dib_gemm(0, n, n, n, _ptr(A), n, _ptr(B), n, _ptr(C), n, None, None, 0, 0, _ptr(desc), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(C, B)  # exact: fp32 MFMA is an fmaf chain, products with 0/1 are exact


def test_eps_matches_oracle_and_host_ref():
    spec = SPECS["odd_shapes_tanh"]
    eng, _ = _engine(spec)
    rows = np.array([0, 5, 17, 123456, 2 ** 31 + 5, 99], dtype=np.int64)
    idx = torch.tensor(rows.astype(np.uint32).view(np.int32), dtype=torch.int32, device=eng.device)
    got = eng.eps(idx, 0, len(rows), seed=0x1234567890ABCDEF, step=77).cpu().numpy()
    ref = orc.philox_normal_all(0x1234567890ABCDEF, 77, rows.astype(np.uint32), spec.number_features,
                                spec.feature_embedding_dimension)
    assert np.abs(got - ref).max() < 1e-5
    host = eng.lib.dib_philox_normal_ref(0x1234567890ABCDEF, 77, 17, 1, 4)
    assert abs(host - ref[2, 1, 4]) < 1e-5
    # contiguous-row form
    got2 = eng.eps(None, 1000, 4, seed=3, step=1).cpu().numpy()
    ref2 = orc.philox_normal_all(3, 1, np.arange(1000, 1004), spec.number_features, spec.feature_embedding_dimension)
    assert np.abs(got2 - ref2).max() < 1e-5


@pytest.mark.parametrize("path", DISPATCH_PATHS)
@pytest.mark.parametrize("name", list(SPECS))
@pytest.mark.parametrize("B", [1, 37, 300, 2100])
def test_forward_backward_parity(name, B, path):
    """The architecture zoo x batch sizes x BOTH SIDES OF EVERY DISPATCH SWITCH (tests/_helpers.py dispatch_path) against the
    float64 oracle.  B = 2100 is beyond the row-tile regime for every layout (> 2048 rows): the fused-eligible entries then
    run the persistent fused kernels over several 256-row tiles per workgroup on the default path too."""
    if B == 2100 and (name not in FUSED_ELIGIBLE or path == "large_batch"):
        pytest.skip("B = 2100 is the large-batch case of the fused-eligible entries (default == large_batch there)")
    if path == "cluster_tiles" and B not in (37, 300):
        pytest.skip("the cluster mode of the row-tile integration kernel: 3 and 19 row tiles on 4 workgroups each")
    with dispatch_path(path):
        _forward_backward_parity(name, B)


def _forward_backward_parity(name, B):
    spec = SPECS[name]
    # (zlib.crc32, not hash(): str hashes are salted per process, and with a different parameter seed every run the odd run hit
    # a ReLU unit within round-off of 0 - one flipped unit moves a gradient column by ~1/sqrt(B) of its scale)
    eng, p = _engine(spec, seed=zlib.crc32(name.encode()) % 1000)
    F, E = spec.number_features, spec.feature_embedding_dimension
    rng = np.random.default_rng(B)
    n = B + 11
    x = rng.standard_normal((n, sum(spec.feature_dimensionalities))).astype(np.float32)
    if spec.output_dimensionality == 1:
        kind = "bce" if spec.output_activation_fn == "sigmoid" else "bce_logits"
        y = rng.integers(0, 2, (n, 1)).astype(np.float32)
    elif name == "pendulum_ragged":
        kind = "mse"
        y = rng.standard_normal((n, spec.output_dimensionality)).astype(np.float32)
    else:
        kind = "sparse_cce_logits"
        y = rng.integers(0, spec.output_dimensionality, (n, 1)).astype(np.float32)
    rows = rng.permutation(n)[:B].astype(np.int32)
    xd, yd = eng.to_device(x), eng.to_device(y)
    idx = eng.to_device(rows, dtype=torch.int32)
    beta, seed, step = 0.37, 11, 5
    eng.set_beta(beta)
    eng.train_step(xd, yd, idx, 0, B, seed, step, kind)
    torch.cuda.synchronize()

    eps = orc.philox_normal_all(seed, step, rows, F, E)
    c = orc.forward(spec, p, x[rows].astype(np.float64), eps)
    task, grads, g_u = orc.backward(spec, p, x[rows].astype(np.float64), y[rows], c, beta, kind)

    def close(got, ref, tol=2e-4):
        got = np.asarray(got, dtype=np.float64)
        return np.abs(got - ref).max() <= tol * (1.0 + np.abs(ref).max())

    enc_out = eng.enc_out(B).cpu().numpy()
    assert close(enc_out[:, :, :E], c.mu), "mu"
    assert close(enc_out[:, :, E:], c.logvar), "logvar"
    assert close(eng.u(B).cpu().numpy(), c.u), "u"
    assert close(eng.pred(B).cpu().numpy(), c.pred), "pred"
    so = eng.step_out(B).cpu().numpy()
    assert np.abs(so[:F] / B - c.kl).max() < 1e-3, "KL per feature (nats)"
    assert abs(so[F] / B - task) < 2e-4 * (1 + abs(task)), "task loss"
    assert so[F + 2] == B
    if kind != "mse":
        assert abs(so[F + 1] / B - orc.accuracy(kind, y[rows], c.pred)) < 1e-6
    assert close(eng.g_u(B).cpu().numpy(), g_u, 3e-4), "g_u"
    gflat = eng.get_flat_grads()
    gref = params_to_flat(eng.blocks, grads, eng.params.numel()).astype(np.float64)
    for b in eng.blocks:
        sl = slice(b["offset"], b["offset"] + b["rows"] * b["cols"])
        ref = gref[sl]
        err = np.abs(gflat[sl] - ref).max()
        assert err <= 3e-4 * (np.abs(ref).max() + 1e-3), (b, err, np.abs(ref).max())


@pytest.mark.parametrize("path", DISPATCH_PATHS)
@pytest.mark.parametrize("name", ["tabular8_default", "no_posenc_leaky"])   # fused kernels / general (elementwise) path
def test_late_annealing_regime_small_sigma_large_mu(name, path):
    with dispatch_path(path):
        _late_annealing_regime(name)


def _late_annealing_regime(name):
    """the backward recovers eps * sigma as u - mu instead of regenerating eps.  That difference cancels once
    sigma << |mu| - the late-annealing state of an informative feature (logvar ~ -16, |mu| ~ 4: sigma = 3e-4, ulp(u) = 5e-7,
    so eps * sigma carries ~1e-3 relative error per element).  What it feeds is only the NOISE term of d loss / d logvar,
    g_u * eps * sigma / 2, itself ~1e-4 of g_u: this test pins the regime - every gradient block at the usual tolerance, and
    the logvar half of the last encoder layer's gradients (the blocks the term enters first) separately and tighter than the
    error the cancellation could cause if it mattered."""
    spec = SPECS[name]
    eng, p = _engine(spec, seed=5)
    F, E = spec.number_features, spec.feature_embedding_dimension
    last = len(spec.feature_encoder_architecture)
    for f in range(F):                       # shrink the last layer and put (mu, logvar) at (+-4, -16) through its bias
        p.enc_W[f][last] *= 0.05
        p.enc_b[f][last][:E] = 4.0 * np.where(np.arange(E) % 2 == 0, 1.0, -1.0)
        p.enc_b[f][last][E:] = -16.0
    eng.set_flat_params(params_to_flat(eng.blocks, p, eng.params.numel()))
    p = flat_to_params(eng.blocks, eng.get_flat_params(), spec)
    B = 300
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, sum(spec.feature_dimensionalities))).astype(np.float32)
    kind = "bce_logits" if spec.output_dimensionality == 1 else "sparse_cce_logits"
    y = rng.integers(0, max(2, spec.output_dimensionality), (B, 1)).astype(np.float32)
    beta, seed, step = 1e-4, 3, 9            # small beta: the KL term does not drown the noise term
    eng.set_beta(beta)
    eng.train_step(eng.to_device(x), eng.to_device(y), None, 0, B, seed, step, kind)
    torch.cuda.synchronize()
    eps = orc.philox_normal_all(seed, step, np.arange(B), F, E)
    c = orc.forward(spec, p, x.astype(np.float64), eps)
    assert np.abs(c.logvar + 16).max() < 1.5 and np.abs(np.abs(c.mu) - 4).max() < 1.5
    task, grads, g_u = orc.backward(spec, p, x.astype(np.float64), y, c, beta, kind)
    gflat = eng.get_flat_grads()
    gref = params_to_flat(eng.blocks, grads, eng.params.numel()).astype(np.float64)
    worst = 0.0
    for b in eng.blocks:
        sl = slice(b["offset"], b["offset"] + b["rows"] * b["cols"])
        ref, got = gref[sl], gflat[sl]
        assert np.abs(got - ref).max() <= 3e-4 * (np.abs(ref).max() + 1e-3), (b, np.abs(got - ref).max(), np.abs(ref).max())
        if b["net"] == 0 and b["layer"] == last:   # logvar columns: [rows, E:2E] of the kernel, [E:2E] of the bias
            r2, g2 = ref.reshape(b["rows"], b["cols"])[:, E:], got.reshape(b["rows"], b["cols"])[:, E:]
            rel = np.abs(g2 - r2).max() / (np.abs(r2).max() + 1e-30)
            worst = max(worst, rel)
            assert rel < 2e-4, (b, rel)
    print("worst relative error of the logvar-half gradients:", worst)


def test_split_batch_wgrad_and_dp_equivalence():
    """B large enough to use split-batch wgrad partials; and two half-batches with
    inv_global_batch = 1/B reproduce the full-batch gradient (the data-parallel contract)."""
    spec = SPECS["tabular8_default"]
    eng, p = _engine(spec, 3)
    B = 8192
    assert eng.lib.dib_layout_wgrad_splits(eng.layout, B) > 1
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B, 8)).astype(np.float32)
    y = (x[:, 0] > 0).astype(np.float32)[:, None]
    xd, yd = eng.to_device(x), eng.to_device(y)
    eng.set_beta(0.05)
    eng.train_step(xd, yd, None, 0, B, 1, 2, "bce_logits")
    g_full = eng.get_flat_grads().astype(np.float64)
    kl_full = eng.step_out(B).cpu().numpy()[:8].copy()
    eps = orc.philox_normal_all(1, 2, np.arange(B), 8, 32)
    c = orc.forward(spec, p, x.astype(np.float64), eps)
    _, grads, _ = orc.backward(spec, p, x.astype(np.float64), y, c, 0.05, "bce_logits")
    gref = params_to_flat(eng.blocks, grads, eng.params.numel()).astype(np.float64)
    assert np.abs(g_full - gref).max() <= 3e-4 * np.abs(gref).max()
    assert np.abs(kl_full / B - c.kl).max() < 1e-3
    h = B // 2
    eng.train_step(xd, yd, None, 0, h, 1, 2, "bce_logits", inv_global_batch=1.0 / B)
    g0 = eng.get_flat_grads().astype(np.float64)
    eng.train_step(xd, yd, None, h, h, 1, 2, "bce_logits", inv_global_batch=1.0 / B)
    g1 = eng.get_flat_grads().astype(np.float64)
    assert np.abs((g0 + g1) - g_full).max() <= 1e-5 * np.abs(g_full).max() + 1e-9


def test_adam_matches_keras_form():
    spec = SPECS["odd_shapes_tanh"]
    eng, p = _engine(spec, 5)
    rng = np.random.default_rng(1)
    st = orc.adam_init(p)
    eng.set_lr(3e-4)
    for it in range(5):
        g = rng.standard_normal(eng.params.numel()).astype(np.float32) * (10.0 ** rng.integers(-4, 1))
        eng.grads.copy_(torch.from_numpy(g))
        eng.adam_step()
        orc.adam_keras_step(p, flat_to_params(eng.blocks, g, spec), st, lr=3e-4)
    got = flat_to_params(eng.blocks, eng.get_flat_params(), spec)
    for a, b in zip(got.tensors(), p.tensors()):
        assert np.abs(a - b).max() < 2e-6
    assert int(eng.t_dev.item()) == 5


def test_encode_deterministic_and_bhattacharyya():
    spec = SPECS["pendulum_ragged"]
    eng, p = _engine(spec, 9)
    rng = np.random.default_rng(2)
    E = spec.feature_embedding_dimension
    for f, d in enumerate(spec.feature_dimensionalities):
        xf = rng.standard_normal((50, d)).astype(np.float32)
        got = eng.encode_feature(f, xf).cpu().numpy()
        ref = orc.encode_feature(spec, p, f, xf.astype(np.float64))
        assert np.abs(got - ref).max() < 2e-4 * (1 + np.abs(ref).max())
        bh = eng.bhattacharyya(got[:, :E], got[:, E:], got[:20, :E], got[:20, E:]).cpu().numpy()
        bref = orc.bhattacharyya_dist_mat(got[:, :E], got[:, E:], got[:20, :E], got[:20, E:])
        assert np.abs(bh - bref).max() < 1e-3 * (1 + np.abs(bref).max())


@pytest.mark.parametrize("path", DISPATCH_PATHS)
def test_fit_trajectory_matches_oracle_fit(path):
    """BASELINE metric 2: per-epoch KL{f} within 1e-3 nats of the oracle over a full fit()
    (same init, same batch order, same counter-based eps), incl. validation and accuracy - on the row-tile kernels (default at
    these batch sizes), on the fused <32,32,8> forward + grouped-GEMM backward (large_batch) and on the grouped-GEMM path."""
    with dispatch_path(path):
        _fit_trajectory()


def _fit_trajectory():
    import dib_amd
    spec = orc.DIBSpec([1, 1, 1, 1], [32, 32], [64, 64], 1, feature_embedding_dimension=8)
    x, y = orc.boolean_circuit_truth_table([0, 1, 2, 3, [0, 2, 0], [2, 4, 3], [0, 5, 1]], 4)  # SI circuit (c)
    x = np.tile(x, (8, 1)).astype(np.float32)
    y = np.tile(y, 8).astype(np.float32)
    model = dib_amd.DistributedIBNet(**spec_kwargs(spec), noise_seed=4, shuffle_seed=6, init_seed=2)
    opt = dib_amd.optimizers.get("adam")
    opt.learning_rate = 3e-3
    model.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-3, 1.0, 2, 6)
    p = flat_to_params(model.param_blocks(), model.get_flat_weights(), spec)
    epochs, bs = 8, 48  # 128 rows -> 2 full batches + a partial one
    hist = model.fit(x, y, epochs=epochs, shuffle=True, batch_size=bs, callbacks=[cb], verbose=False,
                     validation_data=(x[:40], y[:40]))
    ref = orc.fit(spec, p, x, y, epochs=epochs, batch_size=bs, loss_kind="bce_logits",
                  beta_fn=lambda e: orc.beta_schedule(e, 1e-3, 1.0, 2, 6), lr=3e-3, shuffle=True,
                  validation_data=(x[:40], y[:40]), noise_seed=4, shuffle_seed=6, metrics=["accuracy"])
    assert set(ref) == set(hist.history)
    for k in ref:
        got, want = np.array(hist.history[k]), np.array(ref[k])
        tol = 1e-3 if "KL" in k else 2e-3
        assert np.abs(got - want).max() < tol * (1 + np.abs(want).max()), (k, got, want)


def test_north_star_shape_properties():
    """BASELINE config 3 at full size (F=64, B=65536): size-independent properties -
    KL >= 0 and finite, deterministic replay bit-exact, u - mu = exp(lv/2)*eps, grads finite."""
    spec = orc.DIBSpec([1] * 64, [128, 128], [256, 256], 1)
    from dib_amd.engine import HipEngine
    eng = HipEngine(**spec_kwargs(spec), init_seed=1)
    B = 65536
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((B, 64), generator=g)
    y = (x[:, 0] + 0.5 * x[:, 1] * x[:, 2] > 0).float()[:, None]
    xd, yd = eng.to_device(x), eng.to_device(y)
    eng.set_beta(1e-2)
    eng.train_step(xd, yd, None, 0, B, 0, 0, "bce_logits")
    g1 = eng.grads.clone()
    so1 = eng.step_out(B).clone()
    u = eng.u(B).view(B, 64, 32)[:256].clone()
    eo = eng.enc_out(B)[:256].clone()
    eng.train_step(xd, yd, None, 0, B, 0, 0, "bce_logits")
    assert torch.equal(g1, eng.grads), "replay must be bit-exact (fixed-order reductions)"
    assert torch.equal(so1, eng.step_out(B))
    assert torch.isfinite(g1).all() and torch.isfinite(so1).all()
    assert (so1[:64] >= 0).all()
    eps = eng.eps(None, 0, 256, 0, 0)
    resid = u - (eo[:, :, :32] + torch.exp(0.5 * eo[:, :, 32:]) * eps)
    assert resid.abs().max() < 1e-5
    # oracle spot check on the first 64 rows' predictions
    p = flat_to_params(eng.blocks, eng.get_flat_params(), spec)
    c = orc.forward(spec, p, x[:64].numpy().astype(np.float64), orc.philox_normal_all(0, 0, np.arange(64), 64, 32))
    assert np.abs(eng.pred(B)[:64].cpu().numpy() - c.pred).max() < 2e-4 * (1 + np.abs(c.pred).max())


def test_mi_sandwich_bounds_match_oracle():
    """dib_mi_sandwich_rows (float64 log-sum-exp on device) vs the literal restatement of utils.py:36-62."""
    import dib_amd
    spec = SPECS["pendulum_ragged"]
    eng, p = _engine(spec, 4)
    rng = np.random.default_rng(5)
    E = spec.feature_embedding_dimension
    for f, d in enumerate(spec.feature_dimensionalities):
        xf = rng.standard_normal((300, d)).astype(np.float32)
        enc = eng.encode_feature(f, xf)
        lo, up = eng.mi_sandwich_bounds(enc, seed=9, step=2, feature=f)
        e = enc.cpu().numpy().astype(np.float64)
        u = orc.mi_sandwich_sample_u(e[:, :E], e[:, E:], 9, 2, f)
        rlo, rup = orc.mi_sandwich_bounds_batch(e[:, :E], e[:, E:], u)
        assert abs(lo - rlo) < 1e-4 * (1 + abs(rlo)) and abs(up - rup) < 1e-4 * (1 + abs(rup)), (lo, rlo, up, rup)
        assert lo <= up + 1e-9
    # well separated encodings: the reference underflows to inf, the LSE version saturates at log N
    mu = (np.arange(64, dtype=np.float32)[:, None] * 100.0) * np.ones((1, E), dtype=np.float32)
    enc = torch.tensor(np.concatenate([mu, np.zeros_like(mu)], 1), device=eng.device)
    lo, up = eng.mi_sandwich_bounds(enc, 0, 0, 0)
    assert abs(lo - np.log(64)) < 1e-9 and up > 1e3


def test_info_per_feature_callback_and_train_script(tmp_path):
    """The reference-style script path end to end on the GPU: dib_amd.train main() + InfoPerFeatureCallback."""
    import dib_amd
    from dib_amd import train
    hist = train.main(["--dataset", "boolean_circuit", "--number_pretraining_epochs", "2", "--number_annealing_epochs", "3",
                       "--batch_size", "128", "--artifact_outdir", str(tmp_path), "--save_compression_matrices_frequency", "4",
                       "--feature_encoder_architecture", "32", "32", "--integration_network_architecture", "64",
                       "--feature_embedding_dimension", "8"])
    assert len(hist.history["loss"]) == 5 and np.isfinite(hist.history["loss"]).all()
    import os
    assert os.path.exists(os.path.join(str(tmp_path), "distributed_info_plane.png"))
    assert any(f.startswith("feature_0_log10beta") for f in os.listdir(str(tmp_path)))
    d = dib_amd.data.fetch_boolean_circuit()
    model = dib_amd.DistributedIBNet(d["feature_dimensionalities"], [32, 32], [64], 1, feature_embedding_dimension=8)
    model.compile(optimizer="adam", loss=d["loss"], metrics=d["metrics"])
    cb = dib_amd.InfoPerFeatureCallback(1, d["x_valid"], info_bound_batch_size=256, info_bound_number_batches=2)
    model.fit(d["x_train"], d["y_train"], epochs=2, batch_size=256, callbacks=[cb], verbose=False)
    b = np.array(cb.bounds)
    assert b.shape == (20, 2) and np.isfinite(b).all() and (b[:, 0] <= b[:, 1] + 1e-6).all()
    assert (b[:, 1] <= np.log(2) + 0.2).all()  # a binary input carries at most 1 bit


def test_ib_flag_one_wide_feature_on_the_boolean_circuit(tmp_path):
    """reference train.py:111-113 `--ib`: "just treat everything as one feature" - the Boolean circuit's ten +-1 inputs become ONE
    feature of width 10, positionally encoded to a 50-wide encoder input (beyond the fused encoder kernels' 16 columns: the
    grouped-GEMM path with ONE group), train.py's default architecture otherwise.  (1) forward, the single KL, the task loss and
    every gradient block on a batch of circuit rows against the oracle; (2) `python -m dib_amd.train --ib True` end to end: the
    History has KL0 and nothing else, the net learns the circuit during pre-training and pays for it with one channel."""
    import dib_amd
    from dib_amd import train
    d = dib_amd.data.fetch_boolean_circuit()
    spec = orc.DIBSpec([int(np.sum(d["feature_dimensionalities"]))], [128, 128], [256, 256], 1, feature_embedding_dimension=32)
    assert spec.feature_dimensionalities == [10]
    eng, p = _engine(spec, seed=3)
    assert eng.F == 1 and eng.sum_d == 10
    B, beta, seed, step = 128, 0.05, 9, 2
    x, y = np.asarray(d["x_train"], dtype=np.float32), np.asarray(d["y_train"], dtype=np.float32).reshape(-1, 1)
    rows = np.random.default_rng(1).permutation(len(x))[:B].astype(np.int32)
    eng.set_beta(beta)
    eng.train_step(eng.to_device(x), eng.to_device(y), eng.to_device(rows, dtype=torch.int32), 0, B, seed, step, "bce_logits")
    torch.cuda.synchronize()
    c = orc.forward(spec, p, x[rows].astype(np.float64), orc.philox_normal_all(seed, step, rows, 1, 32))
    task, grads, g_u = orc.backward(spec, p, x[rows].astype(np.float64), y[rows], c, beta, "bce_logits")
    close = lambda got, ref, tol=2e-4: np.abs(np.asarray(got, dtype=np.float64) - ref).max() <= tol * (1.0 + np.abs(ref).max())
    enc_out = eng.enc_out(B).cpu().numpy()
    assert close(enc_out[:, :, :32], c.mu) and close(enc_out[:, :, 32:], c.logvar) and close(eng.u(B).cpu().numpy(), c.u)
    assert close(eng.pred(B).cpu().numpy(), c.pred)
    so = eng.step_out(B).cpu().numpy()
    assert c.kl.shape == (1,) and abs(so[0] / B - c.kl[0]) < 1e-3, "the one KL (nats)"
    assert abs(so[1] / B - task) < 2e-4 * (1 + abs(task)) and so[3] == B
    assert close(eng.g_u(B).cpu().numpy(), g_u, 3e-4)
    gflat = eng.get_flat_grads()
    gref = params_to_flat(eng.blocks, grads, eng.params.numel()).astype(np.float64)
    assert [b["rows"] for b in eng.blocks if b["net"] == 0 and b["layer"] == 0 and b["what"] == 0] == [50]
    for b in eng.blocks:
        sl = slice(b["offset"], b["offset"] + b["rows"] * b["cols"])
        err = np.abs(gflat[sl] - gref[sl]).max()
        assert err <= 3e-4 * (np.abs(gref[sl]).max() + 1e-3), (b, err)
    hist = train.main(["--dataset", "boolean_circuit", "--ib", "True", "--number_pretraining_epochs", "300",
                       "--number_annealing_epochs", "200", "--beta_start", "1e-5", "--beta_end", "5", "--learning_rate", "1e-3",
                       "--batch_size", "128", "--artifact_outdir", str(tmp_path)]).history
    assert sorted(k for k in hist if "KL" in k) == ["KL0", "val_KL0"] and len(hist["loss"]) == 500
    assert hist["accuracy"][299] >= 0.98, hist["accuracy"][299]            # the truth table is learnt through one channel
    assert hist["KL0"][299] > 0.758 * np.log(2) - 0.1                        # ... which must carry about H(Y) = 0.758 bits at least
    assert hist["KL0"][-1] < 0.02 and abs(hist["loss"][-1] / np.log(2) - 0.758) < 0.03   # and is closed by the end of the ramp
    assert os.path.exists(os.path.join(str(tmp_path), "distributed_info_plane.png"))


@pytest.mark.parametrize("kind", ["l2sq", "l2", "l1", "linf", "cosine"])
def test_infonce_loss_and_grads_match_oracle(kind):
    """dib_infonce_fwd_bwd vs the numpy restatement of train.py:203-215 / utils.py:131-175 (grads by central differences)."""
    eng, _ = _engine(SPECS["no_hidden"])
    rng = np.random.default_rng(3)
    B, D = 12, 8
    a = rng.standard_normal((B, D)).astype(np.float32)
    b = (a + 0.7 * rng.standard_normal((B, D))).astype(np.float32)
    loss, gx, gy = eng.infonce(eng.to_device(a), eng.to_device(b), kind, 0.7)
    ref = orc.infonce_loss(a, b, kind, 0.7)
    assert abs(float(loss.item()) - ref) < 2e-5 * (1 + abs(ref))
    g1, g2 = orc.infonce_grads_numeric(a, b, kind, 0.7)
    assert np.abs(gx.cpu().numpy() - g1).max() < 2e-4 * (1 + np.abs(g1).max())
    assert np.abs(gy.cpu().numpy() - g2).max() < 2e-4 * (1 + np.abs(g2).max())


@pytest.mark.parametrize("D", [8, 64])
@pytest.mark.parametrize("B", [12, 257, 2048])
@pytest.mark.parametrize("kind", ["l2sq", "l2", "l1", "linf", "cosine"])
def test_infonce_at_working_batch_sizes(kind, B, D):
    """The sizes the path runs at: train.py:34 default batch 128, the chaos notebook 2048 (Chaos_experiments.ipynb:771-821),
    shared space 64 (train.py:60).  The kernels launch dim3(batch, 2) x 256 threads: B > 256 is where their strided loops
    over a similarity row iterate more than once, B = 257 the first ragged trip.  Checker: float64 autograd
    (oracle/dib_torch_cpu.infonce_loss_and_grads, pinned on the numpy restatement on the CPU side).
    The temperature of the distance similarities is set from the data so that positives beat negatives by ~4 nats: a trained
    pair of encoders sits there; at a saturated softmax (l2sq of 64-dimensional N(0,1) embeddings at T = 1: gap 100 nats) the
    gradients are differences of numbers that agree to fp32 round-off (~1e-7 of a 1/B weight) and there is nothing to compare.
    Tolerances: loss 2e-5 relative; gradients 2e-4 of the largest gradient entry (fp32 similarity rows of up to 2048 terms);
    linf 1e-3: its derivative goes to the arg-max coordinate only, and of the B^2 D = 2.7e8 coordinate differences at
    B = 2048, D = 64 a few dozen pairs have their two largest |x_q - y_q| within fp32 round-off of each other - float32 and
    float64 then route that pair's weight to different coordinates (the arg-max is discontinuous)."""
    import dib_torch_cpu as tc
    eng, _ = _engine(SPECS["no_hidden"])
    rng = np.random.default_rng(1000 * B + D)
    a = rng.standard_normal((B, D)).astype(np.float32)
    b = (a + 0.7 * rng.standard_normal((B, D))).astype(np.float32)
    temp = 0.7
    if kind != "cosine":
        S1 = orc.scaled_similarity(a[:256], b[:256], kind, 1.0)
        off = S1[~np.eye(len(S1), dtype=bool)]
        temp = float(max(np.median(np.diag(S1)) - np.median(off), 1.0) / 4.0)
    loss, gx, gy = eng.infonce(eng.to_device(a), eng.to_device(b), kind, temp)
    ref, g1, g2 = tc.infonce_loss_and_grads(a, b, kind, temp)
    assert abs(float(loss.item()) - ref) < 2e-5 * (1 + abs(ref)), (float(loss.item()), ref)
    tol = 1e-3 if kind == "linf" else 2e-4
    ex, ey = np.abs(gx.cpu().numpy() - g1).max(), np.abs(gy.cpu().numpy() - g2).max()
    assert ex < tol * np.abs(g1).max() + 1e-9, ("d/dx", ex, np.abs(g1).max(), temp)
    assert ey < tol * np.abs(g2).max() + 1e-9, ("d/dy", ey, np.abs(g2).max(), temp)


@pytest.mark.parametrize("B,D", [(33, 256), (70, 5), (1, 7), (150, 100), (70, 160)])   # D = 100 / 160: the 2- / 3-accumulator gradient kernels
@pytest.mark.parametrize("kind", ["l2sq", "l2", "l1", "linf", "cosine"])
def test_infonce_edge_shapes(kind, B, D):
    """ragged 32 x 32 pair tiles, the widest supported embedding (256: the similarity kernel's LDS tiles need the raised
    dynamic-LDS limit), a width that does not divide 256, a single-row batch (loss 0, gradients 0)."""
    import dib_torch_cpu as tc
    eng, _ = _engine(SPECS["no_hidden"])
    rng = np.random.default_rng(B + D)
    a = (rng.standard_normal((B, D)) / np.sqrt(D)).astype(np.float32)
    b = (a + 0.5 * rng.standard_normal((B, D)) / np.sqrt(D)).astype(np.float32)
    loss, gx, gy = eng.infonce(eng.to_device(a), eng.to_device(b), kind, 0.5)
    ref, g1, g2 = tc.infonce_loss_and_grads(a, b, kind, 0.5)
    tol = 1e-3 if kind == "linf" else 2e-4
    assert abs(float(loss.item()) - ref) < 2e-5 * (1 + abs(ref)), (float(loss.item()), ref)
    assert np.abs(gx.cpu().numpy() - g1).max() < tol * np.abs(g1).max() + 1e-7
    assert np.abs(gy.cpu().numpy() - g2).max() < tol * np.abs(g2).max() + 1e-7


@pytest.mark.parametrize("B,D", [(128, 64), (100, 64), (64, 20), (33, 6)])
@pytest.mark.parametrize("kind", ["l2sq", "l2", "cosine"])
def test_infonce_one_launch_path(kind, B, D):
    """dib_infonce_small_kernel (batch <= 128, dim <= 64: the reference's default batch and shared space, train.py:34,60) - the
    whole symmetric InfoNCE in one launch - against the float64 autograd checker at the tolerances of the three-launch path,
    against that path itself (dib_set_tuning("infonce_one_launch", 0)), and the loss-only call of the validation pass."""
    import dib_torch_cpu as tc
    from dib_amd import _lib
    eng, _ = _engine(SPECS["no_hidden"])
    rng = np.random.default_rng(B * 7 + D)
    a = rng.standard_normal((B, D)).astype(np.float32)
    b = (a + 0.7 * rng.standard_normal((B, D))).astype(np.float32)
    temp = 0.7
    if kind != "cosine":
        S1 = orc.scaled_similarity(a, b, kind, 1.0)
        off = S1[~np.eye(len(S1), dtype=bool)]
        temp = float(max(np.median(np.diag(S1)) - np.median(off), 1.0) / 4.0)
    ad, bd = eng.to_device(a), eng.to_device(b)
    assert _lib.get_tuning("infonce_one_launch") == 1
    n0 = eng.lib.dib_launch_count()
    loss, gx, gy = eng.infonce(ad, bd, kind, temp)
    assert eng.lib.dib_launch_count() - n0 == 1
    loss_only, _, _ = eng.infonce(ad, bd, kind, temp, want_grads=False)
    try:
        _lib.set_tuning("infonce_one_launch", 0)
        loss3, gx3, gy3 = eng.infonce(ad, bd, kind, temp)
    finally:
        _lib.set_tuning("infonce_one_launch", 1)
    ref, g1, g2 = tc.infonce_loss_and_grads(a, b, kind, temp)
    assert abs(float(loss.item()) - ref) < 2e-5 * (1 + abs(ref)), (float(loss.item()), ref)
    assert float(loss_only.item()) == float(loss.item())
    assert abs(float(loss.item()) - float(loss3.item())) < 1e-5 * (1 + abs(ref))
    for got, got3, want in ((gx, gx3, g1), (gy, gy3, g2)):
        assert np.abs(got.cpu().numpy() - want).max() < 2e-4 * np.abs(want).max() + 1e-9
        assert (got - got3).abs().max().item() < 2e-5 * np.abs(want).max() + 1e-9


def test_mi_sandwich_bounds_at_the_reference_evaluation_size():
    """utils.py:10-11 evaluates the bounds on batches of 1024 (evaluation_batch_size); embedding dimension 32 (train.py:55).
    Device float64 log-sum-exp kernel vs the literal restatement of utils.py:36-62 on the device's own samples."""
    spec = orc.DIBSpec([1, 2], [16], [8], 1, feature_embedding_dimension=32)
    eng, p = _engine(spec, 6)
    rng = np.random.default_rng(7)
    E, N = 32, 1024
    for f, d in enumerate(spec.feature_dimensionalities):
        xf = rng.standard_normal((N, d)).astype(np.float32)
        enc = eng.encode_feature(f, xf)
        lo, up = eng.mi_sandwich_bounds(enc, seed=3, step=5, feature=f)
        e = enc.cpu().numpy().astype(np.float64)
        u = orc.mi_sandwich_sample_u(e[:, :E], e[:, E:], 3, 5, f)
        rlo, rup = orc.mi_sandwich_bounds_batch(e[:, :E], e[:, E:], u)
        assert np.isfinite([rlo, rup]).all()
        assert abs(lo - rlo) < 1e-4 * (1 + abs(rlo)) and abs(up - rup) < 1e-4 * (1 + abs(rup)), (lo, rlo, up, rup)
        assert lo <= up + 1e-9 and lo <= np.log(N) + 1e-9


@pytest.mark.parametrize("n", [37, 2048])
def test_dense_stack_matches_numpy(n):
    """The Y-encoder MLP (DenseStack over dib_gemm_grouped): forward, backward and Keras-Adam vs numpy; n = 2048 (the chaos
    notebook's batch) runs the weight gradients in 32 batch slabs + fixed-order reduce."""
    from dib_amd.dense import DenseStack
    eng, _ = _engine(SPECS["no_hidden"])
    ds = DenseStack(eng, 6, [20, 12], 5, "relu", True, 3, seed=4)
    rng = np.random.default_rng(0)
    y = rng.standard_normal((n, 6)).astype(np.float32)
    out = ds.forward(eng.to_device(y)).cpu().numpy()
    Ws = [ds.kernel(l).cpu().numpy().astype(np.float64) for l in range(3)]
    bs = [rng.standard_normal(ds.dims[l][1]) * 0.1 for l in range(3)]
    for l in range(3):
        ds.bias(l).copy_(torch.tensor(bs[l], dtype=torch.float32))
    out = ds.forward(eng.to_device(y)).cpu().numpy()
    h = orc.positional_encoding(y.astype(np.float64), [2, 4])
    hs = [h]
    for l in range(3):
        z = hs[-1] @ Ws[l] + np.float32(bs[l]).astype(np.float64)
        hs.append(np.maximum(z, 0) if l < 2 else z)
    assert np.abs(out - hs[-1]).max() < 1e-4
    g = rng.standard_normal(out.shape).astype(np.float32)
    ds.backward(eng.to_device(g))
    gg = g.astype(np.float64)
    for l in reversed(range(3)):
        gw = hs[l].T @ gg
        gb = gg.sum(0)
        i, o = ds.dims[l]
        got_w = ds.grads[ds.w_off[l]: ds.w_off[l] + i * o].view(i, o).cpu().numpy()
        got_b = ds.grads[ds.b_off[l]: ds.b_off[l] + o].cpu().numpy()
        assert np.abs(got_w - gw).max() < 2e-4 * (1 + np.abs(gw).max()) and np.abs(got_b - gb).max() < 2e-4 * (1 + np.abs(gb).max())
        if l > 0:
            gg = (gg @ Ws[l].T) * (hs[l] > 0)


@pytest.mark.parametrize("n,units,out,act,pe,gather", [(128, [128, 128], 64, "relu", True, True),
                                                       (37, [128, 128], 64, "leaky_relu", True, False),
                                                       (1000, [64, 32, 48], 16, "relu", False, True),
                                                       (1, [32], 32, None, True, True)])
def test_dense_stack_row_tile_kernels(n, units, out, act, pe, gather):
    """The output encoder at the custom loop's batch sizes (train.py:184-192 at B = 128 .. 1024): dib_mlp_small_fwd / _bwd - gather
    + PositionalEncoding + every layer in one launch, the dgrad chain in another (csrc/dib_small.h) - against float64 numpy, and
    against the layer-by-layer grouped-GEMM launches of the same DenseStack (dib_set_tuning("small_batch", 0))."""
    from dib_amd import _lib
    from dib_amd.dense import DenseStack
    eng, _ = _engine(SPECS["no_hidden"])
    nf = 5 if pe else 1
    ds = DenseStack(eng, 6, units, out, act, pe, nf, seed=4)
    L = len(units) + 1
    rng = np.random.default_rng(n)
    ytab = rng.standard_normal((n + 9, 6)).astype(np.float32)
    rows = rng.permutation(n + 9)[:n].astype(np.int32) if gather else None
    for l in range(L):
        ds.bias(l).copy_(torch.tensor(rng.standard_normal(ds.dims[l][1]) * 0.1, dtype=torch.float32))
    yd = eng.to_device(ytab if gather else ytab[:n])
    rd = eng.to_device(rows, dtype=torch.int32) if gather else None
    assert ds.lib.dib_mlp_small_supported(ctypes.byref(ds._desc), n) == 1
    g = rng.standard_normal((n, out)).astype(np.float32)
    res = {}
    try:
        for small in (1, 0):
            _lib.set_tuning("small_batch", small)
            o = ds.forward(yd, rows=rd)
            assert ds._last["small"] == bool(small)
            o = o.clone()
            ds.backward(eng.to_device(g))
            torch.cuda.synchronize()
            res[small] = (o.cpu().numpy(), ds.grads.clone().cpu().numpy())
    finally:
        _lib.set_tuning("small_batch", 1)
    # float64 numpy
    yb = (ytab[rows] if gather else ytab[:n]).astype(np.float64)
    h = orc.positional_encoding(yb, [2 ** k for k in range(1, nf)]) if pe else yb
    slope = {"relu": 0.0, "leaky_relu": 0.2, None: 1.0}[act]
    Ws = [ds.kernel(l).cpu().numpy().astype(np.float64) for l in range(L)]
    bs = [ds.bias(l).cpu().numpy().astype(np.float64) for l in range(L)]
    hs = [h]
    for l in range(L):
        z = hs[-1] @ Ws[l] + bs[l]
        hs.append(np.where(z > 0, z, slope * z) if l < L - 1 else z)
    ref_g = np.zeros(ds.n_params)
    gg = g.astype(np.float64)
    for l in reversed(range(L)):
        i, o_ = ds.dims[l]
        ref_g[ds.w_off[l]: ds.w_off[l] + i * o_] = (hs[l].T @ gg).reshape(-1)
        ref_g[ds.b_off[l]: ds.b_off[l] + o_] = gg.sum(0)
        if l > 0:
            gg = (gg @ Ws[l].T) * np.where(hs[l] > 0, 1.0, slope)
    for small in (1, 0):
        o, gr = res[small]
        assert np.abs(o - hs[-1]).max() < 2e-5 * (1 + np.abs(hs[-1]).max()), small
        assert np.abs(gr - ref_g).max() < 2e-4 * (1 + np.abs(ref_g).max()), small
    assert np.abs(res[1][0] - res[0][0]).max() < 2e-5 * (1 + np.abs(res[0][0]).max())


@pytest.mark.parametrize("arch,B", [("pendulum", 128), ("pendulum", 100), ("plain", 1000), ("no_row_tiles", 128)])
def test_companion_grids_equal_the_separate_launches(arch, B):
    """include/dib_hip.h dib_integration_fwd_and_mlp_fwd / dib_backward_and_mlp_bwd: the custom loop's two networks in one grid
    each (train.py:203-219) against the separate entry points - the same workgroup code on the same tiles, so every output,
    stash and gradient must be BIT-identical.  "no_row_tiles": an X model whose integration network has no row-tile path
    (tanh) - the output encoder's pass is then launched on its own by the same entry point."""
    from dib_amd.dense import DenseStack
    from dib_amd.engine import HipEngine
    if arch == "no_row_tiles":
        spec = orc.DIBSpec([2, 1], [32, 32], [64], 64, feature_embedding_dimension=16, activation_fn="tanh")
    else:
        spec = _SMALL_ARCHS[arch][0]
    eng = HipEngine(**spec_kwargs(spec), init_seed=3)
    eng.set_beta(0.05)
    D = spec.output_dimensionality
    ds = DenseStack(eng, 6, [128, 128], D, "relu", True, 5, seed=9)
    rng = np.random.default_rng(B)
    nin = sum(spec.feature_dimensionalities)
    xd = eng.to_device(rng.standard_normal((B + 5, nin)).astype(np.float32))
    yd = eng.to_device(rng.standard_normal((B + 5, 6)).astype(np.float32))
    idx = eng.to_device(rng.permutation(B + 5)[:B].astype(np.int32), dtype=torch.int32)
    gp = eng.to_device(rng.standard_normal((B, D)).astype(np.float32) / B)
    gy = eng.to_device(rng.standard_normal((B, D)).astype(np.float32) / B)
    recs = []
    for paired in (False, True):
        eng.grads.zero_(); ds.grads.zero_()
        comp = ds.companion_forward(yd, rows=idx) if paired else None
        assert not paired or comp is not None
        eng.forward(xd, idx, 0, B, 7, 3, companion=comp)
        emb_y = (ds.companion_output() if paired else ds.forward(yd, rows=idx)).clone()
        pred = eng.pred(B).clone()
        compb = ds.companion_backward(gy) if paired else None
        eng.backward_from_pred_grad(gp, idx, 0, B, 7, 3, inv_global_batch=1.0 / B, companion=compb)
        ds.backward(gy, dgrad_done=paired)
        torch.cuda.synchronize()
        recs.append(dict(emb_y=emb_y, pred=pred, xg=eng.grads.clone(), yg=ds.grads.clone(), gu=eng.g_u(B).clone()))
    a, b = recs
    # the output encoder rides in a CLUSTER-mode grid whenever the X model's integration network clusters (round 6): it then runs on
    # the slice primitives (one workgroup per tile, csrc/dib_small.h) - the separate launch's values in another fp32 summation order
    from dib_amd import _lib
    x_hidden = sum(i * o for i, o in zip([spec.number_features * spec.feature_embedding_dimension] + list(spec.integration_network_architecture),
                                         spec.integration_network_architecture))
    clustered = (arch != "no_row_tiles" and _lib.get_tuning("int_cluster") > 1 and x_hidden >= _lib.get_tuning("int_cluster_min_weights")
                 and (B + 15) // 16 * _lib.get_tuning("int_cluster") <= _lib.get_tuning("int_cluster_wgs"))
    for k in a:
        if clustered and k in ("emb_y", "yg"):
            ref, got = a[k].double(), b[k].double()
            if k == "emb_y":
                assert (got - ref).abs().max() <= 3e-5 * (1e-6 + ref.abs().max()) + 1e-6, k
            else:
                assert (got - ref).norm() <= 5e-3 * ref.norm(), k
        else:
            assert torch.equal(a[k], b[k]), k
    assert torch.isfinite(b["yg"]).all() and b["yg"].abs().max() > 0 and b["xg"].abs().max() > 0


@pytest.mark.parametrize("B", [37, 128])
def test_companion_grid_with_a_clustered_output_encoder(B):
    """The paired grid in cluster mode (csrc/dib_small.h dib_small_integration_pair_cluster_kernel): each of the two networks picks
    its own workgroups per row tile.  With a [256, 256] output encoder (>= "int_cluster_min_weights" weights) BOTH networks cluster
    in the paired launches, while the output encoder launched on its own (dib_mlp_small_fwd / _bwd) stays on one workgroup per
    tile: same values in another fp32 summation order - outputs elementwise, gradients in norm.  (test_companion_grids_equal_the_
    separate_launches keeps the bit-identity for the [128, 128] encoder of the reference's loop, which does not cluster.)"""
    from dib_amd.dense import DenseStack
    from dib_amd.engine import HipEngine
    spec = _SMALL_ARCHS["pendulum"][0]
    eng = HipEngine(**spec_kwargs(spec), init_seed=3)
    eng.set_beta(0.05)
    D = spec.output_dimensionality
    ds = DenseStack(eng, 6, [256, 256], D, "relu", True, 5, seed=9)
    rng = np.random.default_rng(B)
    nin = sum(spec.feature_dimensionalities)
    xd = eng.to_device(rng.standard_normal((B + 5, nin)).astype(np.float32))
    yd = eng.to_device(rng.standard_normal((B + 5, 6)).astype(np.float32))
    idx = eng.to_device(rng.permutation(B + 5)[:B].astype(np.int32), dtype=torch.int32)
    gp = eng.to_device(rng.standard_normal((B, D)).astype(np.float32) / B)
    gy = eng.to_device(rng.standard_normal((B, D)).astype(np.float32) / B)
    recs = []
    for paired in (False, True, True):
        eng.grads.zero_(); ds.grads.zero_()
        comp = ds.companion_forward(yd, rows=idx) if paired else None
        eng.forward(xd, idx, 0, B, 7, 3, companion=comp)
        emb_y = (ds.companion_output() if paired else ds.forward(yd, rows=idx)).clone()
        pred = eng.pred(B).clone()
        compb = ds.companion_backward(gy) if paired else None
        eng.backward_from_pred_grad(gp, idx, 0, B, 7, 3, inv_global_batch=1.0 / B, companion=compb)
        ds.backward(gy, dgrad_done=paired)
        torch.cuda.synchronize()
        recs.append(dict(emb_y=emb_y, pred=pred, xg=eng.grads.clone(), yg=ds.grads.clone(), gu=eng.g_u(B).clone()))
    a, b, c = recs
    for k in a:
        assert torch.equal(b[k], c[k]), ("replay", k)
        ref, got = a[k].double(), b[k].double()
        if k in ("emb_y", "pred"):
            assert (got - ref).abs().max() <= 3e-5 * (1e-6 + ref.abs().max()) + 1e-6, k
        else:
            assert (got - ref).norm() <= 5e-3 * ref.norm(), (k, ((got - ref).norm() / ref.norm()).item())
    assert torch.isfinite(b["yg"]).all() and b["yg"].abs().max() > 0


def test_infonce_training_loop_on_pendulum(tmp_path):
    """BASELINE config 2 path: the custom InfoNCE loop (train.py:180-289) on simulated double-pendulum data -
    runs end to end, the InfoNCE loss drops below its untrained value ~ 2 ln B, series have the reference shapes."""
    from dib_amd import train
    out = train.main(["--dataset", "double_pendulum", "--infonce_loss", "True", "--data_path", str(tmp_path),
                      "--pendulum_number_trajectories", "6", "--number_pretraining_epochs", "2",
                      "--number_annealing_epochs", "3", "--batch_size", "256", "--learning_rate", "1e-3",
                      "--artifact_outdir", str(tmp_path), "--beta_start", "1e-4", "--beta_end", "1e-2"])
    # the reference's loop takes epoch_steps[-1] steps, so it records number_epochs - 1 boundaries (train.py:236-240)
    assert out["kl"].shape == (4, 4) and out["kl_validation"].shape == (4, 4) and out["beta"].shape == (4,)
    assert np.isfinite(out["loss_infonce"]).all() and np.isfinite(out["kl"]).all()
    assert out["loss_infonce"][-1] < out["loss_infonce"][0] < 2 * np.log(256) + 0.5
    assert abs(out["beta"][0] - 1e-4) < 1e-10 and abs(out["beta"][-1] - 1e-4 * 100 ** (1 / 3)) < 1e-7


def test_autograd_bridge_matches_oracle_gradients():
    """Custom-loop contract (reference train.py:196-220): `model.forward_autograd(x)` is differentiable under
    torch.autograd; gradients of  mean-loss(pred) + kl_loss  equal the oracle's, and a torch optimizer over
    `model.flat_parameters` updates the weights the kernels use."""
    import dib_amd
    spec = SPECS["boolean4_32x32"]
    model = dib_amd.DistributedIBNet(**spec_kwargs(spec), noise_seed=5, init_seed=3)
    model.beta.assign(0.2)
    eng = model._ensure_engine()
    p = flat_to_params(eng.blocks, eng.get_flat_params(), spec)
    rng = np.random.default_rng(0)
    B = 50
    x = rng.standard_normal((B, 4)).astype(np.float32)
    y = rng.integers(0, 2, (B, 1)).astype(np.float32)
    pred, kl_loss = model.forward_autograd(x)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.tensor(y, device=pred.device)) + kl_loss
    loss.backward()
    got = model.flat_parameters.grad.cpu().numpy()
    c = orc.forward(spec, p, x.astype(np.float64), orc.philox_normal_all(5, 0, np.arange(B), 4, 32))
    task, grads, _ = orc.backward(spec, p, x.astype(np.float64), y, c, 0.2, "bce_logits")
    ref = params_to_flat(eng.blocks, grads, eng.params.numel()).astype(np.float64)
    assert abs(float(loss.item()) - (task + 0.2 * c.kl.sum())) < 2e-4
    assert np.abs(got - ref).max() < 3e-4 * (np.abs(ref).max() + 1e-3)
    before = eng.get_flat_params().copy()
    opt = torch.optim.SGD([model.flat_parameters], lr=0.1)
    opt.step()
    after = eng.get_flat_params()
    assert np.allclose(after, before - 0.1 * got, atol=1e-6) and not np.allclose(after, before)


def test_hipgraph_fit_is_bit_identical_to_eager_and_matches_oracle(monkeypatch):
    """fit() can replay the captured step (hipGraph, DIB_ENABLE_GRAPHS=1): History must be bit-identical to the eager
    launch sequence (same kernels, device-resident noise step counter) and match the oracle like the eager path does."""
    import dib_amd
    spec = orc.DIBSpec([1, 1, 1, 1], [32, 32], [64, 64], 1, feature_embedding_dimension=8)
    x, y = orc.boolean_circuit_truth_table([0, 1, 2, 3, [1, 1, 3], [0, 4, 0], [2, 2, 5]], 4)  # SI circuit (d)
    x = np.tile(x, (33, 1)).astype(np.float32)[:520]   # 520 rows: 8 full batches of 64 + a partial one (eager)
    y = np.tile(y, 33).astype(np.float32)[:520]

    def run():
        model = dib_amd.DistributedIBNet(**spec_kwargs(spec), noise_seed=4, shuffle_seed=6, init_seed=2)
        opt = dib_amd.optimizers.get("adam")
        opt.learning_rate = 2e-3
        model.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
        cb = dib_amd.InfoBottleneckAnnealingCallback(1e-3, 0.5, 2, 6)
        p0 = flat_to_params(model.param_blocks(), model.get_flat_weights(), spec)
        h = model.fit(x, y, epochs=9, shuffle=True, batch_size=64, callbacks=[cb], verbose=False,
                      validation_data=(x[:100], y[:100]))
        return h.history, model.get_flat_weights(), p0, model

    monkeypatch.setenv("DIB_ENABLE_GRAPHS", "1")
    hist_g, w_g, p0, model_g = run()
    assert getattr(model_g._engine, "step_dev", None) is not None, "graph path was not taken"
    monkeypatch.setenv("DIB_ENABLE_GRAPHS", "0")
    hist_e, w_e, _, model_e = run()
    assert getattr(model_e._engine, "step_dev", None) is None
    for k in hist_e:
        assert hist_g[k] == hist_e[k], k
    assert np.array_equal(w_g, w_e)
    ref = orc.fit(spec, p0, x, y, epochs=9, batch_size=64, loss_kind="bce_logits",
                  beta_fn=lambda e: orc.beta_schedule(e, 1e-3, 0.5, 2, 6), lr=2e-3, shuffle=True,
                  validation_data=(x[:100], y[:100]), noise_seed=4, shuffle_seed=6, metrics=["accuracy"])
    for k in ref:
        got, want = np.array(hist_g[k]), np.array(ref[k])
        tol = 1e-3 if "KL" in k else 3e-3
        assert np.abs(got - want).max() < tol * (1 + np.abs(want).max()), (k, got, want)


def test_science_level_si_circuit_information_allocation():
    """SURVEY section 4 science-level KAT on SI circuit (c) of the reference notebook (Boolean_circuits.ipynb:987-992;
    Y = AND(XOR(AND(x2,x0),x3),x1), H(Y) = 0.811 bits, Shapley values [0.096, 0.377, 0.096, 0.242]):
    with beta small the DIB predicts perfectly; as beta is annealed the features are dropped in the order of their
    importance (x0/x2 first, then x3, x1 last); at large beta all KL -> 0 and the loss -> H(Y)."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location(
        "si_circuit_run", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "si_circuit_run.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    kl, loss_bits, acc, beta = mod.run()
    assert acc[99] == 1.0 and loss_bits[99] < 0.05                       # end of pre-training (beta = 1e-3)
    mid = kl[450]                                                        # beta ~ 0.27
    assert mid[1] > 0.2 and mid[1] > mid[3] > max(mid[0], mid[2]), mid   # x1 kept longest, then x3, x0/x2 gone first
    assert kl[-1].sum() < 0.02 and abs(loss_bits[-1] - 0.811) < 0.05 and abs(acc[-1] - 0.75) < 1e-6


def test_science_level_paper_circuit_gates_are_dropped_in_the_order_the_reference_notebook_printed():
    """The paper's 10-input circuit (`train.py --dataset boolean_circuit`, reference data.py:21-81) through DistributedIBNet.fit
    with the train.py architecture on a compressed schedule, against an outcome of the REFERENCE'S OWN TensorFlow run: cell 7 of
    complex_systems/InfoDecomp_Boolean_circuits.ipynb printed "Sequence of selected subsets: [0 1 2 5 6 7 8 9], [0 1 2 5 7 8 9],
    [2 5 7 8 9], [2 5 9], [2 9], [2], []" (threshold 0.1 bits) - gates {3, 4} lose their information first, then 6, then {0, 1},
    then {7, 8}, then 5, then 9, gate 2 last.  Also: pre-training reaches accuracy 1, and at the end of the ramp every KL is 0 and
    the loss is H(Y) = 0.758 bits (the notebook's "Entropy of Y", cell 5)."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location(
        "paper_circuit_run", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "paper_circuit_run.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    for kw in (dict(epochs_pre=1000, beta_start=1e-5, seed=2), dict(epochs_pre=500, beta_start=1e-6, seed=4)):
        drop, kl, loss_bits, acc, beta = mod.run(**kw)
        pre = kw["epochs_pre"]
        # (validation runs with the noise ON, reference train.py:263-265: one borderline row of the 1024 may flip in a given
        # epoch - the bar is the loss, 0.02 bits of the 0.758 there are to explain, and at most 2 rows off)
        assert acc[pre - 1] >= 1.0 - 2.0 / 1024 and acc[pre - 10: pre].max() == 1.0 and loss_bits[pre - 1] < 0.02, \
            (kw, acc[pre - 1], loss_bits[pre - 1])
        assert drop.min() > pre, (kw, drop)                                  # nothing is given up before the ramp starts
        assert mod.group_order_violations(drop) == [], (kw, drop.tolist())
        assert kl[-1].sum() < 0.02 and abs(loss_bits[-1] - 0.758) < 0.02, (kw, kl[-1].sum(), loss_bits[-1])
        # the exhaustive subset informations I(X_S;Y) of Boolean_circuits.ipynb:425-434 (the notebook's own statements
        # executed: tests/golden/subset_mi.npz) are a CEILING on the info-plane point of every recorded epoch - the predictive
        # information H(Y) - loss can never exceed what the gates still being transmitted carry about Y (plus the few hundredths
        # of a bit the nearly-closed channels may leak, bounded by their KL).  0.03 bits: sampling noise of one epoch's draws.
        fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subset_mi.npz"))
        info = {int((c.astype(np.int64) << np.arange(10)).sum()): float(v) for c, v in zip(fx["all_on_off_combos"], fx["all_mis_bits"])}
        excess, masks = mod.info_ceiling_excess(kl, loss_bits, info, float(fx["entropy_y_bits"]))
        assert excess[5:].max() <= 0.03, (kw, float(excess.max()), int(excess.argmax()), bin(int(masks[excess.argmax()])))
        assert len(set(masks.tolist())) >= 6                                 # the ramp walks through a sequence of subsets
        late = masks[-1] == 0 and abs((float(fx["entropy_y_bits"]) - loss_bits[-1])) < 0.02
        assert late, (kw, masks[-1], loss_bits[-1])                          # empty subset at the end: no predictive information


def test_north_star_architecture_multi_step_trajectory():
    """The exact BASELINE config-3 architecture (F=64 scalar features, encoder [128,128], E=32, integration [256,256],
    1-unit logit; fused fwd/bwd + skinny output-layer kernels + split-batch wgrads + Keras-Adam) over 3 optimizer steps
    at B=4096 against the float64 PyTorch-CPU restatement with the same counter-based noise: per-feature KL within
    1e-3 nats (the BASELINE tolerance), loss and final parameters close."""
    from dib_torch_cpu import TorchCpuDIB
    from dib_amd.engine import HipEngine
    spec = orc.DIBSpec([1] * 64, [128, 128], [256, 256], 1)
    eng = HipEngine(**spec_kwargs(spec), init_seed=7)
    p = flat_to_params(eng.blocks, eng.get_flat_params(), spec)
    ref = TorchCpuDIB(spec, p, dtype=torch.float64)
    B = 4096
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, 64)).astype(np.float32)
    w = rng.standard_normal(8)
    y = ((x[:, :8] @ w + 0.5 * x[:, 0] * x[:, 1]) > 0).astype(np.float32)[:, None]
    xd, yd = eng.to_device(x), eng.to_device(y)
    xt, yt = torch.tensor(x, dtype=torch.float64), torch.tensor(y, dtype=torch.float64)
    beta, lr = 3e-2, 3e-4
    eng.set_beta(beta)
    eng.set_lr(lr)
    for step in range(3):
        eng.train_step(xd, yd, None, 0, B, 5, step, "bce_logits")
        so = eng.step_out(B).cpu().numpy()
        eng.adam_step()
        eps = torch.tensor(orc.philox_normal_all(5, step, np.arange(B), 64, 32))
        task, kl, _ = ref.train_step(xt, yt, eps, beta, "bce_logits", lr=lr)
        assert np.abs(so[:64] / B - kl.numpy()).max() < 1e-3, (step, "per-feature KL")
        assert abs(so[64] / B - task) < 2e-4 * (1 + abs(task)), (step, "task loss")
    got = flat_to_params(eng.blocks, eng.get_flat_params(), spec)
    for a, b in zip(got.tensors(), [t.detach().numpy() for t in ref.tensors()]):
        assert np.abs(a - b).max() < 5e-5, "parameters after 3 Adam steps"


def _random_spec(rng, row_tiles=False):
    if row_tiles:   # architectures the row-tile kernels cover (csrc/dib_api.hip sb_enc / sb_int): widths % 16 == 0, inputs <= 15 wide
        F = int(rng.integers(1, 6))
        pe = bool(rng.integers(0, 2))
        nf = int(rng.integers(1, 6)) if pe else 1
        dims = [int(v) for v in rng.integers(1, 15 // nf + 1, F)]
        w = [16, 32, 48, 64, 96, 128]
        E = int(rng.choice([8, 16, 24, 32]))
        while (F * E) % 16:
            E += 8
        return orc.DIBSpec(dims, [int(rng.choice(w)), int(rng.choice(w))], [int(rng.choice(w)) for _ in range(int(rng.integers(1, 4)))],
                           int(rng.integers(1, 6)), use_positional_encoding=pe, number_positional_encoding_frequencies=nf,
                           activation_fn=[None, "relu", "leaky_relu"][int(rng.integers(0, 3))], feature_embedding_dimension=E)
    F = int(rng.integers(1, 6))
    dims = [int(v) for v in rng.integers(1, 7, F)]
    enc = [int(v) for v in rng.integers(1, 70, int(rng.integers(0, 4)))]
    integ = [int(v) for v in rng.integers(1, 70, int(rng.integers(0, 3)))]
    act = [None, "relu", "leaky_relu", "tanh", "sigmoid", "elu", "softplus"][int(rng.integers(0, 7))]
    return orc.DIBSpec(dims, enc, integ, int(rng.integers(1, 6)), use_positional_encoding=bool(rng.integers(0, 2)),
                       number_positional_encoding_frequencies=int(rng.integers(1, 6)), activation_fn=act,
                       feature_embedding_dimension=int(rng.integers(1, 41)))


@pytest.mark.parametrize("path", DISPATCH_PATHS)
@pytest.mark.parametrize("case", range(20))
def test_random_architectures_forward_backward(case, path):
    if path == "cluster_tiles" and case < 12:
        pytest.skip("cases 12+ are the ones inside the row-tile kernels' coverage")
    with dispatch_path(path):
        _random_architecture(case)


def _random_architecture(case):
    """Seeded random architectures (ragged feature widths, 0-3 encoder layers, arbitrary unit counts / embedding widths /
    activations / output widths, with and without positional encoding, arbitrary batch) through the general grouped-GEMM
    path (or the fused kernels when they apply): predictions, per-feature KL and every gradient vs the oracle."""
    rng = np.random.default_rng(1000 + case)
    spec = _random_spec(rng, row_tiles=case >= 12)   # cases 12+: inside the row-tile kernels' coverage (default path runs them)
    eng, p = _engine(spec, seed=case)
    F, E = spec.number_features, spec.feature_embedding_dimension
    B = int(rng.integers(1, 200))
    x = rng.standard_normal((B, sum(spec.feature_dimensionalities))).astype(np.float32)
    if spec.output_dimensionality == 1:
        kind, y = "bce_logits", rng.integers(0, 2, (B, 1)).astype(np.float32)
    elif case % 2:
        kind, y = "mse", rng.standard_normal((B, spec.output_dimensionality)).astype(np.float32)
    else:
        kind, y = "sparse_cce_logits", rng.integers(0, spec.output_dimensionality, (B, 1)).astype(np.float32)
    beta = float(10 ** rng.uniform(-3, 0))
    eng.set_beta(beta)
    eng.train_step(eng.to_device(x), eng.to_device(y), None, 0, B, 21, case, kind)
    eps = orc.philox_normal_all(21, case, np.arange(B), F, E)
    c = orc.forward(spec, p, x.astype(np.float64), eps)
    task, grads, _ = orc.backward(spec, p, x.astype(np.float64), y, c, beta, kind)
    so = eng.step_out(B).cpu().numpy()
    assert np.abs(eng.pred(B).cpu().numpy() - c.pred).max() <= 3e-4 * (1 + np.abs(c.pred).max()), spec
    assert np.abs(so[:F] / B - c.kl).max() < 1e-3 * (1 + np.abs(c.kl).max()), spec
    assert abs(so[F] / B - task) < 3e-4 * (1 + abs(task)), spec
    gflat = eng.get_flat_grads()
    gref = params_to_flat(eng.blocks, grads, eng.params.numel()).astype(np.float64)
    for b in eng.blocks:
        sl = slice(b["offset"], b["offset"] + b["rows"] * b["cols"])
        assert np.abs(gflat[sl] - gref[sl]).max() <= 5e-4 * (np.abs(gref[sl]).max() + 1e-3), (spec, b)


@pytest.mark.parametrize("name", ["tabular8_default", "boolean4_32x32", "pendulum_ragged"])
def test_inference_forward_skips_stashes_but_not_results(name):
    """DIB_FWD_INFERENCE (validation / predict): same prediction, (mu|logvar), u and KL sums bit for bit; a training
    step afterwards is unaffected."""
    spec = SPECS[name]
    eng, _ = _engine(spec, seed=5)
    rng = np.random.default_rng(1)
    B = 300
    x = eng.to_device(rng.standard_normal((B, sum(spec.feature_dimensionalities))).astype(np.float32))
    eng.forward(x, None, 0, B, 3, 7)
    ref = [eng.pred(B).clone(), eng.enc_out(B).clone(), eng.u(B).clone(), eng.step_out(B).clone()]
    eng.forward(x, None, 0, B, 3, 7, inference=True)
    got = [eng.pred(B), eng.enc_out(B), eng.u(B), eng.step_out(B)]
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    if spec.output_dimensionality == 1:
        y = eng.to_device((rng.random((B, 1)) > 0.5).astype(np.float32))
        eng.set_beta(0.1)
        eng.train_step(x, y, None, 0, B, 3, 8, "bce_logits")
        g1 = eng.grads.clone()
        eng.forward(x, None, 0, B, 3, 9, inference=True)     # an evaluation in between must not disturb anything
        eng.train_step(x, y, None, 0, B, 3, 8, "bce_logits")
        assert torch.equal(g1, eng.grads)


def test_c_host_program_drives_a_training_step_through_the_c_abi():
    """examples/c_abi_step.c: plain C11 + HIP runtime, no Python / torch in the process.  Its per-feature KL, task loss
    and gradient checksums must equal what the Python engine computes from the same closed-form inputs."""
    import subprocess
    import __graft_entry__ as ge
    exe = ge.build_c_example()
    B = 300
    out = subprocess.run([exe, str(B)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    vals = {ln.split()[0]: ln.split()[1:] for ln in out.stdout.splitlines() if ln and ln.split()[0] in
            ("params", "KL", "task_loss", "grad_sum")}

    def pattern(i0, n, scale):  # the C program's generator, float32 throughout
        i = np.arange(i0, i0 + n, dtype=np.int64)
        t = np.float32(0.37) * (i % 1009).astype(np.float32) + np.float32(0.001) * (i % 7919).astype(np.float32)
        return (np.float32(scale) * np.sin(t.astype(np.float32))).astype(np.float32)

    from dib_amd.engine import HipEngine
    eng = HipEngine([1, 1, 1, 1], [32, 32], [64], 1, feature_embedding_dimension=32, device="cuda:0", init_seed=0)
    assert int(vals["params"][0]) == eng.n_params
    eng.set_flat_params(pattern(0, eng.n_params, 0.2))
    x = pattern(12345, B * 4, 1.5).reshape(B, 4)
    y = (pattern(777, B, 1.0) > 0).astype(np.float32)[:, None]
    eng.set_beta(0.25)
    eng.train_step(eng.to_device(x), eng.to_device(y), None, 0, B, 42, 3, "bce_logits")
    so = eng.step_out(B).cpu().numpy()
    kl_c = np.array([float(v) for v in vals["KL"]])
    assert np.abs(kl_c - so[:4] / B).max() < 1e-5 * (1 + np.abs(kl_c).max())
    assert abs(float(vals["task_loss"][0]) - so[4] / B) < 1e-5
    g = eng.get_flat_grads().astype(np.float64)
    assert abs(float(vals["grad_sum"][0]) - g.sum()) < 1e-4 * (1 + np.abs(g).sum())
    assert abs(float(vals["grad_sum"][2]) - np.abs(g).sum()) < 1e-4 * (1 + np.abs(g).sum())


@pytest.mark.parametrize("kind,B", [("bce_logits", 300), ("mse", 37), ("bce_logits", 8192 + 5)])
def test_fused_output_head_equals_the_unfused_sequence(kind, B):
    """dib_output_head_fused (output Dense(1) + loss + its backward in one pass, what train_step uses) against the separate
    dib_integration_fwd -> dib_loss_fwd_bwd -> dib_integration_bwd sequence: same predictions, loss sums, accuracy and
    gradients (different summation order only)."""
    spec = orc.DIBSpec([1] * 6, [32, 32], [256, 64], 1, feature_embedding_dimension=32)
    eng, _ = _engine(spec, 3)
    kind_id = {"bce_logits": 0, "mse": 3}[kind]
    assert eng.lib.dib_output_head_fused_supported(eng.layout, kind_id) == 1
    rng = np.random.default_rng(B)
    x = rng.standard_normal((B, 6)).astype(np.float32)
    y = (rng.random((B, 1)) > 0.5).astype(np.float32) if kind == "bce_logits" else rng.standard_normal((B, 1)).astype(np.float32)
    xd, yd = eng.to_device(x), eng.to_device(y)
    eng.set_beta(0.1)
    eng.train_step(xd, yd, None, 0, B, 2, 3, kind, accumulate=False)            # fused head
    g_f, so_f, pred_f = eng.grads.clone(), eng.step_out(B).clone(), eng.pred(B).clone()
    eng.forward(xd, None, 0, B, 2, 3)                                            # unfused sequence
    eng.loss(kind, yd, None, 0, B, 1.0 / B)
    eng.backward(None, 0, B, 2, 3, 1.0 / B)
    g_u, so_u, pred_u = eng.grads, eng.step_out(B), eng.pred(B)
    assert torch.equal(pred_f, pred_u) or (pred_f - pred_u).abs().max() < 1e-6
    assert (so_f - so_u).abs().max() <= 1e-5 * (1 + so_u.abs().max())
    assert so_f[6 + 1] == so_u[6 + 1] and so_f[6 + 2] == B                       # accuracy count, rows
    assert (g_f - g_u).abs().max() <= 2e-5 * g_u.abs().max()


def _legacy_step(eng, xd, yd, B, seed, step, kind, opt):
    """The step as rounds 1-4 launched it - every reduction, the metric accumulation, Adam and its counter bump as separate
    entry points of include/dib_hip.h - for comparison with the one-launch tail (csrc/dib_tail.h)."""
    from dib_amd._lib import LOSS_KINDS, check
    lib, ws, st, P = eng.lib, eng.workspace(B), eng._stream(), _ptr
    k = LOSS_KINDS[kind]
    inv = 1.0 / B
    fused = bool(lib.dib_output_head_fused_supported(eng.layout, k))
    check(lib.dib_encoder_bank_fwd(eng.layout, P(xd), xd.stride(0), None, 0, B, P(eng.params), seed, step, 0, P(ws), st))
    if fused:
        check(lib.dib_integration_fwd_hidden(eng.layout, B, P(eng.params), P(ws), st))
        check(lib.dib_output_head_fused(eng.layout, k, P(yd), yd.stride(0), None, 0, B, inv, 0, P(eng.params), P(eng.grads), P(ws), st))
        check(lib.dib_integration_bwd_hidden(eng.layout, B, P(eng.params), P(eng.grads), P(ws), st))
    else:
        check(lib.dib_integration_fwd(eng.layout, B, P(eng.params), P(ws), st))
        check(lib.dib_loss_fwd_bwd(eng.layout, k, P(yd), yd.stride(0), None, 0, B, inv, 0, P(ws), st))
        check(lib.dib_integration_bwd(eng.layout, B, P(eng.params), P(eng.grads), P(ws), st))
    check(lib.dib_encoder_bank_bwd(eng.layout, B, P(eng.params), P(eng.grads), P(eng.beta_dev), inv, P(ws), st))
    check(lib.dib_grads_finalize(eng.layout, B, P(eng.grads), P(ws), st))
    check(lib.dib_metrics_accumulate(eng.layout, B, P(eng.beta_dev), inv, P(eng.metrics_acc), P(ws), st))
    if opt == "adam":
        eng.adam_step()
    else:
        eng.sgd_step()


@pytest.mark.parametrize("opt", ["adam", "sgd"])
@pytest.mark.parametrize("small", [0, 1])
@pytest.mark.parametrize("arch,B", [("north", 128), ("north", 1500), ("north", 8192 + 5), ("general", 300), ("general", 2048),
                                    ("wide_out", 640)])
def test_step_tail_equals_the_separate_launches(arch, B, small, opt):
    """dib_step_tail (one launch: bucket reduce + fused-head reduce + KL / loss sums + metrics + optimizer + counter bump)
    against the separate entry points it replaces, three steps in a row on twin engines: gradients, the per-step scalars,
    the History accumulators and Adam's step count bit for bit (same fixed summation orders); parameters and moments to one
    ulp-level tolerance.  small = 0: the large-batch kernels on both sides.  small = 1 (batches in the row-tile regime): train_step runs the
    integration network's forward + head + dgrad chain as ONE row-tile launch (dib_integration_head_step), the separate entry
    points run it in pieces (hidden forward / fused head kernel / dgrad chain from the stashes): same values, different
    summation order in the head - compared to fp32 tolerance instead."""
    from dib_amd import _lib
    from dib_amd.engine import HipEngine
    if small and B > 1024:
        pytest.skip("beyond the row-tile regime for these layouts (dib_set_tuning small_wgs)")
    _lib.set_tuning("small_batch", small)
    try:
        _step_tail_case(arch, B, small, opt, HipEngine)
    finally:
        _lib.set_tuning("small_batch", 1)


def _step_tail_case(arch, B, small, opt, HipEngine):
    spec = {"north": orc.DIBSpec([1] * 6, [128, 128], [256, 256], 1, feature_embedding_dimension=32),            # fused kernels + fused head
            "general": orc.DIBSpec([2, 1, 3], [48], [40, 24], 1, feature_embedding_dimension=8),                  # grouped GEMMs + fused head
            "wide_out": orc.DIBSpec([1, 2], [32, 32], [64], 3, feature_embedding_dimension=8)}[arch]              # unfused loss
    kind = "mse" if arch == "wide_out" else "bce_logits"
    rng = np.random.default_rng(B)
    nin = sum(spec.feature_dimensionalities)
    x = rng.standard_normal((B, nin)).astype(np.float32)
    y = (rng.random((B, spec.output_dimensionality)) > 0.5).astype(np.float32)
    a, b = HipEngine(**spec_kwargs(spec), init_seed=5), HipEngine(**spec_kwargs(spec), init_seed=5)
    xd, yd = a.to_device(x), a.to_device(y)
    for eng in (a, b):
        eng.set_beta(0.05)
        eng.set_lr(1e-3)
    tup = ("adam", 0.9, 0.999, 1e-7) if opt == "adam" else ("sgd",)
    for step in range(3):
        a.train_step(xd, yd, None, 0, B, 7, step, kind, optimizer=tup)
        _legacy_step(b, xd, yd, B, 7, step, kind, opt)
        torch.cuda.synchronize()
        if small:
            for ta, tb in ((a.grads, b.grads), (a.step_out(B), b.step_out(B)), (a.metrics_acc, b.metrics_acc)):
                assert (ta - tb).abs().max() <= 3e-5 * tb.abs().max() + 1e-12, (step, (ta - tb).abs().max(), tb.abs().max())
        else:
            assert torch.equal(a.grads, b.grads), (step, (a.grads - b.grads).abs().max())
            assert torch.equal(a.step_out(B), b.step_out(B))
            assert torch.equal(a.metrics_acc, b.metrics_acc)
        assert int(a.t_dev.item()) == int(b.t_dev.item()) == (step + 1 if opt == "adam" else 0)
        if small:   # Adam's first steps are sign-like (m / sqrt(v)): rounding-level gradient differences do not stay small
            b.params.copy_(a.params); b.adam_m.copy_(a.adam_m); b.adam_v.copy_(a.adam_v)
            continue
        for ta, tb in ((a.params, b.params), (a.adam_m, b.adam_m), (a.adam_v, b.adam_v)):
            assert (ta - tb).abs().max() <= 1e-6 * tb.abs().max() + 1e-12
    # the validation tail: KL / loss sums + metrics in one launch, and the fused head without its gradient
    a.eval_step(xd, yd, None, 0, B, 7, 99, kind)
    b.forward(xd, None, 0, B, 7, 99, inference=True)
    b.loss(kind, yd, None, 0, B, 1.0 / B)
    b.accumulate_metrics(B, 1.0 / B)
    so_a, so_b = a.step_out(B), b.step_out(B)
    assert torch.equal(so_a[: spec.number_features], so_b[: spec.number_features])
    assert (so_a - so_b).abs().max() <= 1e-5 * (1 + so_b.abs().max())
    assert (a.metrics_acc - b.metrics_acc).abs().max() <= 1e-5 * (1 + b.metrics_acc.abs().max())
    assert (a.pred(B) - b.pred(B)).abs().max() < 1e-5


def test_tuning_switchboard_is_the_only_hidden_input():
    """include/dib_hip.h dib_set_tuning / dib_get_tuning: every documented key round-trips, unknown keys and negative values
    are refused, and "fused_encoder" = 0 really selects the grouped-GEMM path for layouts created afterwards."""
    from dib_amd import _lib
    from dib_amd.engine import HipEngine
    lib = _lib.load_library()
    for key in ("fwd_small_wgs", "fwd_narrow_wgs", "stream_rows", "split_policy", "split_overhead", "fused_encoder", "fused_head",
                "small_batch", "small_wgs", "mlp_row_tiles", "infonce_one_launch", "attn_small_bwd_waves", "wgrad_flat_tile", "wgrad_max_splits", "num_cus",
                "int_cluster", "int_cluster_wgs", "int_cluster_min_weights", "int_cluster_short_exchange", "attn_fwd_waves"):
        v = _lib.get_tuning(key)
        _lib.set_tuning(key, v + 1)
        assert _lib.get_tuning(key) == v + 1
        _lib.set_tuning(key, v)
    assert lib.dib_set_tuning(b"no_such_key", 1) == -1 and lib.dib_set_tuning(b"fused_head", -3) == -1
    assert _lib.get_tuning("num_cus") == 0   # 0 = the current device's own compute-unit count (no process-wide "the device")
    spec = orc.DIBSpec([1] * 4, [128, 128], [256], 1, feature_embedding_dimension=32)
    try:
        _lib.set_tuning("fused_head", 0)
        eng = HipEngine(**spec_kwargs(spec), init_seed=1)
        assert lib.dib_output_head_fused_supported(eng.layout, 0) == 0
    finally:
        _lib.set_tuning("fused_head", 1)
    assert lib.dib_output_head_fused_supported(eng.layout, 0) == 1


_SMALL_ARCHS = {
    # reference default (train.py:36-44): fused-eligible encoders, [256, 256] integration network, fused 1-unit head
    "north": (orc.DIBSpec([1] * 10, [128, 128], [256, 256], 1, feature_embedding_dimension=32), "bce_logits"),
    # pendulum layout of the InfoNCE path (train.py:184-192): ragged features, 64-wide output = the shared space
    "pendulum": (orc.DIBSpec([2, 1, 2, 1], [128, 128], [256, 256], 64, feature_embedding_dimension=32), "mse"),
    # narrow layers: one / two column tiles per wave pass, three integration layers, leaky_relu
    "narrow": (orc.DIBSpec([1, 3, 2], [64, 32], [128, 48, 32], 1, feature_embedding_dimension=16, activation_fn="leaky_relu"), "mse"),
    # no positional encoding, one integration layer, 16-wide general output
    "plain": (orc.DIBSpec([4, 2], [32, 64], [64], 16, use_positional_encoding=False, feature_embedding_dimension=16), "mse"),
}


@pytest.mark.parametrize("B", [1, 16, 37, 128, 1000, 2048])
@pytest.mark.parametrize("linear", [True, False])
@pytest.mark.parametrize("arch", sorted(_SMALL_ARCHS))
def test_small_batch_row_tile_kernels_equal_the_large_batch_path(arch, linear, B):
    """csrc/dib_small.h (16-row tiles; the regime is (row tiles x features) <= "small_wgs" = 512 and <= 2048 rows: "north" with its 10
    features leaves it between B = 128 and 1000 - both runs then take the large-batch kernels - the others stay in it up to
    2048) against the large-batch kernels on the SAME engine, switched with
    dib_set_tuning("small_batch", .): forward stashes, KL / loss sums, prediction, every gradient, the validation step and the
    custom-loss contract, to fp32 summation-order tolerance.  linear=True (no activation): every quantity elementwise - the
    whole plumbing with no kinks.  linear=False (the architecture's own relu / leaky_relu): forward quantities elementwise;
    a hidden unit whose pre-activation is within rounding of 0 may take the other branch of act' in the other summation
    order (one row of dL/du at B = 1000 in profiles/r05e_*), so gradients are compared in norm and at most 0.5 % of the rows
    of dL/du may differ.  The parity zoo against the float64 oracle runs on the small path by default."""
    import dataclasses
    from dib_amd import _lib
    from dib_amd.engine import HipEngine
    spec, kind = _SMALL_ARCHS[arch]
    if linear:
        spec = dataclasses.replace(spec, activation_fn=None) if dataclasses.is_dataclass(spec) else \
            orc.DIBSpec(**dict(spec_kwargs(spec), activation_fn=None))
    rng = np.random.default_rng(B + len(arch))
    nin = sum(spec.feature_dimensionalities)
    x = rng.standard_normal((B + 3, nin)).astype(np.float32)
    y = rng.standard_normal((B + 3, spec.output_dimensionality)).astype(np.float32)
    if kind == "bce_logits":
        y = (y > 0).astype(np.float32)
    eng = HipEngine(**spec_kwargs(spec), init_seed=11)
    eng.set_beta(0.07)
    xd, yd = eng.to_device(x), eng.to_device(y)
    idx = eng.to_device(rng.permutation(B + 3)[:B].astype(np.int32), dtype=torch.int32)
    outs = []
    try:
        for small in (1, 0):
            _lib.set_tuning("small_batch", small)
            eng.metrics_acc.zero_()
            eng.train_step(xd, yd, idx, 0, B, 5, 9, kind)
            torch.cuda.synchronize()
            rec = dict(grads=eng.grads.clone(), step_out=eng.step_out(B).clone(), pred=eng.pred(B).clone(),
                       enc_out=eng.enc_out(B).clone(), u=eng.u(B).clone(), g_u=eng.g_u(B).clone(),
                       h1=eng.enc_h(B, 0).clone(), h2=eng.enc_h(B, 1).clone(), int_h0=eng.int_h(B, 0).clone(),
                       metrics=eng.metrics_acc.clone())
            eng.metrics_acc.zero_()
            eng.eval_step(xd, yd, None, 2, B, 5, 77, kind)
            torch.cuda.synchronize()
            rec.update(val_step_out=eng.step_out(B).clone(), val_pred=eng.pred(B).clone(), val_metrics=eng.metrics_acc.clone())
            # custom-loss contract (the InfoNCE loop): forward, a caller-made dL/dpred, backward_from_pred_grad
            eng.forward(xd, idx, 0, B, 5, 10)
            gp = torch.sin(torch.arange(B * spec.output_dimensionality, device=eng.device, dtype=torch.float32)).view(B, -1) / B
            eng.backward_from_pred_grad(gp, idx, 0, B, 5, 10)
            torch.cuda.synchronize()
            rec.update(custom_pred=eng.pred(B).clone(), custom_grads=eng.grads.clone(), custom_g_u=eng.g_u(B).clone())
            outs.append(rec)
    finally:
        _lib.set_tuning("small_batch", 1)
    s, l = outs
    bad = {}
    for k in s:
        a, ref = s[k].double(), l[k].double()
        scale = 1e-6 + ref.abs().max().item()
        err = (a - ref).abs().max().item()
        if linear or not ("grads" in k or "g_u" in k):
            if err > 3e-5 * scale:
                bad[k] = (err, scale)
        elif "g_u" in k:    # rows that crossed a kink of the activation
            rows = ((a - ref).abs().max(dim=1).values > 3e-5 * scale).sum().item()
            if rows > max(1, B // 200):
                bad[k] = ("rows", rows)
        else:
            rel = ((a - ref).norm() / (ref.norm() + 1e-30)).item()
            if rel > 5e-3:
                bad[k] = ("norm", rel)
    assert not bad, bad
    assert torch.isfinite(s["grads"]).all()


@pytest.mark.parametrize("B", [1, 37, 128, 300, 512])
@pytest.mark.parametrize("cl", [2, 4, 8, -4])
@pytest.mark.parametrize("arch", sorted(_SMALL_ARCHS))
def test_integration_cluster_equals_one_workgroup_per_tile(arch, cl, B):
    """csrc/dib_small.h cluster mode (dib_set_tuning("int_cluster", cl): every row tile of the integration network on cl workgroups,
    each a column slice of every layer, slices exchanged through L2 behind per-(tile, layer) arrival counters) against the
    one-workgroup-per-tile kernel on the SAME engine: training step (forward + 1-unit head + dgrad chain, or the general output
    layer), validation step (no stashes: the exchange buffers) and the custom-loss contract (the backward launched on its own from
    the stashes).  A slice's columns are contracted by one workgroup but in its own batch order: fp32 summation-order tolerance for
    the forward quantities; gradients in norm (a unit within rounding of a kink may take the other branch of act').  Slices of
    widths with fewer 16-column tiles than workgroups are empty ("narrow": 48 and 32 wide on 4 and 8 workgroups).  Each mode twice
    in a row: the same bits (the counters clean themselves, every sum has a fixed order)."""
    from dib_amd import _lib
    from dib_amd.engine import HipEngine
    spec, kind = _SMALL_ARCHS[arch]
    rng = np.random.default_rng(B + 7 * abs(cl))
    nin = sum(spec.feature_dimensionalities)
    x = rng.standard_normal((B + 3, nin)).astype(np.float32)
    y = rng.standard_normal((B + 3, spec.output_dimensionality)).astype(np.float32)
    if kind == "bce_logits":
        y = (y > 0).astype(np.float32)
    eng = HipEngine(**spec_kwargs(spec), init_seed=13)
    eng.set_beta(0.07)
    xd, yd = eng.to_device(x), eng.to_device(y)
    idx = eng.to_device(rng.permutation(B + 3)[:B].astype(np.int32), dtype=torch.int32)
    old = {k: _lib.get_tuning(k) for k in ("int_cluster", "int_cluster_wgs", "int_cluster_min_weights", "int_cluster_short_exchange")}
    if cl < 0:   # - 4: four workgroups per tile on the AGENT-SCOPE protocol for every exchange - what a cluster that the dispatcher
        cl = -cl  # did not place on one XCD takes (csrc/dib_small.h dib_small_cluster_exchange); same values, same bits as the short one
        _lib.set_tuning("int_cluster_short_exchange", 0)

    def run():
        eng.metrics_acc.zero_()
        eng.train_step(xd, yd, idx, 0, B, 5, 9, kind)
        rec = dict(grads=eng.grads.clone(), step_out=eng.step_out(B).clone(), pred=eng.pred(B).clone(), g_u=eng.g_u(B).clone(),
                   int_h0=eng.int_h(B, 0).clone())
        eng.eval_step(xd, yd, None, 2, B, 5, 77, kind)
        rec.update(val_step_out=eng.step_out(B).clone(), val_pred=eng.pred(B).clone())
        eng.forward(xd, idx, 0, B, 5, 10)
        gp = torch.sin(torch.arange(B * spec.output_dimensionality, device=eng.device, dtype=torch.float32)).view(B, -1) / B
        eng.backward_from_pred_grad(gp, idx, 0, B, 5, 10)
        rec.update(custom_pred=eng.pred(B).clone(), custom_grads=eng.grads.clone(), custom_g_u=eng.g_u(B).clone())
        torch.cuda.synchronize()
        return rec

    try:
        _lib.set_tuning("int_cluster_wgs", 256)
        _lib.set_tuning("int_cluster_min_weights", 0)
        outs = []
        for mode in (cl, 0):
            _lib.set_tuning("int_cluster", mode)
            a, b = run(), run()
            for k in a:
                assert torch.equal(a[k], b[k]), (mode, k)
            outs.append(a)
    finally:
        for k, v in old.items():
            _lib.set_tuning(k, v)
    c, o = outs
    bad = {}
    for k in c:
        a, ref = c[k].double(), o[k].double()
        scale = 1e-6 + ref.abs().max().item()
        if "grads" in k or "g_u" in k:
            rel = ((a - ref).norm() / (ref.norm() + 1e-30)).item()
            if rel > 5e-3:
                bad[k] = ("norm", rel)
        elif (a - ref).abs().max().item() > 3e-5 * scale + 1e-6:   # (+ 1e-6: a prediction that is itself a cancelling sum, B = 1)
            bad[k] = ((a - ref).abs().max().item(), scale)
    assert not bad, bad
    assert torch.isfinite(c["grads"]).all()


@pytest.mark.parametrize("B", [512, 1024, 2048])
def test_alternating_entry_points_leave_no_stale_gradient_slabs(B):
    """A workspace holds `nsplit` partial slabs per parameter block and the reducers sum all of them; a weight-gradient launch that
    picks FEWER splits than an earlier launch over the same block (the split rule prices whole launches: at B = 512 the row-tile
    regime's one grouped launch picks 3 slabs, the per-layer launches behind dib_backward's custom-loss entry of a 1-unit output
    pick 4) must not leave that launch's upper slabs in the sum (csrc/dib_api.hip retire_stale_slabs; found in round 6: a training step
    after a custom-loss step differed by 6e-3 in every GEMM-made gradient block).  Training step / custom-loss step / training
    step / custom-loss step on one engine = the bits of each on a fresh engine."""
    from dib_amd.engine import HipEngine
    spec = orc.DIBSpec([1] * 10, [128, 128], [256, 256], 1, feature_embedding_dimension=32)
    rng = np.random.default_rng(B)
    x = rng.standard_normal((B, 10)).astype(np.float32)
    y = (rng.random((B, 1)) > 0.5).astype(np.float32)

    def make():
        eng = HipEngine(**spec_kwargs(spec), init_seed=13)
        eng.set_beta(0.07)
        return eng

    e = make()
    xd, yd = e.to_device(x), e.to_device(y)

    def train(eng):
        eng.train_step(xd, yd, None, 0, B, 5, 9, "bce_logits")
        torch.cuda.synchronize()
        return eng.grads.clone()

    def custom(eng):
        eng.forward(xd, None, 0, B, 5, 10)
        gp = torch.sin(torch.arange(B, device=eng.device, dtype=torch.float32)).view(B, 1) / B
        eng.backward_from_pred_grad(gp, None, 0, B, 5, 10)
        torch.cuda.synchronize()
        return eng.grads.clone()

    ref_t, ref_c = train(make()), custom(make())
    for i in range(2):
        assert torch.equal(train(e), ref_t), ("training step", i)
        assert torch.equal(custom(e), ref_c), ("custom-loss step", i)
    assert torch.equal(train(e), ref_t)


@pytest.mark.parametrize("B", [128, 2048 + 77, 16384])
def test_workspace_needs_only_dib_workspace_init(B):
    """include/dib_hip.h workspace contract: whatever a workspace held before, dib_workspace_init makes it usable - it zeroes
    exactly the regions whose unwritten parts are read (split-batch gradient slabs beyond a launch's chosen split count, the
    per-step scalars, the tail's arrival counters) and writes the merged weight-gradient table.  A step on a workspace that
    was filled with NaN and re-initialised must reproduce the step on the fresh one bit for bit (a stale slab, a stale counter
    or a missing table entry would show as a NaN or a different sum)."""
    from dib_amd._lib import check
    spec = orc.DIBSpec([1] * 50, [128, 128], [256, 256], 1, feature_embedding_dimension=32)   # F = 50: ragged split choices
    eng, _ = _engine(spec, 2)
    rng = np.random.default_rng(B)
    x = rng.standard_normal((B, 50)).astype(np.float32)
    y = (rng.random((B, 1)) > 0.5).astype(np.float32)
    xd, yd = eng.to_device(x), eng.to_device(y)
    eng.set_beta(0.02)
    p0 = eng.params.clone()
    outs = []
    for poison in (False, True):
        eng.params.copy_(p0)
        eng.reset_optimizer()
        eng.metrics_acc.zero_()
        ws = eng.workspace(B)
        if poison:
            ws.fill_(float("nan"))
            check(eng.lib.dib_workspace_init(eng.layout, B, _ptr(ws), eng._stream()), "dib_workspace_init")
        for step in range(2):
            eng.train_step(xd, yd, None, 0, B, 3, step, "bce_logits", optimizer=("adam", 0.9, 0.999, 1e-7))
        eng.eval_step(xd, yd, None, 0, B, 3, 50, "bce_logits")
        torch.cuda.synchronize()
        outs.append((eng.grads.clone(), eng.params.clone(), eng.metrics_acc.clone(), eng.step_out(B).clone()))
    for a, b in zip(*outs):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)
