"""Building-block entry points of include/dib_st.h / dib_hip.h that the model-level tests reach only at a few shapes, each
against a float64 NumPy statement of the same formula: every template variant of the row softmax (the grouped-GEMM attention
path: tf.keras.layers.MultiHeadAttention's softmax, ...set_transformer.ipynb:332-389), Add + LayerNormalization at both width
classes, act' masking, the plain SGD step, the 64 x 128 forward tile that only a tuning key selects, and the 128 x 64 tiles of
very tall skinny products.  (profiles/r06_suite_kernel_coverage.txt: every kernel symbol of the library is launched by a test.)"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _lib():
    from dib_amd import _lib
    return _lib.load_library(), _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("P", [1, 50, 64, 200, 700, 3000, 5000])   # kernel variants <1> <= 64 < <4> <= 256 < <16> <= 1024 < <64> <= 4096 < <0>
def test_softmax_rows_forward_backward_vs_float64(P):
    lib, L = _lib()
    rng = np.random.default_rng(P)
    rows, ld, scale = 37, (P + 3) // 4 * 4 + 4, 0.3
    s = np.zeros((rows, ld), dtype=np.float32)
    s[:, :P] = 3.0 * rng.standard_normal((rows, P))
    dP = np.zeros((rows, ld), dtype=np.float32)
    dP[:, :P] = rng.standard_normal((rows, P))
    z = scale * s[:, :P].astype(np.float64)
    p_ref = np.exp(z - z.max(1, keepdims=True))
    p_ref /= p_ref.sum(1, keepdims=True)
    g = dP[:, :P].astype(np.float64)
    ds_ref = scale * p_ref * (g - (p_ref * g).sum(1, keepdims=True))
    sd, gd = torch.from_numpy(s).cuda(), torch.from_numpy(dP).cuda()
    L.check(lib.dib_softmax_rows_fwd(_ptr(sd), rows, P, ld, scale, _stream()), "dib_softmax_rows_fwd")
    L.check(lib.dib_softmax_rows_bwd(_ptr(sd), _ptr(gd), rows, P, ld, scale, _stream()), "dib_softmax_rows_bwd")
    torch.cuda.synchronize()
    assert np.abs(sd.cpu().numpy()[:, :P] - p_ref).max() < 2e-6
    assert np.abs(gd.cpu().numpy()[:, :P] - ds_ref).max() < 5e-6 * (1 + np.abs(ds_ref).max())


@pytest.mark.parametrize("T,D,slabs", [(100, 32, 1), (77, 64, 1), (300, 64, 3), (33, 256, 1), (1, 48, 2)])
def test_add_layernorm_forward_backward_vs_float64(T, D, slabs):
    """tf.keras.layers.Add() -> LayerNormalization(epsilon=1e-3) (width classes <= 32 and <= 256), the second addend arriving as
    split-K slabs; backward ds (gradient of both addends) and [dgamma | dbeta]."""
    lib, L = _lib()
    rng = np.random.default_rng(T + D)
    a = rng.standard_normal((T, D)).astype(np.float32)
    b = rng.standard_normal((slabs, T, D)).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(D)).astype(np.float32)
    dy = rng.standard_normal((T, D)).astype(np.float32)
    eps = 1e-3
    s = a.astype(np.float64) + b.astype(np.float64).sum(0)
    mean, var = s.mean(1, keepdims=True), s.var(1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (s - mean) * rstd
    y_ref = xhat * gamma + beta
    dxh = dy.astype(np.float64) * gamma
    ds_ref = rstd * (dxh - dxh.mean(1, keepdims=True) - xhat * (dxh * xhat).mean(1, keepdims=True))
    dgb_ref = np.concatenate([(dy * xhat).sum(0), dy.astype(np.float64).sum(0)])
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    ad, bd, gd, btd, dyd = dev(a), dev(b), dev(gamma), dev(beta), dev(dy)
    y, xh, rs = torch.empty(T, D, device="cuda"), torch.empty(T, D, device="cuda"), torch.empty(T, device="cuda")
    L.check(lib.dib_add_layernorm_fwd(_ptr(ad), _ptr(bd), slabs, T * D, T, D, _ptr(gd), _ptr(btd), eps, _ptr(y), _ptr(xh), _ptr(rs),
                                      _stream()), "dib_add_layernorm_fwd")
    ws = torch.zeros(int(lib.dib_add_layernorm_bwd_workspace_bytes(T, D)) // 4 + 4, device="cuda")
    ds, dgb = torch.empty(T, D, device="cuda"), torch.empty(2 * D, device="cuda")
    L.check(lib.dib_add_layernorm_bwd(_ptr(dyd), _ptr(xh), _ptr(rs), _ptr(gd), T, D, _ptr(ds), _ptr(dgb), _ptr(ws), _stream()),
            "dib_add_layernorm_bwd")
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - y_ref).max() < 2e-5 * (1 + np.abs(y_ref).max())
    assert np.abs(xh.cpu().numpy() - xhat).max() < 2e-5 and np.abs(rs.cpu().numpy() - rstd[:, 0]).max() < 1e-5 * rstd.max()
    assert np.abs(ds.cpu().numpy() - ds_ref).max() < 3e-5 * (1 + np.abs(ds_ref).max())
    assert np.abs(dgb.cpu().numpy() - dgb_ref).max() < 3e-5 * (1 + np.abs(dgb_ref).max())


@pytest.mark.parametrize("act,slope", [(1, 0.0), (2, None), (7, 0.1), (0, 1.0)])
def test_act_grad_mul_vs_numpy(act, slope):
    """out = g * act'(y) from the post-activation value (relu / Keras leaky_relu / LeakyReLU(0.1) / linear)."""
    lib, L = _lib()
    rng = np.random.default_rng(act)
    n = 10007
    g, y = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    if slope is None:
        import dib_oracle as orc
        slope = float(orc._act_grad_from_output("leaky_relu", np.array([-1.0]))[0])
    want = g * np.where(y > 0, 1.0, slope)
    gd, yd, out = torch.from_numpy(g).cuda(), torch.from_numpy(y).cuda(), torch.empty(n, device="cuda")
    L.check(lib.dib_act_grad_mul(_ptr(gd), _ptr(yd), act, n, _ptr(out), _stream()), "dib_act_grad_mul")
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - want).max() <= 2e-7 * np.abs(want).max()   # (the slope is a float32 constant on the device)


def test_sgd_step_vs_numpy():
    """tf.keras.optimizers.SGD (train.py:128: any Keras optimizer name): theta -= lr * grad_scale * g, lr a device scalar."""
    lib, L = _lib()
    rng = np.random.default_rng(3)
    n = 4099
    p, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    pd, gd, lr = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda(), torch.full((1,), 0.05, device="cuda")
    L.check(lib.dib_sgd_step(_ptr(pd), _ptr(gd), n, _ptr(lr), 0.5, _stream()), "dib_sgd_step")
    torch.cuda.synchronize()
    want = p.astype(np.float64) - 0.05 * 0.5 * g
    assert np.abs(pd.cpu().numpy() - want).max() < 1e-6


@pytest.mark.parametrize("mode,M,N,K,tuning", [(0, 1000, 200, 50, ("fwd_narrow_wgs", 0)),    # forward on 64 x 128 tiles
                                               (0, 66000, 64, 16, None), (1, 66000, 48, 24, None)])   # 128 x 64 tiles
def test_gemm_tile_shapes_the_default_rules_rarely_pick(mode, M, N, K, tuning):
    """dib_gemm_grouped (C = relu(A @ B + bias) / C = (A @ B^T) * [aux > 0]) on the tile shapes that only extreme aspect ratios
    or a tuning key select, vs NumPy float64."""
    from dib_amd._gemm_plan import _Gemm, _d
    lib, L = _lib()
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((K, N) if mode == 0 else (N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    aux = rng.standard_normal((M, N)).astype(np.float32)
    if mode == 0:
        ref = np.maximum(A.astype(np.float64) @ Bm.astype(np.float64) + bias, 0)
    else:
        ref = (A.astype(np.float64) @ Bm.astype(np.float64).T) * (aux > 0)
    Ad, Bd, Cd = torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda(), torch.full((M, N), float("nan"), device="cuda")
    bd, xd = torch.from_numpy(bias).cuda(), torch.from_numpy(aux).cuda()
    g = _Gemm(mode, [_d(0, K, 0, N if mode == 0 else K, 0, N, M, N, K, bias_off=0 if mode == 0 else -1, aux_off=0, ldaux=N)],
              Ad, Bd, Cd, bias=bd if mode == 0 else None, aux=xd if mode == 1 else None, act=1)
    g.upload(torch.device("cuda"))
    old = L.get_tuning(tuning[0]) if tuning else None
    n0 = lib.dib_launch_count()
    try:
        if tuning:
            L.set_tuning(*tuning)
        g.run(lib, _stream())
        torch.cuda.synchronize()
    finally:
        if tuning:
            L.set_tuning(tuning[0], old)
    assert lib.dib_launch_count() == n0 + 1
    assert np.abs(Cd.cpu().numpy() - ref).max() < 2e-5 * (1 + np.abs(ref).max())
