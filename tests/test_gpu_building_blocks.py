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


@pytest.mark.parametrize("M,N,K,groups,nsplit", [(32, 1536, 1600, 3, 4), (32, 256, 100, 1, 1), (20, 300, 777, 2, 3)])
def test_weight_gradient_flat_tile_vs_numpy(M, N, K, groups, nsplit):
    """dib_gemm_grouped mode 2 (C = A^T @ B over the rows, + column sums of B) for a <= 32-row operand against a wide one - the set
    transformer's q / k / v kernels [32, heads x key_dim] - on the 32 x 256 tile (dib_gemm_kernel<2,1,2,32,true>) and, with
    dib_set_tuning("wgrad_flat_tile", 0), on the 64 x 128 tile: split slabs summed in order vs NumPy float64."""
    from dib_amd._gemm_plan import _Gemm, _d
    lib, L = _lib()
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((groups, K, M)).astype(np.float32)
    Bm = rng.standard_normal((groups, K, N)).astype(np.float32)
    ref = np.einsum("gkm,gkn->gmn", A.astype(np.float64), Bm.astype(np.float64))
    ref_b = Bm.astype(np.float64).sum(1)
    Ad, Bd = torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda()
    per = (M * N + N + 3) // 4 * 4                      # one group's [kernel | bias] block
    stride = groups * per
    rps = ((K + nsplit - 1) // nsplit + 31) // 32 * 32
    descs = [_d(g * K * M, M, g * K * N, N, g * per, N, M, N, K, bias_off=g * per + M * N) for g in range(groups)]
    outs = []
    try:
        for flat in (1, 0):
            L.set_tuning("wgrad_flat_tile", flat)
            slabs = torch.full((nsplit * stride,), float("nan"), device="cuda")
            gm = _Gemm(2, descs, Ad, Bd, slabs, bias_out=slabs, nsplit=nsplit, rows_per_split=rps, split_stride=stride)
            gm.upload(torch.device("cuda"))
            gm.run(lib, _stream())
            torch.cuda.synchronize()
            tot = slabs.view(nsplit, stride).double().sum(0).cpu().numpy()
            outs.append(tot)
            for g in range(groups):
                got = tot[g * per: g * per + M * N].reshape(M, N)
                assert np.abs(got - ref[g]).max() < 2e-5 * (1 + np.abs(ref[g]).max()), (flat, g)
                got_b = tot[g * per + M * N: g * per + M * N + N]
                assert np.abs(got_b - ref_b[g]).max() < 2e-5 * (1 + np.abs(ref_b[g]).max()), (flat, g, "column sums")
    finally:
        L.set_tuning("wgrad_flat_tile", 1)


@pytest.mark.parametrize("B,P,H", [(2, 50, 3), (1, 64, 2), (3, 1, 1), (2, 33, 12)])
def test_attention_forward_with_projections_vs_float64(B, P, H):
    """dib_attention_fwd_proj (<= 64 particles): tf.keras.layers.MultiHeadAttention(H, 128)(x, x, x) up to the output projection -
    q / k / v = x @ W_i + b_i per head computed INSIDE the attention launch and written for the backward, softmax(q k^T / sqrt(128)) v,
    the per-query log-sum-exp - against NumPy float64, and against the two-launch path (projection GEMM + dib_attention_fwd)."""
    lib, L = _lib()
    rng = np.random.default_rng(B * 100 + P + H)
    D, K = 32, 128
    HK, T = H * K, B * P
    assert lib.dib_attention_fwd_proj_supported(P, K, D) == 1 and lib.dib_attention_fwd_proj_supported(65, K, D) == 0
    x = rng.standard_normal((T, D)).astype(np.float32)
    # one flat parameter buffer: [W_q | b_q | W_k | b_k | W_v | b_v], 16-byte aligned blocks
    w = [(rng.standard_normal((D, HK)) / np.sqrt(D)).astype(np.float32) for _ in range(3)]
    bvec = [(0.1 * rng.standard_normal(HK)).astype(np.float32) for _ in range(3)]
    flat, w_off, b_off = [], [], []
    for wi, bi in zip(w, bvec):
        w_off.append(sum(len(f) for f in flat)); flat.append(wi.reshape(-1))
        b_off.append(sum(len(f) for f in flat)); flat.append(bi)
    params = torch.from_numpy(np.concatenate(flat)).cuda()
    xd = torch.from_numpy(x).cuda()
    q, k, v, o = (torch.full((T, HK), float("nan"), device="cuda") for _ in range(4))
    lse = torch.empty(B * H * P, device="cuda")
    scale = 1.0 / np.sqrt(K)
    L.check(lib.dib_attention_fwd_proj(_ptr(xd), D, _ptr(params), (ctypes.c_int64 * 3)(*w_off), (ctypes.c_int64 * 3)(*b_off), B, P, H, K, D,
                                       HK, scale, _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _stream()), "dib_attention_fwd_proj")
    torch.cuda.synchronize()
    x64 = x.astype(np.float64)
    qr, kr, vr = (x64 @ wi.astype(np.float64) + bi for wi, bi in zip(w, bvec))
    for got, ref, name in ((q, qr, "q"), (k, kr, "k"), (v, vr, "v")):
        assert np.abs(got.cpu().numpy() - ref).max() < 2e-5 * (1 + np.abs(ref).max()), name
    o_ref, lse_ref = np.zeros((T, HK)), np.zeros((B, H, P))
    for b in range(B):
        for hh in range(H):
            sl, cs = slice(b * P, (b + 1) * P), slice(hh * K, (hh + 1) * K)
            s = scale * qr[sl, cs] @ kr[sl, cs].T
            m = s.max(1, keepdims=True)
            e = np.exp(s - m)
            o_ref[sl, cs] = (e / e.sum(1, keepdims=True)) @ vr[sl, cs]
            lse_ref[b, hh] = (m + np.log(e.sum(1, keepdims=True)))[:, 0]
    assert np.abs(o.cpu().numpy() - o_ref).max() < 3e-5 * (1 + np.abs(o_ref).max())
    assert np.abs(lse.cpu().numpy().reshape(B, H, P) - lse_ref).max() < 3e-5 * (1 + np.abs(lse_ref).max())
    # the two-launch path on the kernel's own q / k / v: same attention
    o2, lse2 = torch.empty_like(o), torch.empty_like(lse)
    L.check(lib.dib_attention_fwd(_ptr(q), _ptr(k), _ptr(v), B, P, H, K, HK, scale, _ptr(o2), _ptr(lse2), None, _stream()), "dib_attention_fwd")
    torch.cuda.synchronize()
    assert (o2 - o).abs().max().item() < 1e-5 * (1 + o.abs().max().item()) and (lse2 - lse).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,P,H", [(2, 50, 3), (1, 64, 2), (3, 1, 1), (2, 33, 12)])
def test_attention_backward_with_projection_gradient_vs_float64(B, P, H):
    """dib_attention_bwd_proj (<= 64 particles): dq / dk / dv bit-identical to dib_attention_bwd's 8-wave kernel, and slab 1 + head
    of dx = dq_h W_q[:, head]^T + dk_h W_k[:, head]^T + dv_h W_v[:, head]^T against NumPy float64 on the kernel's own dq / dk / dv;
    slab 0 is left alone."""
    lib, L = _lib()
    rng = np.random.default_rng(B * 10 + P + H)
    D, K = 32, 128
    HK, T = H * K, B * P
    q, k, v, d_o = (torch.from_numpy(rng.standard_normal((T, HK)).astype(np.float32)).cuda() for _ in range(4))
    w = [(rng.standard_normal((D, HK)) / np.sqrt(D)).astype(np.float32) for _ in range(3)]
    params = torch.from_numpy(np.concatenate([wi.reshape(-1) for wi in w])).cuda()
    w_off = (ctypes.c_int64 * 3)(0, D * HK, 2 * D * HK)
    scale = 1.0 / np.sqrt(K)
    o, lse = torch.empty(T, HK, device="cuda"), torch.empty(B * H * P, device="cuda")
    L.check(lib.dib_attention_fwd(_ptr(q), _ptr(k), _ptr(v), B, P, H, K, HK, scale, _ptr(o), _ptr(lse), None, _stream()), "dib_attention_fwd")
    ws = torch.zeros(int(lib.dib_attention_bwd_workspace_bytes(B, P, H)) // 4 + 4, device="cuda")
    ref = [torch.empty(T, HK, device="cuda") for _ in range(3)]
    L.check(lib.dib_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(d_o), _ptr(lse), None, B, P, H, K, HK, scale, _ptr(ref[0]),
                                  _ptr(ref[1]), _ptr(ref[2]), _ptr(ws), _stream()), "dib_attention_bwd")
    got = [torch.full((T, HK), float("nan"), device="cuda") for _ in range(3)]
    stride = T * D + 8
    dx = torch.full(((1 + H) * stride,), 7.0, device="cuda")
    L.check(lib.dib_attention_bwd_proj(_ptr(q), _ptr(k), _ptr(v), _ptr(d_o), _ptr(lse), B, P, H, K, D, HK, scale, _ptr(got[0]), _ptr(got[1]),
                                       _ptr(got[2]), _ptr(params), w_off, _ptr(dx), stride, _stream()), "dib_attention_bwd_proj")
    torch.cuda.synchronize()
    for a_, b_, nm in zip(got, ref, "qkv"):
        assert torch.equal(a_, b_), f"d{nm}"
    dxn = dx.view(1 + H, stride)[:, :T * D].reshape(1 + H, T, D).cpu().numpy()
    assert (dxn[0] == 7.0).all()
    g64 = [g.cpu().numpy().astype(np.float64) for g in got]
    for hh in range(H):
        cs = slice(hh * K, (hh + 1) * K)
        want = sum(g64[i][:, cs] @ w[i].astype(np.float64)[:, cs].T for i in range(3))
        assert np.abs(dxn[1 + hh] - want).max() < 3e-5 * (1 + np.abs(want).max()), hh
