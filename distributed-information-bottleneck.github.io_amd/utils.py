"""Host-side analysis helpers mirroring the reference's `utils.py` names that the hot path's callbacks
touch (reference utils.py:177-262).  The Bhattacharyya / KL matrices use the closed form for
diagonal Gaussians instead of the reference's dense [N,M,d,d] diagonal tensors (same values, see
tests/test_oracle_golden.py); the on-device version is dib_bhattacharyya (csrc/dib_elementwise.h).
"""
from __future__ import annotations

import numpy as np


def bhattacharyya_dist_mat(mus1, logvars1, mus2, logvars2):
    """[N, M] Bhattacharyya distances between diagonal Gaussians (reference utils.py:177-212)."""
    mu1 = np.asarray(mus1, dtype=np.float64)[:, None, :]
    mu2 = np.asarray(mus2, dtype=np.float64)[None, :, :]
    lv1 = np.asarray(logvars1, dtype=np.float64)[:, None, :]
    lv2 = np.asarray(logvars2, dtype=np.float64)[None, :, :]
    assert mu1.shape[-1] == mu2.shape[-1]
    sbar = 0.5 * (np.exp(lv1) + np.exp(lv2))
    term1 = 0.125 * np.sum((mu1 - mu2) ** 2 / sbar, axis=-1)
    term2 = 0.5 * (np.sum(np.log(sbar), axis=-1) - 0.5 * (np.sum(lv1, -1) + np.sum(lv2, -1)))
    return term1 + term2


def kl_divergence_mat(mus1, logvars1, mus2, logvars2):
    """[N, M] KL(N1 || N2) (reference utils.py:214-246)."""
    mu1 = np.asarray(mus1, dtype=np.float64)[:, None, :]
    mu2 = np.asarray(mus2, dtype=np.float64)[None, :, :]
    lv1 = np.asarray(logvars1, dtype=np.float64)[:, None, :]
    lv2 = np.asarray(logvars2, dtype=np.float64)[None, :, :]
    d = mu1.shape[-1]
    return 0.5 * (np.sum(lv2, -1) - np.sum(lv1, -1) - d + np.sum(np.exp(lv1 - lv2), -1)
                  + np.sum((mu2 - mu1) ** 2 * np.exp(-lv2), -1))


def compute_entropy_bits(probability_arr):
    """reference utils.py:248-249."""
    p = np.asarray(probability_arr, dtype=np.float64)
    return -np.sum(p * np.log2(np.where(p > 0, p, 1)))


def compute_entropy(seq):
    """reference utils.py:257-261: empirical entropy (bits) of a symbol sequence."""
    _, counts = np.unique(seq, return_counts=True)
    p = counts / np.sum(counts)
    return -np.sum(p * np.log2(p))


def entropy_rate_scaling_ansatz(N, h_inf, gamma, c):
    """reference utils.py:251-254 (Schurmann & Grassberger 1995)."""
    return h_inf + np.log2(N) / (N ** gamma) / np.abs(c)


def estimate_mi_sandwich_bounds(encoder, dataset, evaluation_batch_size=1024, number_evaluation_batches=8, seed=0):
    """Lower (InfoNCE) and upper (leave-one-out) bounds, in nats, on the information transmitted by one feature
    encoder (reference utils.py:10-73; Poole et al. 2019).

    `encoder` is `model.feature_encoders[f]`; `dataset` is that feature's data, array-like [N, d_f] (the reference
    takes a tf.data.Dataset and draws `number_evaluation_batches` shuffled batches with wrap-around, utils.py:68-71;
    here the batches are drawn with a seeded numpy generator).  The N x N x E pairwise Gaussian log-densities are
    evaluated on the GPU in float64 with a log-sum-exp (dib_mi_sandwich_rows), so well separated encodings give
    log N instead of the reference's underflow."""
    x = np.asarray(dataset, dtype=np.float32)
    if x.ndim == 1:
        x = x[:, None]
    n = x.shape[0]
    bs = int(evaluation_batch_size)
    rng = np.random.default_rng(seed)
    model = encoder._model
    eng = model._ensure_engine()
    estimates = []
    for b in range(int(number_evaluation_batches)):
        rows = rng.permutation(n)[:bs] if n >= bs else rng.integers(0, n, bs)
        enc_out = eng.encode_feature(encoder.index, x[rows])
        estimates.append(eng.mi_sandwich_bounds(enc_out, seed, b, encoder.index))
    return np.mean(np.stack(estimates, 0), 0)
