"""CTW entropy-rate estimator: product (csrc/dib_ctw.cpp through include/dib_ctw.h) and the
Python restatement (oracle/ctw_oracle.py) against golden vectors produced by the reference's own C++ source
(tests/golden/make_golden_ctw.py), and - when the reference build is present - against it live.  CPU only."""
import ctypes
import os
import re
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import dib_amd  # noqa: E402,F401
from dib_amd import ctw  # noqa: E402
import ctw_oracle  # noqa: E402


def _golden():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ctw_golden.npz"))
    off = g["offsets"]
    return [(g["symbols"][off[i]:off[i + 1]], int(g["alphabet"][i]), float(g["rate"][i])) for i in range(len(off) - 1)]


def _same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


def test_product_matches_reference_golden_bit_exact():
    for seq, alphabet, rate in _golden():
        got = ctw.estimate_entropy(seq, alphabet)
        assert _same(got, rate), (alphabet, len(seq), got, rate)


def test_oracle_restatement_matches_reference_golden_bit_exact():
    checked = 0
    for seq, alphabet, rate in _golden():
        if len(seq) > 4000:
            continue  # pure-Python loops: small cases only
        got = ctw_oracle.estimate_entropy(seq, alphabet)
        assert _same(got, rate), (alphabet, len(seq), got, rate)
        checked += 1
    assert checked >= 60


def test_node_count_matches_oracle_tree():
    rng = np.random.default_rng(3)
    for alphabet, n in ((2, 300), (4, 500), (3, 64)):
        s = rng.integers(0, alphabet, n)
        assert ctw.node_count(s, alphabet) == ctw_oracle.count_nodes(s, alphabet)
    per = np.tile([0, 1, 1], 100)
    assert ctw.node_count(per, 2) == ctw_oracle.count_nodes(per, 2)


def test_known_answers():
    assert ctw.estimate_entropy([1], 2) == 1.0                      # one symbol, KT estimator: 1 bit
    assert np.isnan(ctw.estimate_entropy([], 2))                    # 0/0 like the reference
    assert ctw.estimate_entropy(np.zeros(5000, int), 1) == 0.0      # unary alphabet carries no information
    rng = np.random.default_rng(0)
    assert abs(ctw.estimate_entropy(rng.integers(0, 4, 100000), 4) - 2.0) < 5e-3       # iid uniform: log2 |A|
    assert ctw.estimate_entropy(np.tile([0, 1], 5000), 2) < 0.01                       # periodic: rate -> 0
    p = 0.1
    h = -(p * np.log2(p) + (1 - p) * np.log2(1 - p))
    assert abs(ctw.estimate_entropy((rng.random(200000) < p).astype(int), 2) - h) < 5e-3  # Bernoulli(0.1)


def test_batch_equals_singles_and_accepts_2d():
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 3, n) for n in (1, 10, 1000, 0, 333, 5000)]
    singles = np.array([ctw.estimate_entropy(s, 3) for s in seqs])
    for threads in (1, 2, 0):
        got = ctw.estimate_entropy_batch(seqs, 3, threads=threads)
        assert all(_same(a, b) for a, b in zip(got, singles))
    grid = rng.integers(0, 2, (7, 400))
    assert np.array_equal(ctw.estimate_entropy_batch(grid, 2), np.array([ctw.estimate_entropy(r, 2) for r in grid]))
    assert ctw.estimate_entropy_batch([], 2).shape == (0,)


def test_thread_safety_mixed_alphabets():
    """The reference keeps |A| and beta in static members (cppctw.cpp:86-87); concurrent calls with different
    alphabets must not interfere here."""
    rng = np.random.default_rng(9)
    jobs = [(rng.integers(0, a, 20000), a) for a in (2, 5, 3, 16, 2, 7, 4, 9)]
    want = [ctw.estimate_entropy(s, a) for s, a in jobs]
    got = [None] * len(jobs)

    def run(i):
        got[i] = ctw.estimate_entropy(*jobs[i])

    ths = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert got == want


def test_argument_errors():
    with pytest.raises(ValueError):
        ctw.estimate_entropy([0, 1, 2], 2)          # symbol outside the alphabet (the reference indexes out of bounds)
    with pytest.raises(ValueError):
        ctw.estimate_entropy([0, -1], 2)
    with pytest.raises(ValueError):
        ctw.estimate_entropy([0], 0)
    with pytest.raises(ValueError):
        ctw.estimate_entropy([0], 128)
    lib = ctw.load_library()
    out = ctypes.c_double()
    assert lib.dib_ctw_estimate_entropy(None, 3, 2, ctypes.byref(out)) == -1
    assert lib.dib_ctw_estimate_entropy(None, 0, 2, None) == -1
    bad = np.array([0, 3], dtype=np.int8)
    offs = np.array([0, 1, 2], dtype=np.int64)
    rates = np.zeros(2)
    assert lib.dib_ctw_estimate_entropy_batch(bad.ctypes.data, offs.ctypes.data, 2, 2, 1, rates.ctypes.data) == -1
    assert rates[0] == 1.0 and np.isnan(rates[1])   # the valid sequence is still computed


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dib_ctw.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dib_ctw_\w+)\s*\(", hdr))
    assert declared == set(ctw.SIGNATURES), declared ^ set(ctw.SIGNATURES)
    lib = ctw.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert b"dib_ctw" in lib.dib_ctw_version()


def test_against_live_reference_build_when_present():
    """oracle/_ref/libctw_ref.so = the reference's cppctw.cpp compiled where it lies (oracle/Makefile)."""
    path = os.path.join(ROOT, "oracle", "_ref", "libctw_ref.so")
    if not os.path.exists(path):
        pytest.skip("reference build not present (no /root/reference on this machine)")
    ref = ctypes.CDLL(path)
    ref.ref_ctw_estimate_entropy.restype = ctypes.c_double
    ref.ref_ctw_estimate_entropy.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
    rng = np.random.default_rng(77)
    for _ in range(40):
        a = int(rng.integers(1, 20))
        n = int(rng.integers(1, 30000))
        kind = rng.integers(0, 3)
        if kind == 0:
            s = rng.integers(0, a, n)
        elif kind == 1:
            s = np.tile(rng.integers(0, a, int(rng.integers(1, 50))), n // 10 + 1)[:n]      # periodic
        else:
            s = np.minimum((rng.random(n) ** 3 * a).astype(int), a - 1)                     # skewed
        s8 = np.ascontiguousarray(s, dtype=np.int8)
        want = ref.ref_ctw_estimate_entropy(s8.ctypes.data, s8.size, a)
        assert _same(ctw.estimate_entropy(s, a), want), (a, n, kind)


def test_chaos_data_generators_and_entropy_rate_kat():
    """dib_amd.chaos_data mirrors reference chaos/chaos_data.py:3-55; the logistic map at r = 4 with the generating
    partition x > 1/2 is a fair coin (entropy rate exactly 1 bit/symbol) - an end-to-end known answer for
    generator -> symbolisation -> CTW."""
    from dib_amd import chaos_data
    d = chaos_data.generate_data("logistic", number_iterations=1000, number_skip_iterations=10, seed=1)
    assert d.shape == (1000, 1)
    assert np.allclose(d[1:, 0], d[:-1, 0] * (1 - d[:-1, 0]) * 3.7115, rtol=0, atol=0)      # default r (chaos_data.py:21)
    h = chaos_data.generate_data("henon", 500, 20, seed=2)
    assert h.shape == (500, 2) and np.array_equal(h[1:, 1], h[:-1, 0])
    assert np.allclose(h[1:, 0], 1 - 1.4 * h[:-1, 0] ** 2 + 0.3 * h[:-1, 1])
    k = chaos_data.generate_data("ikeda", 300, 20, seed=3)
    assert k.shape == (300, 2) and np.isfinite(k).all()
    assert np.array_equal(chaos_data.generate_data("logistic", 50, 5, seed=7), chaos_data.generate_data("logistic", 50, 5, seed=7))
    with pytest.raises(ValueError):
        chaos_data.generate_data("lorenz")
    traj = chaos_data.generate_data("logistic", 200000, 1000, seed=11, r=4.0)[:, 0]
    rate = ctw.estimate_entropy((traj > 0.5).astype(int), 2)
    assert abs(rate - 1.0) < 1e-2, rate
