"""Generates tests/golden/ctw_golden.npz from the REFERENCE's own CTW source (compiled where it lies by
oracle/Makefile into oracle/_ref/libctw_ref.so).  Run here, where /root/reference exists:
    make -C oracle && python tests/golden/make_golden_ctw.py
The fixture travels to machines without the reference (GPU box)."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libctw_ref.so"))
ref.ref_ctw_estimate_entropy.restype = ctypes.c_double
ref.ref_ctw_estimate_entropy.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int]


def ref_rate(seq, alphabet):
    s = np.ascontiguousarray(seq, dtype=np.int8)
    return ref.ref_ctw_estimate_entropy(s.ctypes.data, s.size, alphabet)


def cases():
    rng = np.random.default_rng(20241008)
    out = []
    for a in (1, 2, 3, 4, 7, 16, 100):
        for n in (1, 2, 3, 5, 17, 100, 700, 3000):
            out.append((rng.integers(0, a, n), a))
    out += [(np.tile([0, 1, 2], 300), 3), (np.zeros(2000, int), 2), (np.tile([0, 0, 1, 0, 1, 1, 1], 200), 2),
            (np.tile(rng.integers(0, 2, 37), 40), 2),                 # long period: deep contexts, MAX_DEPTH cut-off
            ((rng.random(4000) < 0.1).astype(int), 2)]
    x, xs = 0.3, []
    for _ in range(4000):                                             # logistic map r = 3.7115 (chaos_data.py default)
        x = 3.7115 * x * (1 - x)
        xs.append(x)
    xs = np.array(xs)
    out += [((xs > 0.5).astype(int), 2), (np.digitize(xs, [0.3, 0.6, 0.8]), 4)]
    out += [(rng.integers(0, 4, 200000), 4)]                          # one large case (product only; oracle skips it)
    return out


if __name__ == "__main__":
    cs = cases()
    flat = np.concatenate([np.asarray(s, dtype=np.int8) for s, _ in cs])
    offsets = np.concatenate([[0], np.cumsum([len(s) for s, _ in cs])]).astype(np.int64)
    alph = np.array([a for _, a in cs], dtype=np.int32)
    rates = np.array([ref_rate(s, a) for s, a in cs], dtype=np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ctw_golden.npz"), symbols=flat, offsets=offsets,
                        alphabet=alph, rate=rates)
    print(len(cs), "cases;", "rates[:5] =", rates[:5])
