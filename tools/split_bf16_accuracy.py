"""CPU study for the split-bf16 option: how accurate is an fp32 product emulated with 3 or 6
bf16 x bf16 -> fp32 products, on the shapes of the north-star encoder (K = 128 contraction, weights ~ glorot, activations
~ relu outputs)?  Pure NumPy: bf16 = fp32 rounded to 8 significand bits (round-to-nearest-even); partial products are
exact in fp32 (8 x 8 bits), accumulation in float32 like the MFMA accumulator.
    python tools/split_bf16_accuracy.py
"""
import numpy as np


def to_bf16(x):
    """fp32 -> nearest bf16 (returned as fp32 with the low 16 bits cleared)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    hi = to_bf16(x)
    mid = to_bf16(x - hi)
    lo = to_bf16(x - hi - mid)
    return hi, mid, lo


def mm32(a, b):
    return np.matmul(a.astype(np.float32), b.astype(np.float32))  # fp32 accumulate


def emulate(a, b, terms):
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    pairs = {1: [(ah, bh)],
             3: [(ah, bh), (ah, bm), (am, bh)],
             6: [(ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)],
             9: [(x, y) for x in (ah, am, al) for y in (bh, bm, bl)]}[terms]
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for x, y in reversed(pairs):  # small terms first
        acc = acc + mm32(x, y)
    return acc


def main():
    rng = np.random.default_rng(0)
    B, K, N = 4096, 128, 128
    a = np.maximum(rng.standard_normal((B, K)), 0).astype(np.float32)                      # relu activations
    w = rng.uniform(-1, 1, (K, N)).astype(np.float32) * np.float32(np.sqrt(6.0 / (K + N)))    # glorot-uniform
    ref = a.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(ref).max()
    rows = [("fp32 (np.matmul, fp32 accumulate)", mm32(a, w))] + [(f"bf16 x{t}", emulate(a, w, t)) for t in (1, 3, 6, 9)]
    print(f"[B,K]x[K,N] = [{B},{K}]x[{K},{N}], max|ref| = {scale:.3f}")
    for name, got in rows:
        err = np.abs(got.astype(np.float64) - ref)
        print(f"{name:36s} max abs err {err.max():.3e}   max err / max|ref| {err.max() / scale:.3e}   rms {np.sqrt((err ** 2).mean()):.3e}")
    # wgrad-like contraction over the batch (K = 65536): accumulation length matters more than the split
    Bk = 65536
    h = np.maximum(rng.standard_normal((Bk, 64)), 0).astype(np.float32)
    g = (rng.standard_normal((Bk, 64)) * 1e-3).astype(np.float32)
    ref = h.astype(np.float64).T @ g.astype(np.float64)
    scale = np.abs(ref).max()
    print(f"wgrad-like [64,{Bk}]x[{Bk},64], max|ref| = {scale:.3e}")
    for name, got in [("fp32", mm32(h.T, g))] + [(f"bf16 x{t}", emulate(h.T.copy(), g, t)) for t in (3, 6)]:
        err = np.abs(got.astype(np.float64) - ref)
        print(f"{name:36s} max abs err {err.max():.3e}   max err / max|ref| {err.max() / scale:.3e}")


if __name__ == "__main__":
    main()
