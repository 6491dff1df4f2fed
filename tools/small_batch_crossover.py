"""Where do the row-tile kernels (csrc/dib_small.h) stop paying?  Keras-path training step (fused head + Adam in the tail) of the
reference architecture for F features x B rows with dib_set_tuning("small_batch", 1 | 0).  Rows above the library's row-tile
limit need a variant built with -DDIB_SMALL_MAX_BATCH=<rows> (DIB_LIB_PATH=exp/lib_SB2048.so).
usage: python tools/small_batch_crossover.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dib_amd import _lib  # noqa: E402
from dib_amd.engine import HipEngine  # noqa: E402


def step_us(F, B, small, out=1, kind="bce_logits"):
    _lib.set_tuning("small_batch", small)
    eng = HipEngine([1] * F, [128, 128], [256, 256], out, feature_embedding_dimension=32)
    rng = np.random.default_rng(0)
    x = eng.to_device(rng.standard_normal((B, F)).astype(np.float32))
    y = eng.to_device((rng.random((B, out)) > 0.5).astype(np.float32))
    opt = ("adam", 0.9, 0.999, 1e-7)
    for it in range(8):
        eng.train_step(x, y, None, 0, B, 1, it, kind, optimizer=opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for it in range(n):
        eng.train_step(x, y, None, 0, B, 1, it, kind, optimizer=opt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


if __name__ == "__main__":
    print("F     B   tiles*F  row-tile us  large us   ratio")
    for F in (2, 4, 10, 24, 64):
        for B in (128, 256, 512, 1024, 2048):
            s, l = step_us(F, B, 1), step_us(F, B, 0)
            print(f"{F:<4d} {B:5d} {((B + 15) // 16) * F:8d} {s:11.1f} {l:9.1f} {s / l:7.2f}", flush=True)
    _lib.set_tuning("small_batch", 1)
