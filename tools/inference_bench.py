"""GPU: forward-only throughput of BASELINE config 3 (validation / predict path) with and without DIB_FWD_INFERENCE."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dib_amd  # noqa: E402,F401
from dib_amd.engine import HipEngine  # noqa: E402

F, B = 64, 65536
eng = HipEngine([1] * F, [128, 128], [256, 256], 1, device="cuda:0", init_seed=0)
x = eng.to_device(np.random.default_rng(0).standard_normal((B, F)).astype(np.float32))
for inference in (False, True):
    for i in range(3):
        eng.forward(x, None, 0, B, 0, i, inference=inference)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        eng.forward(x, None, 0, B, 0, i, inference=inference)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"forward inference={inference}: {ms:.3f} ms/batch = {B / ms / 1e3:.2f} M samples/s")
