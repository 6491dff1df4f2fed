"""GPU probe: error of the device noise generator (Philox4x32-10 + Box-Muller on hardware transcendentals,
csrc/dib_common.h:dib_eps4) against the fp64 oracle on ~1 M normals.  Tooling only (imports oracle/ as the checker).
Measured on MI355X: mean |diff| 1.1e-7, max 2.7e-5 (one sample with u0 within 2^-24 of 1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dib_amd  # noqa: E402,F401
import dib_oracle as orc  # noqa: E402
from dib_amd.engine import HipEngine  # noqa: E402

eng = HipEngine([1] * 8, [128, 128], [256, 256], 1, device="cuda:0", init_seed=0)
n = 4096
got = eng.eps(None, 0, n, seed=12345, step=3).cpu().numpy()
ref = orc.philox_normal_all(12345, 3, np.arange(n), got.shape[1], got.shape[2])
d = np.abs(got - ref)
print("eps max abs diff", d.max(), "mean", d.mean(), "n", d.size, "max|eps|", np.abs(ref).max())
