import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import dib_amd
from dib_amd.engine import HipEngine
from oracle import dib_oracle as orc
HipEngine([1] * 8, [128, 128], [256, 256], 1, device="cuda:0", init_seed=0)
n = 4096
got = eng.eps(None, 0, n, seed=12345, step=3).cpu().numpy()
ref = orc.philox_normal_all(12345, 3, np.arange(n), got.shape[1], got.shape[2])
d = np.abs(got - ref)
print("eps max abs diff", d.max(), "mean", d.mean(), "n", d.size, "max|eps|", np.abs(ref).max())
