"""Cluster mode of the row-tile integration kernel (csrc/dib_small.h; dib_set_tuning "int_cluster" / "int_cluster_wgs"): training-step
and validation-step time of the reference's default layout (train.py:36-44: 10 features, encoders [128, 128], integration [256, 256],
embedding 32) by batch size and REQUESTED workgroups per row tile (the library halves the request until row tiles x cluster <= 256),
same process, interleaved repeats."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dib_amd import _lib
    from dib_amd.engine import HipEngine
    F = int(os.environ.get("DIB_SWEEP_F", "10"))
    eng = HipEngine([1] * F, [128, 128], [256, 256], 1, feature_embedding_dimension=32)
    _lib.set_tuning("int_cluster_wgs", 256)
    _lib.set_tuning("int_cluster_min_weights", 0)
    rng = np.random.default_rng(0)
    for B in (32, 64, 128, 256, 512, 1024, 2048):
        x = eng.to_device(rng.standard_normal((B, F)).astype(np.float32))
        y = eng.to_device((rng.random((B, 1)) > 0.5).astype(np.float32))
        res = {}
        for rep in range(3):
            for cl in (0, 2, 4, 8):
                if (B + 15) // 16 * F > 512:   # beyond the row-tile regime
                    continue
                _lib.set_tuning("int_cluster", cl)
                for kind in ("train", "val"):
                    def step(i):
                        if kind == "train":
                            eng.train_step(x, y, None, 0, B, 1, i, "bce_logits", optimizer=("adam", 0.9, 0.999, 1e-7))
                        else:
                            eng.eval_step(x, y, None, 0, B, 1, i, "bce_logits")
                    for i in range(20):
                        step(i)
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    n = 400
                    for i in range(n):
                        step(i)
                    torch.cuda.synchronize()
                    res.setdefault((cl, kind), []).append((time.perf_counter() - t) / n * 1e6)
        print(f"F={F} B={B:5d}  us/step (min of 3)  " + "  ".join(
            f"cl={cl}: train {min(res[(cl, 'train')]):6.1f} val {min(res[(cl, 'val')]):6.1f}" for cl in (0, 2, 4, 8) if (cl, "train") in res), flush=True)


if __name__ == "__main__":
    main()
