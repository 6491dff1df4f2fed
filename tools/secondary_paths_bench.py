"""Step / call times of the paths beside the headline at the sizes the reference runs them, one MI355X:
  infonce   the custom InfoNCE training loop (train.py:180-289): 4 features (pendulum layout [2,1,2,1]), Y encoder [128,128],
            shared space 64, similarity l2; batch 128 (train.py default) and 2048 (chaos notebook)
  mi        InfoPerFeatureCallback.on_epoch_end (models.py:188-223): F features x 8 batches of 1024 encodings, E = 32
  compress  SaveCompressionMatricesCallback-style deterministic encode + Bhattacharyya matrix of 1024 points
python tools/secondary_paths_bench.py [infonce|mi|compress ...]   (wrap in rocprofv3 --kernel-trace --stats for the kernels)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def t_infonce(batch):
    import dib_amd
    from dib_amd import infonce
    rng = np.random.default_rng(0)
    n = batch * 16
    x = rng.standard_normal((n, 6)).astype(np.float32)
    y = (x + 0.3 * rng.standard_normal((n, 6))).astype(np.float32)
    model = dib_amd.DistributedIBNet([2, 1, 2, 1], [128, 128], [256, 256], 64, feature_embedding_dimension=32)
    kw = dict(batch_size=batch, number_pretraining_epochs=1, number_annealing_epochs=2, beta_start=1e-4, beta_end=1.0,
              learning_rate=3e-4, shared_dimensionality=64, similarity="l2")
    infonce.fit_infonce(model, x, y, x[:batch * 2], y[:batch * 2], **kw)      # warm-up (3 epochs x 16 steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kw.update(number_pretraining_epochs=2, number_annealing_epochs=4)
    infonce.fit_infonce(model, x, y, x[:batch * 2], y[:batch * 2], **kw)      # 6 epochs x 16 train steps + 6 x 3 validation steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = 5 * 16 + 6 * 3
    return {"path": "infonce fit loop", "batch": batch, "ms_per_step": round(1e3 * dt / steps, 3)}


def t_mi(F):
    import dib_amd
    rng = np.random.default_rng(1)
    xv = rng.standard_normal((8192, F)).astype(np.float32)
    model = dib_amd.DistributedIBNet([1] * F, [128, 128], [256, 256], 1, feature_embedding_dimension=32)
    model._ensure_engine()
    cb = dib_amd.InfoPerFeatureCallback(1, xv, info_bound_batch_size=1024, info_bound_number_batches=8)
    cb.set_model(model) if hasattr(cb, "set_model") else setattr(cb, "model", model)
    cb.on_epoch_end(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cb.on_epoch_end(0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"path": "InfoPerFeatureCallback.on_epoch_end", "features": F, "ms": round(1e3 * dt, 2),
            "ms_per_feature_batch": round(1e3 * dt / (F * 8), 4)}


def main():
    which = sys.argv[1:] or ["infonce", "mi"]
    if "infonce" in which:
        for b in (128, 2048):
            print(json.dumps(t_infonce(b)), flush=True)
    if "mi" in which:
        for F in (10, 64):
            print(json.dumps(t_mi(F)), flush=True)


if __name__ == "__main__":
    main()
