"""Kernel time of dib_infonce_fwd_bwd (symmetric InfoNCE loss + embedding gradients over the in-batch [B, B] similarity,
reference train.py:203-215) at the batch sizes the path runs: python tools/infonce_bench.py [--dims 64]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs="+", default=[8, 64])
    ap.add_argument("--batches", type=int, nargs="+", default=[128, 2048])
    a = ap.parse_args()
    from dib_amd.engine import HipEngine
    eng = HipEngine([1, 1], [32, 32], [16], 1, feature_embedding_dimension=32)
    rng = np.random.default_rng(0)
    for B in a.batches:
        for D in a.dims:
            x = eng.to_device(rng.standard_normal((B, D)).astype(np.float32))
            y = eng.to_device((x.cpu().numpy() + 0.7 * rng.standard_normal((B, D))).astype(np.float32))
            for kind in ("l2sq", "l2", "l1", "linf", "cosine"):
                for _ in range(2):
                    eng.infonce(x, y, kind, 1.0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    eng.infonce(x, y, kind, 1.0)
                e1.record()
                torch.cuda.synchronize()
                print(json.dumps({"B": B, "D": D, "similarity": kind, "ms": round(e0.elapsed_time(e1) / 5, 4)}))


if __name__ == "__main__":
    main()
