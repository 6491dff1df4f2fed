"""What the tf.data shuffle BUFFER does to BASELINE config 2: the reference's batches come from a
10 000-element buffer over the time-ordered pendulum rows (train.py:226-227, data.py:122-123: trajectory after trajectory), so a
batch's in-batch negatives are a handful of neighbouring trajectories; a whole-dataset permutation (what this project drew until
round 4) gives negatives from all trajectories.  Runs the custom InfoNCE loop at train.py's defaults (B = 128, lr 3e-4, l2,
[128,128] encoders, shared space 64) on 100 simulated trajectories at a FIXED beta with both streams and prints the
training / validation InfoNCE loss per epoch.

    python tools/stream_effect.py [data_dir] [epochs] [beta]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dib_amd
from dib_amd import infonce


def run(d, buffer_size, epochs, beta, seed=0):
    model = dib_amd.DistributedIBNet(d["feature_dimensionalities"], [128, 128], [256, 256], 64, feature_embedding_dimension=32,
                                     noise_seed=seed, init_seed=seed, shuffle_seed=seed)
    t0 = time.time()
    out = infonce.fit_infonce(model, d["x_train"], d["y_train"], d["x_valid"], d["y_valid"], batch_size=128,
                              number_pretraining_epochs=epochs + 1, number_annealing_epochs=1, beta_start=beta, beta_end=beta,
                              learning_rate=3e-4, similarity="l2", temperature=1.0, seed=seed, shuffle_buffer=buffer_size)
    return out, time.time() - t0


if __name__ == "__main__":
    data_dir = sys.argv[1] if len(sys.argv) > 1 else "exp/pendulum100"
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    beta = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
    d = dib_amd.data.fetch_double_pendulum(data_path=data_dir, pendulum_number_trajectories=100, seed=0)
    n = len(d["x_train"])
    print(f"pendulum: {n} training rows (time-ordered, {n // 2400} trajectories), {len(d['x_valid'])} validation rows; B = 128, "
          f"beta = {beta:g} fixed, {epochs} epochs of {n / 128:.0f} steps; log(128) = {np.log(128):.3f} nats is the loss of an "
          "uninformative encoder pair (x 2 for the symmetric loss)")
    for name, buf in (("tf.data shuffle(10 000) over the sequential stream [reference, this round]", 10_000),
                      ("buffer = dataset (all rows reachable at every draw) [rounds 2-4 drew permutations]", n)):
        out, secs = run(d, buf, epochs, beta)
        print(f"{name}: {secs:.1f} s")
        print("   train InfoNCE loss / epoch     :", np.round(out["loss_infonce"], 4).tolist())
        print("   validation InfoNCE loss / epoch:", np.round(out["loss_infonce_validation"], 4).tolist())
        print("   sum KL (nats) / epoch          :", np.round(out["kl_total"], 3).tolist())
