"""Science-level check: DIB on SI circuit (c) of the reference notebook (Boolean_circuits.ipynb:987-992):
Y = AND(XOR(AND(x2,x0), x3), x1); Shapley values [0.096, 0.377, 0.096, 0.242]; H(Y) = 0.811 bits."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import dib_amd
import dib_oracle as orc


def run(epochs_pre=100, epochs_anneal=500, seed=0):
    x, y = orc.boolean_circuit_truth_table([0, 1, 2, 3, [0, 2, 0], [2, 4, 3], [0, 5, 1]], 4)
    x = np.tile(x, (8, 1)).astype(np.float32)
    y = np.tile(y, 8).astype(np.float32)
    m = dib_amd.DistributedIBNet([1, 1, 1, 1], [32, 32], [64, 64], 1, feature_embedding_dimension=8, noise_seed=seed,
                                 init_seed=seed, shuffle_seed=seed)
    opt = dib_amd.optimizers.get("adam"); opt.learning_rate = 3e-3
    m.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-3, 3.0, epochs_pre, epochs_anneal)
    h = m.fit(x, y, epochs=epochs_pre + epochs_anneal, batch_size=64, callbacks=[cb], verbose=False).history
    kl = np.stack([h[f"KL{f}"] for f in range(4)], -1) / np.log(2)
    beta = np.array(h["beta"])
    loss_bits = (np.array(h["loss"]) - beta * kl.sum(-1) * np.log(2)) / np.log(2)
    return kl, loss_bits, np.array(h["accuracy"]), beta


if __name__ == "__main__":
    kl, loss, acc, beta = run()
    for e in (0, 50, 99, 150, 250, 350, 450, 520, 560, 599):
        print(e, f"beta={beta[e]:.4f}", "KL bits", np.round(kl[e], 3), f"loss={loss[e]:.3f} acc={acc[e]:.3f}")
    print("integrated KL over annealing", np.round(kl[100:].sum(0), 1))
