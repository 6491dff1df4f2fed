"""Phase timing of the small-batch row-tile kernels (csrc/dib_small.h), diagnostic: needs a -DDIB_SMALL_TIMING build of the
library (bash tools/build_variant.sh STIMING -DDIB_SMALL_TIMING; DIB_LIB_PATH=exp/lib_STIMING.so python tools/small_phase_timing.py).
Workgroup (0, 0) marks the 100 MHz wall clock at each phase boundary; printed in microseconds for a training step and a
validation step of the reference's default Boolean-circuit layout at B = 128 (train.py:30-44)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dib_amd import _lib
    from dib_amd.engine import HipEngine
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    B = int(nums[0]) if nums else 128
    for kv in [a for a in sys.argv[1:] if "=" in a]:   # dib_set_tuning keys, e.g. int_cluster=0
        k, v = kv.split("=")
        _lib.set_tuning(k, int(v))
    eng = HipEngine([1] * 10, [128, 128], [256, 256], 1, feature_embedding_dimension=32)
    rng = np.random.default_rng(0)
    x = eng.to_device(rng.standard_normal((B, 10)).astype(np.float32))
    y = eng.to_device((rng.random((B, 1)) > 0.5).astype(np.float32))
    read = eng.lib.dib_small_debug_read
    out = (ctypes.c_longlong * 64)()

    def marks():
        assert read(out) == 0
        return np.array(list(out), dtype=np.float64) / 100.0   # microseconds

    for it in range(5):
        eng.train_step(x, y, None, 0, B, 1, it, "bce_logits", optimizer=("adam", 0.9, 0.999, 1e-7))
    t = marks()
    names1 = ["zero-fill + gather + posenc", "(weights ptrs)", "layer 1", "layer 2", "layer 3", "reparam + KL"]
    print("training step, encoder forward (us):", {n: round(t[i + 1] - t[i], 2) for i, n in enumerate(names1) if i + 1 <= 5 and i != 1}, "total", round(t[5] - t[0], 2))
    print("training step, integration kernel (us): load u %.2f | fwd L1 %.2f | fwd L2 %.2f | head %.2f | dgrad L2 %.2f | dgrad -> g_u %.2f | total %.2f" % (
        t[17] - t[16], t[18] - t[17], t[19] - t[18], t[22] - t[19], t[25] - t[23], t[28] - t[25], t[28] - t[16]))
    if t[30] > t[17]:   # cluster mode: [slice computed | exchange + reload] per layer
        print("  cluster mode: fwd L1 %.2f + %.2f | fwd L2 %.2f + %.2f | dgrad L2 %.2f + %.2f" % (
            t[30] - t[17], t[18] - t[30], t[31] - t[18], t[19] - t[31], t[33] - t[23], t[25] - t[33]))
    if t[30] <= t[17]:   # (one workgroup per tile: the marks inside the layer primitive belong to the ENCODER forward's three layers)
        for nm, b in (("enc fwd L1", 6), ("enc fwd L2", 45), ("enc fwd L3", 34)):
            print("    inside %s (wave 0): issue %.2f | loads + MFMAs %.2f | partials + barrier %.2f | reduce + epilogue %.2f | barrier %.2f" % (
                nm, t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], t[b + 4] - t[b + 3], t[b + 5] - t[b + 4]))
    if t[30] > t[17]:
        for nm, b in (("fwd L1", 6), ("fwd L2", 45), ("dgrad L2", 34), ("dgrad g_u", 51)):
            print("    inside %s (wave 0): issue %.2f | loads + MFMAs %.2f | partials + barrier %.2f | reduce + epilogue %.2f | barrier %.2f" % (
                nm, t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], t[b + 4] - t[b + 3], t[b + 5] - t[b + 4]))
    print("training step, encoder backward (us): loads + d(mu|logvar) %.2f | dgrad L3 %.2f | dgrad L2 %.2f | dW1 partial %.2f | total %.2f" % (
        t[41] - t[40], t[42] - t[41], t[43] - t[42], t[44] - t[43], t[44] - t[40]))
    print("gaps (us): enc fwd end -> integration start %.2f ; integration end -> enc bwd start %.2f" % (t[16] - t[5], t[40] - t[28]))
    for it in range(3):
        eng.eval_step(x, y, None, 0, B, 1, 100 + it, "bce_logits")
    t = marks()
    print("validation step, encoder forward total %.2f ; integration: load u %.2f | fwd L1 %.2f | fwd L2 %.2f | head %.2f | total %.2f" % (
        t[5] - t[0], t[17] - t[16], t[18] - t[17], t[19] - t[18], t[22] - t[19], t[22] - t[16]))


def infonce_main():
    """phase marks [56, 64) of dib_infonce_small_kernel at the reference's default batch (B = 128, shared space 64, l2)"""
    from dib_amd.engine import HipEngine
    B, D = 128, 64
    eng = HipEngine([1, 1], [], [], 1, feature_embedding_dimension=4)
    rng = np.random.default_rng(0)
    a = rng.standard_normal((B, D)).astype(np.float32)
    b = (a + 0.7 * rng.standard_normal((B, D))).astype(np.float32)
    ad, bd = eng.to_device(a), eng.to_device(b)
    read = eng.lib.dib_small_debug_read
    out = (ctypes.c_longlong * 64)()
    names = ["stage X, Y + norms", "S = sim(X Y^T)", "row / column lse", "loss", "coefficients + R", "C . Other (MFMA)", "epilogue"]
    for it in range(4):
        junk = torch.randn(1 << 22, device=eng.device).sum()   # other work between the launches, as in a training step
        eng.infonce(ad, bd, "l2", 4.0)
        assert read(out) == 0
        t = np.array(list(out), dtype=np.float64) / 100.0
        print("dib_infonce_small_kernel (us):", {n: round(t[57 + i] - t[56 + i], 2) for i, n in enumerate(names)}, "total", round(t[63] - t[56], 2))


if __name__ == "__main__":
    infonce_main() if "infonce" in sys.argv[1:] else main()
