"""Worker of tests/test_gpu_concurrency.py (fresh process: the FIRST launches of every kernel - the ones that set per-device
kernel attributes - happen while two host threads are inside the library).

include/dib_hip.h "Threads": entry points are thread-safe across distinct (device, stream, workspace) triples.  Two host threads,
each on its own HIP stream with its own parameter / gradient / Adam buffers and workspace, step concurrently; afterwards the same
work runs on one thread, one engine after the other.  Every buffer must hold the same BITS.

  mode "two_layouts":  thread 0 = a large-batch layout (fused encoder-bank kernels, split-batch weight gradients),
                       thread 1 = the reference-default layout at B = 128 (row-tile kernels) - different kernels in flight
  mode "same_arch":    both threads the same architecture and batch (two layouts): the same kernels and the same function-local
                       attribute flags from both threads
  mode "shared_layout": ONE dib_layout used by both threads with distinct workspaces and buffers
  mode "four_small":   FOUR threads and streams of the reference-default step: its integration kernel runs in cluster mode (8 workgroups
                       per row tile that wait for each other inside the kernel) - four such launches in flight at once
"""
import sys
import threading

import numpy as np
import torch

import dib_oracle as orc
from _helpers import spec_kwargs


def _work(kind):
    if kind == "large":
        return orc.DIBSpec([1] * 8, [128, 128], [256, 256], 1, feature_embedding_dimension=32), 4096 + 37
    return orc.DIBSpec([1] * 10, [128, 128], [256, 256], 1, feature_embedding_dimension=32), 128


def _data(spec, B, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, sum(spec.feature_dimensionalities))).astype(np.float32)
    y = (rng.random((B, 1)) > 0.5).astype(np.float32)
    return x, y


def _run(eng, x, y, B, steps, seed):
    xd, yd = eng.to_device(x), eng.to_device(y)
    eng.set_beta(0.03)
    eng.set_lr(1e-3)
    for s in range(steps):
        eng.train_step(xd, yd, None, 0, B, seed, s, "bce_logits", optimizer=("adam", 0.9, 0.999, 1e-7))
    eng.eval_step(xd, yd, None, 0, B, seed, 1000, "bce_logits")
    torch.cuda.current_stream().synchronize()
    return [t.clone() for t in (eng.params, eng.grads, eng.adam_m, eng.adam_v, eng.metrics_acc, eng.step_out(B), eng.pred(B))]


def main(mode, steps=6):
    from dib_amd.engine import HipEngine
    kinds = {"two_layouts": ("large", "small"), "same_arch": ("small", "small"), "shared_layout": ("large", "large"),
             "four_small": ("small",) * 4}[mode]   # four streams of clustered row-tile steps at once (the header's co-residency note)
    works = [_work(k) for k in kinds]
    data = [_data(spec, B, 5 + i) for i, (spec, B) in enumerate(works)]
    out = [None] * len(kinds)
    err = []
    barrier = threading.Barrier(len(kinds))
    shared = {}
    restore = []

    def thread(i):
        try:
            spec, B = works[i]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                eng = HipEngine(**spec_kwargs(spec), init_seed=3 + i)     # layout create + table upload on this thread's stream
                if mode == "shared_layout":
                    stream.synchronize()
                    barrier.wait()
                    if i == 0:
                        shared["layout"] = eng.layout
                    barrier.wait()
                    if i == 1:                                            # thread 1 borrows thread 0's layout; its own is kept for the end
                        restore.append((eng, eng.layout))
                        eng.layout = shared["layout"]
                barrier.wait()
                out[i] = _run(eng, *data[i], B, steps, seed=11 + i)
                barrier.wait()
                shared[i] = eng                                           # keep the engines (and the shared layout) alive until the end
        except Exception as e:   # noqa: BLE001
            err.append(repr(e))
            barrier.abort()

    ts = [threading.Thread(target=thread, args=(i,)) for i in range(len(kinds))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for eng, own in restore:
        eng.layout = own
    torch.cuda.synchronize()
    launches = shared[0].lib.dib_launch_count()
    # the same work, one engine after the other on the default stream
    for i, (spec, B) in enumerate(works):
        eng = HipEngine(**spec_kwargs(spec), init_seed=3 + i)
        ref = _run(eng, *data[i], B, steps, seed=11 + i)
        for a, b in zip(out[i], ref):
            assert torch.isfinite(b).all()
            assert torch.equal(a, b), (mode, i, float((a - b).abs().max()))
    serial_launches = shared[0].lib.dib_launch_count() - launches
    assert serial_launches == launches, (launches, serial_launches)      # the relaxed atomic counter lost no increment
    print("CONCURRENCY_OK", mode, launches)


if __name__ == "__main__":
    main(sys.argv[1])
