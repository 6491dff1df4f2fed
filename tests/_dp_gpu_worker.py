"""Worker for tests/test_gpu_dp_and_cache.py: a small fit() on the GPU; under torch.distributed.run it takes the RCCL
data-parallel path of fit() (two-bucket all-reduce, metric all-reduce), otherwise the single-process path."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(out_path, batch=1024):
    import torch.distributed as dist
    distributed = "RANK" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{lr}"))
    import dib_amd
    rng = np.random.default_rng(0)
    n = 2 * batch + 1  # tail batch of ONE row: with > 1 rank some ranks get none (same collectives on every rank)
    x = rng.standard_normal((n, 8)).astype(np.float32)
    y = (x[:, 0] * x[:, 1] > 0).astype(np.float32)
    model = dib_amd.DistributedIBNet([1] * 8, [128, 128], [256, 256], 1, noise_seed=1, shuffle_seed=2, init_seed=3)
    opt = dib_amd.optimizers.get("adam")
    opt.learning_rate = 1e-3
    model.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-3, 0.5, 1, 2)
    hist = model.fit(x, y, epochs=3, batch_size=batch, callbacks=[cb], verbose=False, validation_data=(x[:300], y[:300]))
    if not distributed or dist.get_rank() == 0:
        np.savez(out_path, params=model.get_flat_weights(), **{k: np.array(v) for k, v in hist.history.items()})
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
