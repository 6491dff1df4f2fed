"""Worker for tests/test_gpu_dp_and_cache.py: a few SetTransformerDIB.train_step calls on the GPU; under
torch.distributed.run they take the RCCL data-parallel branch (neighbourhood shard, gradient + statistics all-reduce),
otherwise the single-process branch."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(out_path):
    import torch.distributed as dist
    distributed = "RANK" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{lr}"))
    import dib_amd
    from dib_amd import set_transformer as st
    if distributed:
        st._FORCE_DP_BRANCH = True   # world size 1 would otherwise take the single-process branch: run the RCCL calls
    m = dib_amd.SetTransformerDIB(number_attention_blocks=2, init_seed=1, noise_seed=2)
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((6, 64, 12)).astype(np.float32)
    y = (rng.random((6, 1)) > 0.5).astype(np.float32)
    series = []
    for step in range(3):
        m.lr_dev.fill_(m.learning_rate_schedule(step + 1, 1e-3, 10))
        m.beta_dev.fill_(m.beta_schedule(step, 1e-3, 1e-1, 3))
        bce = m.train_step(feats, y)
        series.append([float(bce.item()), float(m.last["kl"].item())])
    val = m.train_step(feats[:4], y[:4], training=False)
    series.append([float(val.item()), float(m.last["kl"].item())])
    torch.cuda.synchronize()
    if not distributed or dist.get_rank() == 0:
        np.savez(out_path, params=m.params.cpu().numpy(), series=np.array(series), t=int(m.t_dev.item()))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
