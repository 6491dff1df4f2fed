#!/usr/bin/env python
"""bench.py - headline benchmark of the MI355X-native Distributed-IB training path.

    python bench.py --gpus N --steps K --warmup W
(N>1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank
per GPU, RCCL all-reduce of the flat gradient buffer.)

Metric (BASELINE.json): DIB train samples/sec for one full step = fwd + per-feature KL + loss + bwd +
Adam (+ gradient all-reduce when N>1).  Workload = BASELINE config 3: 64 scalar features, per-GPU batch
65536, architecture fixed to the reference train.py defaults (encoder [128,128], E=32, positional
frequencies [2,4,8,16], integration [256,256], out=1, ReLU, Adam lr 3e-4), synthetic tabular data
(BASELINE.md section 4), fp32 end to end like the reference.  Weak scaling: per-GPU batch fixed.

One JSON line on rank 0 with `roofline` (dominant kernel = the grouped fp32-MFMA GEMM family, timed live
with HIP events inside libdib_hip.so) and `cpu_baseline` (PyTorch-CPU eager restatement of the TF graph,
oracle/dib_torch_cpu.py, timed on this box's host cores on a bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_SAMPLE = 13141504      # GEMM FLOPs fwd+dgrad+wgrad, config 3
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
# HBM bytes per launch of each kernel from rocprofv3 PMC passes (profiles/, see DESIGN.md "Measurement"); filled in
# from the committed counter collection, None where not collected.
def _load_hbm_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic_per_kernel.json")) as fh:
            return {k: v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"] for k, v in json.load(fh).items()}
    except Exception:  # noqa: BLE001
        return {}


HBM_TRAFFIC = _load_hbm_traffic()
F, E, BATCH = 64, 32, 65536
ENC, INTEG = [128, 128], [256, 256]


def synthetic(n_rows, seed=20241008):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n_rows, F), dtype=np.float32)
    w = rng.standard_normal(8).astype(np.float32)
    y = ((x[:, :8] @ w + 0.5 * x[:, 0] * x[:, 1]) > 0).astype(np.float32)[:, None]
    return x, y


def flops_by_kernel():
    """algorithmic GEMM FLOPs per sample executed by each kernel symbol for BASELINE config 3
    (sum = FLOPS_PER_SAMPLE; recompute inside the fused backward is NOT counted)."""
    enc = [(5, 128), (128, 128), (128, 2 * E)]
    integ = [(F * E, 256), (256, 256), (256, 1)]
    fl = lambda i, o: 2 * i * o
    out = {
        "dib_fused_encoder_fwd_kernel": sum(fl(i, o) for i, o in enc) * F,            # 3 encoder layers fwd
        "dib_fused_encoder_bwd_kernel": (sum(fl(i, o) for i, o in enc[1:]) + fl(*enc[0])) * F,  # dgrads L3, L2 + wgrad L1
        "dib_gemm_kernel<0, 2, 2, 64>": fl(*integ[0]) + fl(*integ[1]),                    # integration fwd, N >= 128
        "dib_gemm_kernel<1, 2, 2, 64>": fl(*integ[0]) + fl(*integ[1]),                    # integration dgrads, N >= 128
        "dib_gemm_kernel<2, 2, 1, 32>": fl(*enc[2]) * F,                                  # encoder layer-3 wgrad (N = 64)
        "dib_skinny_{fwd,dgrad,wgrad}_kernel": 3 * fl(*integ[2]),                         # 256 -> 1 output layer (HBM-bound)
        "dib_gemm_kernel<2, 2, 2, 64>": fl(*integ[0]) + fl(*enc[1]) * F,                  # wgrads with M, N >= 128
        "dib_gemm_kernel<2, 1, 2, 32>": fl(*integ[1]),                                    # integration 256 x 256 wgrad: 64-row tiles
    }
    assert sum(out.values()) == FLOPS_PER_SAMPLE
    return out


def _cpu_baseline_worker(threads, budget_s):
    """runs in a subprocess (hard wall-clock bound by the parent)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dib_oracle as orc
    from dib_torch_cpu import TorchCpuDIB
    torch.set_num_threads(threads)
    spec = orc.DIBSpec([1] * F, ENC, INTEG, 1)
    params = orc.glorot_uniform_init(spec, 0, dtype=np.float32)
    model = TorchCpuDIB(spec, params)
    # calibrate on a small batch, then size the timed batch to the budget
    xs, ys = synthetic(512)
    eps = torch.randn(512, F, E)
    model.train_step(torch.from_numpy(xs), torch.from_numpy(ys), eps, 1e-3, "bce_logits")
    t0 = time.perf_counter()
    model.train_step(torch.from_numpy(xs), torch.from_numpy(ys), eps, 1e-3, "bce_logits")
    per_row = (time.perf_counter() - t0) / 512
    b = int(min(8192, max(512, 2 ** int(np.log2(max(1.0, budget_s / 4 / per_row))))))
    x, y = synthetic(b)
    xt, yt, eps = torch.from_numpy(x), torch.from_numpy(y), torch.randn(b, F, E)
    model.train_step(xt, yt, eps, 1e-3, "bce_logits")
    t0 = time.perf_counter()
    steps = 0
    while steps < 8:
        model.train_step(xt, yt, eps, 1e-3, "bce_logits")
        steps += 1
        if time.perf_counter() - t0 > budget_s / 2:
            break
    el = time.perf_counter() - t0
    print(json.dumps({"value": round(steps * b / el, 1), "unit": "samples/s", "cores": threads, "kind": "port",
                      "sample": f"{steps} steps x {b} rows of the same 64-feature workload (fwd+KL+bwd+Keras-Adam), "
                                f"PyTorch-CPU eager restatement of the TF graph (oracle/dib_torch_cpu.py; not "
                                f"TensorFlow), {threads} threads of {os.cpu_count()} host cpus, {el:.1f}s"}))


def cpu_baseline(budget_s=24.0):
    """PyTorch-CPU eager restatement of the reference TF graph on a bounded sample of the same workload.
    Thread count capped at 32 (more threads make the many tiny per-feature ops slower, measured), run in a
    subprocess with a hard timeout so the default bench always finishes within minutes."""
    import subprocess
    threads = max(1, min(32, os.cpu_count() or 1))
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads),
                              str(budget_s)], capture_output=True, text=True, timeout=budget_s * 4 + 60,
                             env=dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES=""))
        return json.loads(res.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "samples/s", "cores": threads, "kind": "port",
                "sample": f"cpu baseline did not finish: {type(e).__name__}"}


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-worker":
        _cpu_baseline_worker(int(sys.argv[2]), float(sys.argv[3]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (default: BASELINE config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-buckets", type=int, default=2, choices=[1, 2],
                    help="gradient all-reduce buckets: 2 = integration bucket overlapped with the encoder backward")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run always take the RCCL path (also with 1 rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback on the product path)"
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(dev)

    import dib_amd  # noqa: F401
    from dib_amd.engine import HipEngine
    eng = HipEngine([1] * F, ENC, INTEG, 1, device=dev, init_seed=0)
    B = args.batch
    n_rows = B * 4  # 4 distinct batches per rank, cycled
    x, y = synthetic(n_rows, seed=20241008 + rank)
    xd, yd = eng.to_device(x), eng.to_device(y)
    eng.set_beta(1e-3)
    eng.set_lr(3e-4)
    inv_gb = 1.0 / (B * world)

    enc_off, enc_cnt = eng.part_range(0)

    def step(i):
        row0 = (i % 4) * B
        if dist is None:
            eng.train_step(xd, yd, None, row0, B, 0, i, "bce_logits", inv_global_batch=inv_gb)
        elif args.dp_buckets == 1:  # single all-reduce of the whole flat gradient buffer after the backward
            eng.train_step(xd, yd, None, row0, B, 0, i, "bce_logits", inv_global_batch=inv_gb)
            dist.all_reduce(eng.grads)
        else:
            # two gradient buckets: the integration network's all-reduce (RCCL over xGMI) is issued as soon as its
            # gradients are final and overlaps the encoder-bank backward; the encoder bucket follows the backward
            pending = []
            eng.train_step(xd, yd, None, row0, B, 0, i, "bce_logits", inv_global_batch=inv_gb,
                           on_integration_grads_ready=lambda g: pending.append(dist.all_reduce(g, async_op=True)))
            pending.append(dist.all_reduce(eng.grads[enc_off: enc_off + enc_cnt], async_op=True))
            for w in pending:
                w.wait()
        eng.adam_step()

    for i in range(args.warmup):
        step(i)
    timing = (not args.no_kernel_timing) and hasattr(eng, "profile_enable")
    if timing:
        eng.profile_enable(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_summary() if timing else None
    if timing:
        eng.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        sps = args.steps * B * world / elapsed
        out = {"metric": "DIB train samples/sec (fwd+KL+bwd+Adam)", "value": round(sps, 1), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE config 3: synthetic tabular, 64 scalar features, posenc [2,4,8,16], "
                                      "encoder [128,128], E=32, integration [256,256], out=1, BCE-from-logits, Adam",
                          "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                          "params": eng.n_params, "flops_per_sample": FLOPS_PER_SAMPLE},
               "step_roofline": {"bound": "mfma", "achieved": round(sps * FLOPS_PER_SAMPLE / 1e12 / world, 3),
                                 "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(sps * FLOPS_PER_SAMPLE / 1e12 / world / PEAK_F32_MFMA_TFLOPS, 4),
                                 "note": "whole-step algorithmic GEMM FLOPs / wall time, per GPU"}}
        if prof:
            fl = flops_by_kernel()
            per = {}
            for name, (ms, cnt) in prof.items():
                if name in fl and cnt:
                    tf = fl[name] * B * args.steps / (ms * 1e-3) / 1e12
                    per[name] = {"launches": cnt, "avg_launch_ms": round(ms / cnt, 5), "ms_per_step": round(ms / args.steps, 4),
                                 "flops_per_launch": fl[name] * B * args.steps // cnt, "achieved": round(tf, 2),
                                 "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
            if per:
                dom = max(per, key=lambda k: per[k]["ms_per_step"])
                out["roofline"] = {"bound": "mfma", "achieved": per[dom]["achieved"], "peak": PEAK_F32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": per[dom]["frac"], "traffic": HBM_TRAFFIC.get(dom),
                                   "kernel": dom, "avg_launch_ms": per[dom]["avg_launch_ms"],
                                   "launches": per[dom]["launches"], "flops_per_launch": per[dom]["flops_per_launch"]}
                out["roofline_by_kernel"] = per
                out["mfma_kernels_ms_per_step"] = round(sum(v["ms_per_step"] for v in per.values()), 4)
        if "roofline" not in out:
            out["roofline"] = dict(out["step_roofline"], traffic=None)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
