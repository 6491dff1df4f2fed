#!/usr/bin/env python
"""bench.py - headline benchmark of the MI355X-native Distributed-IB training path.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak]

`--gpus N` with N > 1 and no launcher environment: bench.py re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU, RCCL
(`nccl` backend) all-reduce of the flat gradient buffer in two buckets.  Launched BY torch.distributed.run (the driver's
form) it reads RANK / LOCAL_RANK / WORLD_SIZE and insists that WORLD_SIZE == N.

Metric (BASELINE.json): DIB train samples/sec for one full step = fwd + per-feature KL + loss + bwd + Adam (+ gradient
all-reduce when N>1).  Workload = BASELINE config 3: synthetic tabular data, 2^20 rows x 64 scalar features
resident in HBM, global batch 65536 (16 distinct batches), architecture = the reference train.py defaults (encoder
[128,128], E=32, positional frequencies [2,4,8,16], integration [256,256], out=1, ReLU, BCE-from-logits, Adam lr 3e-4),
fp32 end to end like the reference.

Scaling: default STRONG - the 65536-row global batch is sharded, rank r takes rows [r*B/N, (r+1)*B/N) of
each global batch; `--scaling weak` keeps 65536 rows per GPU.  With N > 1 the other mode is measured too and reported
under `extra`.

Timing: W warm-up steps, then `--blocks` (default 3) timed blocks of EXACTLY K steps, each bracketed by barrier +
torch.cuda.synchronize() on both sides, max over ranks; `value` is the MEDIAN block (box-to-box and run-to-run spread is
6-8 %, so a single block is noisy); all blocks are listed.  These blocks run WITHOUT per-kernel event timing.  One more
K-step block then runs with HIP events around every MFMA kernel (inside libdib_hip.so, on the launch stream) for the
`roofline` of the dominant kernel; its step time is reported as `ms_per_step_kernel_timing`.

One JSON line on rank 0 with `roofline`, `cpu_baseline` (PyTorch-CPU eager restatement of the TF graph,
oracle/dib_torch_cpu.py, timed on this box's host cores on a bounded sample; N=1 only) and `extra` (BASELINE config 4,
F = 50 shell features, same step at B = 65536; N=1 only).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_SAMPLE = 13141504      # GEMM FLOPs fwd+dgrad+wgrad, config 3
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
N_ROWS = 1 << 20                 # dataset of 2^20 rows held on the device
E, GLOBAL_BATCH = 32, 65536
ENC, INTEG = [128, 128], [256, 256]


# HBM bytes per launch of each kernel from rocprofv3 PMC passes (profiles/, see DESIGN.md "Measurement"); filled in
# from the committed counter collection, None where not collected.
def _load_hbm_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic_per_kernel.json")) as fh:
            return {k: v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"] for k, v in json.load(fh).items()}
    except Exception:  # noqa: BLE001
        return {}


HBM_TRAFFIC = _load_hbm_traffic()
# `roofline.traffic` is NOT collected by this run (PMC counters need rocprofv3 around the process): it is the per-launch
# FETCH_SIZE + WRITE_SIZE of the same kernel at the same workload from the committed counter passes named here.
TRAFFIC_SOURCE = ("profiles/hbm_traffic_per_kernel.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                  "(tools/collect_profiles.sh + tools/summarize_profiles.py), committed with the round's profiles - a constant "
                  "read from that file, not measured in this run")


def synthetic(n_rows, n_features=64, seed=20241008):
    """x ~ N(0,1), y = 1[sum_{j<8} w_j x_j + 0.5 x_0 x_1 > 0]."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n_rows, n_features), dtype=np.float32)
    w = rng.standard_normal(8).astype(np.float32)
    y = ((x[:, :8] @ w + 0.5 * x[:, 0] * x[:, 1]) > 0).astype(np.float32)[:, None]
    return x, y


def gemm_flops_per_sample(n_features, in_dim=5):
    """SURVEY 8(a) generic formula: fwd = sum_f sum_l 2 in out + sum_l 2 in out; bwd = 2 fwd - sum_f 2 in_1 out_1."""
    enc = [(in_dim, ENC[0]), (ENC[0], ENC[1]), (ENC[1], 2 * E)]
    integ = [(n_features * E, INTEG[0]), (INTEG[0], INTEG[1]), (INTEG[1], 1)]
    fwd = n_features * sum(2 * i * o for i, o in enc) + sum(2 * i * o for i, o in integ)
    return 3 * fwd - n_features * 2 * enc[0][0] * enc[0][1]


def flops_by_kernel(F=64):
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
        "dib_gemm_kernel<2, 1, 2, 64>": fl(*integ[1]),                                    # integration 256 x 256 wgrad: 64-row tiles
    }
    assert sum(out.values()) == gemm_flops_per_sample(F)
    return out


assert gemm_flops_per_sample(64) == FLOPS_PER_SAMPLE


def per_kernel_roofline(prof, n_features, rows_per_step, steps):
    """{kernel symbol: launches, avg ms, ms/step, algorithmic FLOPs per launch, achieved TFLOP/s, fraction of the fp32-MFMA peak}
    from the library's live HIP-event timing (dib_profile_summary) of `steps` steps of `rows_per_step` rows."""
    fl = flops_by_kernel(n_features)
    per = {}
    for name, (ms, cnt) in prof.items():
        if name in fl and cnt:
            tf = fl[name] * rows_per_step * steps / (ms * 1e-3) / 1e12
            per[name] = {"launches": cnt, "avg_launch_ms": round(ms / cnt, 5), "ms_per_step": round(ms / steps, 4),
                         "flops_per_launch": fl[name] * rows_per_step * steps // cnt, "achieved": round(tf, 2),
                         "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
    rest = {n: round(ms / steps, 4) for n, (ms, cnt) in prof.items() if n not in fl and cnt}
    return per, rest


def _cpu_baseline_worker(threads, budget_s):
    """runs in a subprocess (hard wall-clock bound by the parent)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dib_oracle as orc
    from dib_torch_cpu import TorchCpuDIB
    torch.set_num_threads(threads)
    F = 64
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
    while steps < 64:  # about budget_s / 2 seconds of CPU work (>= 10 s at the default budget)
        model.train_step(xt, yt, eps, 1e-3, "bce_logits")
        steps += 1
        if time.perf_counter() - t0 > budget_s / 2:
            break
    el = time.perf_counter() - t0
    # the reference's own default size (train.py:30-34 on the Boolean circuit: F = 10, B = 128, BASELINE configs[0]) - a step is
    # ~100 tiny eager ops, so fewer threads are faster: best of 1 and 4 threads, about 2 s each
    small = {}
    spec_s = orc.DIBSpec([1] * 10, ENC, INTEG, 1)
    xs_, ys_ = synthetic(128, 10)
    xs_, ys_, eps_s = torch.from_numpy(xs_), torch.from_numpy(ys_), torch.randn(128, 10, E)
    for nt in (1, 4):
        torch.set_num_threads(nt)
        ms = TorchCpuDIB(spec_s, orc.glorot_uniform_init(spec_s, 0, dtype=np.float32))
        for _ in range(3):
            ms.train_step(xs_, ys_, eps_s, 1e-3, "bce_logits")
        t1, n = time.perf_counter(), 0
        while time.perf_counter() - t1 < 2.0:
            ms.train_step(xs_, ys_, eps_s, 1e-3, "bce_logits")
            n += 1
        small[nt] = (time.perf_counter() - t1) / n
    nt_best = min(small, key=small.get)
    print(json.dumps({"value": round(steps * b / el, 1), "unit": "samples/s", "cores": threads, "kind": "port",
                      "reference_default_size": {"workload": "F = 10, B = 128 training step (Boolean-circuit default of train.py)",
                                                 "ms_per_train_step": round(1e3 * small[nt_best], 3), "cores": nt_best,
                                                 "seconds_for_the_reference_88000_steps": round(small[nt_best] * 88000, 1)},
                      "sample": f"{steps} steps x {b} rows of the same 64-feature workload (fwd+KL+bwd+Keras-Adam), "
                                f"PyTorch-CPU eager restatement of the TF graph (oracle/dib_torch_cpu.py; not "
                                f"TensorFlow), {threads} threads of {os.cpu_count()} host cpus, {el:.1f}s"}))


def cpu_baseline(budget_s=24.0):
    """PyTorch-CPU eager restatement of the reference TF graph on a bounded sample of the same workload.
    Thread count capped at 32 (more threads make the many tiny per-feature ops slower, measured), run in a
    subprocess with a hard timeout so the default bench always finishes within minutes."""
    threads = max(1, min(32, os.cpu_count() or 1))
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads),
                              str(budget_s)], capture_output=True, text=True, timeout=budget_s * 4 + 60,
                             env=dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES=""))
        return json.loads(res.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "samples/s", "cores": threads, "kind": "port",
                "sample": f"cpu baseline did not finish: {type(e).__name__}"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _respawn_under_launcher(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) of this same command line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def config5_set_transformer(dev, nb=4, npart=4096, nfeat=16, steps=4, warmup=2):
    """BASELINE config 5 on one GPU: per-particle set-transformer DIB (notebook ...set_transformer.ipynb:332-431), `nb`
    neighbourhoods x 4096 particles, 3-D positions (16 derived per-particle features), the notebook's architecture; one step =
    fwd + KL + BCE + bwd + Adam.  Second block of steps with HIP events around the attention kernels for the roofline of the
    dominant kernel, dib_attn_bwd_kernel, priced on ALGORITHMIC FLOPs: the four backward products dV, dP, dQ, dK =
    2 x the forward's two (the recomputed S = Q K^T the kernel also executes is not counted)."""
    from dib_amd import SetTransformerDIB
    from dib_amd.engine import profile_summary
    torch.cuda.empty_cache()
    st = SetTransformerDIB(particle_feature_dimensions=nfeat, device=dev)
    rng = np.random.default_rng(5)
    xs = torch.from_numpy(rng.standard_normal((nb, npart, nfeat)).astype(np.float32)).to(dev)
    ys = torch.from_numpy((rng.random((nb, 1)) > 0.5).astype(np.float32)).to(dev)
    st.beta_dev.fill_(1e-3)

    def block(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            st.train_step(xs, ys)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    block(warmup)
    dt5 = block(steps)
    st.lib.dib_profile_enable(1)
    dt5_prof = block(steps)
    prof = profile_summary(st.lib)
    st.lib.dib_profile_enable(0)
    D, H, K, nblk = st.bottleneck_dimension, st.number_heads_per_mha, st.key_dim, st.number_attention_blocks
    attn_fwd = 4 * H * K * npart * npart                        # S = Q K^T and O = P V per neighbourhood and block
    per_blk = 3 * 2 * D * H * K * npart + attn_fwd + 2 * H * K * D * npart + 2 * (D * 128 + 128 * D) * npart
    fwd = 2 * (nfeat * 5 * 128 + 128 * 128 + 128 * 64) * npart + nblk * per_blk
    out = {"workload": f"BASELINE config 5: per-particle set-transformer DIB, {nb} neighbourhoods x {npart} particles x "
                       f"{nfeat} features (3-D positions), {nblk} x [MHA {H} x {K}, Add+LN, FF, Add+LN], fp32",
           "value": round(nb / dt5, 2), "unit": "neighbourhoods/s", "ms_per_step": round(1e3 * dt5, 2), "steps": steps,
           "algorithmic_TFLOPs": round(3 * fwd * nb / dt5 / 1e12, 2),
           "step_roofline_frac": round(3 * fwd * nb / dt5 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
           "attention": st.attention_impl, "params": st.n_params,
           "ms_per_step_kernel_timing": round(1e3 * dt5_prof, 2)}
    by = {}
    for name, fl in (("dib_attn_fwd_kernel", attn_fwd * nb), ("dib_attn_bwd_kernel", 2 * attn_fwd * nb)):
        if name in prof and prof[name][1]:
            ms, cnt = prof[name]
            tf = fl / (ms / cnt * 1e-3) / 1e12
            by[name] = {"launches": cnt, "avg_launch_ms": round(ms / cnt, 4), "ms_per_step": round(ms / steps, 3),
                        "flops_per_launch": fl, "achieved": round(tf, 2), "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
    if by:
        dom = max(by, key=lambda k: by[k]["ms_per_step"])
        out["roofline"] = {"bound": "mfma", "achieved": by[dom]["achieved"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": by[dom]["frac"], "traffic": HBM_TRAFFIC.get(dom + "@config5"),
                           "traffic_source": TRAFFIC_SOURCE, "kernel": dom,
                           "avg_launch_ms": by[dom]["avg_launch_ms"], "launches": by[dom]["launches"],
                           "flops_per_launch": by[dom]["flops_per_launch"],
                           "note": "algorithmic FLOPs: the 4 backward tile products = 2 x forward; " +
                                   ("score-stash mode: exactly these 4 are executed (the forward left the raw score tiles in HBM)"
                                    if st.last["plan"]["stash"] is not None else
                                    "recompute mode: the kernel executes a 5th product, S = q k^T")}
        out["attention_score_stash"] = st.last["plan"]["stash"] is not None
        out["roofline_by_kernel"] = by
    del st, xs, ys
    torch.cuda.empty_cache()
    return out


def fit_surface(dev, batch, engine_s_per_step):
    """Throughput of the surface north_star names - DistributedIBNet.compile + fit (reference train.py:138-166) - on BASELINE
    config 3: epochs of 16 steps of `batch` rows over the 2^20-row dataset, shuffle on, the annealing callback and the History
    accounting on; next to the engine-level number of the headline (HipEngine.train_step + adam_step called directly).
    One warm-up epoch (workspace allocation, dataset upload), then two timed epochs."""
    import dib_amd
    x, y = synthetic(N_ROWS, 64)
    model = dib_amd.DistributedIBNet([1] * 64, ENC, INTEG, 1, feature_embedding_dimension=E, device=dev)
    opt = dib_amd.optimizers.get("adam")
    opt.learning_rate = 3e-4
    model.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-3, 1.0, 1, 2)
    times = []

    class _Clock:   # a Keras-style callback: wall time of every epoch, device drained at both ends
        model = None

        def set_model(self, m):
            self.model = m

        def on_epoch_begin(self, epoch, logs=None):
            torch.cuda.synchronize()
            self.t0 = time.perf_counter()

        def on_epoch_end(self, epoch, logs=None):
            torch.cuda.synchronize()
            times.append(time.perf_counter() - self.t0)

    torch.cuda.synchronize()
    t_fit = time.perf_counter()
    hist = model.fit(x, y[:, 0], epochs=3, shuffle=True, batch_size=batch, callbacks=[cb, _Clock()], verbose=False)
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t_fit
    steps = N_ROWS // batch
    dt = statistics.median(times[1:]) / steps
    # the boundary hands over HOST arrays: fit uploads the dataset once (it stays resident in HBM for every epoch) - the
    # PCIe-inclusive rate is the whole call (upload + workspace allocation + first-touch + 3 epochs) over its 3 * steps steps
    t0 = time.perf_counter()
    xd = torch.from_numpy(x).to(dev)
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0
    del xd
    return {"workload": f"DistributedIBNet.fit on BASELINE config 3: epochs of {steps} steps x {batch} rows, shuffle=True, "
                        "InfoBottleneckAnnealingCallback + History on; median of 2 timed epochs after 1 warm-up epoch",
            "value": round(batch / dt, 1), "unit": "samples/s", "ms_per_step": round(1e3 * dt, 4),
            "engine_ms_per_step": round(1e3 * engine_s_per_step, 4), "fit_over_engine": round(dt / engine_s_per_step, 4),
            "epochs_ms": [round(1e3 * t, 2) for t in times], "final_loss": round(float(hist.history["loss"][-1]), 5),
            "whole_call_incl_upload": {"ms": round(1e3 * t_fit, 2), "steps": 3 * steps,
                                       "samples_per_s": round(3 * steps * batch / t_fit, 1),
                                       "note": "model.fit from host numpy arrays to History, 3 epochs: dataset upload over PCIe, "
                                               "workspace allocation and the warm-up epoch included - never the headline"},
            "dataset_upload": {"bytes": int(x.nbytes), "ms": round(1e3 * t_up, 2), "GB_per_s": round(x.nbytes / t_up / 1e9, 2),
                               "note": "pageable numpy -> HBM, once per fit"}}


def config2_infonce_loop(dev, batch):
    """BASELINE config 2's training path: the custom InfoNCE loop (reference train.py:180-289) on the double-pendulum layout -
    feature dimensionalities [2, 1, 2, 1] (angles as unit vectors, velocities), positional encoding, encoder [128, 128], E = 32,
    integration [256, 256] -> 64-d shared space, Y encoder [128, 128] (a DenseStack), similarity l2, Adam.  Synthetic data of
    that shape; one step = X forward + Y forward + [B, B] InfoNCE loss and gradients + both backwards + both Adam updates."""
    import dib_amd
    from dib_amd import infonce
    rng = np.random.default_rng(7)
    n = batch * 16
    x = rng.standard_normal((n, 6)).astype(np.float32)
    y = (x + 0.3 * rng.standard_normal((n, 6))).astype(np.float32)
    model = dib_amd.DistributedIBNet([2, 1, 2, 1], ENC, INTEG, 64, feature_embedding_dimension=E, device=dev)
    kw = dict(batch_size=batch, number_pretraining_epochs=1, number_annealing_epochs=1, beta_start=1e-4, beta_end=1.0,
              learning_rate=3e-4, shared_dimensionality=64, similarity="l2")
    infonce.fit_infonce(model, x, y, x[:batch], y[:batch], **kw)                  # warm-up: 16 train + 4 validation steps
    torch.cuda.synchronize()
    lib = model._engine.lib
    l0 = int(lib.dib_launch_count())
    t0 = time.perf_counter()
    kw.update(number_pretraining_epochs=2, number_annealing_epochs=3)
    infonce.fit_infonce(model, x, y, x[:batch], y[:batch], **kw)                  # 4 x 16 train steps + 4 x 2 validation steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (4 * 16 + 4 * 2)
    launches = int(lib.dib_launch_count()) - l0   # library kernels of 64 training + 8 validation steps (the loop runs no torch kernel per step)
    # The 72-step call above carries the call's ONE-TIME work (a fresh output encoder: parameter init, uploads, its launch plans;
    # dataset upload; the batch stream) - ~ 25 us per step at this length, nothing at the reference's (1 875 steps per epoch,
    # train.py:222-236).  Second figure: the same loop over 16 x as many steps per epoch with the encoder built beforehand
    # (the kernel timeline of the step: profiles/r06l_config2_loop_b128_timeline.txt).
    from dib_amd.dense import DenseStack
    nl = batch * 256
    xl = rng.standard_normal((nl, 6)).astype(np.float32)
    yl = (xl + 0.3 * rng.standard_normal((nl, 6))).astype(np.float32)
    yenc = DenseStack(model._engine, 6, [128, 128], 64, "relu", True, 5, seed=1)
    kw.update(number_pretraining_epochs=1, number_annealing_epochs=1)
    infonce.fit_infonce(model, xl[:batch * 16], yl[:batch * 16], xl[:batch], yl[:batch], output_encoder=yenc, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kw.update(number_pretraining_epochs=1, number_annealing_epochs=2)
    infonce.fit_infonce(model, xl, yl, xl[:batch], yl[:batch], output_encoder=yenc, **kw)   # 2 x 256 train + 2 x 2 validation steps
    torch.cuda.synchronize()
    dt_long = (time.perf_counter() - t0) / (2 * 256 + 2 * 2)
    # algorithmic GEMM FLOPs of one TRAINING step (SURVEY 8a conventions: fwd + dgrad + wgrad, first-layer dgrads excluded):
    # X model on the pendulum layout, Y encoder 30 -> 128 -> 128 -> 64, and the InfoNCE products S = X Y^T, C Y, C^T X
    x_enc = [[(5 * d, ENC[0]), (ENC[0], ENC[1]), (ENC[1], 2 * E)] for d in (2, 1, 2, 1)]
    x_int = [(4 * E, INTEG[0]), (INTEG[0], INTEG[1]), (INTEG[1], 64)]
    fwd_x = sum(2 * i * o for f in x_enc for i, o in f) + sum(2 * i * o for i, o in x_int)
    y_enc = [(30, 128), (128, 128), (128, 64)]
    fwd_y = sum(2 * i * o for i, o in y_enc)
    per_sample = 3 * fwd_x - sum(2 * f[0][0] * f[0][1] for f in x_enc) + 3 * fwd_y - 2 * 30 * 128
    flops = per_sample * batch + 3 * 2 * batch * batch * 64
    # 64 of the 72 timed steps are training steps (the 8 validation steps run the forward halves only): the fraction below
    # prices every timed step as a training step's FLOPs x 64/72 - a slight overstatement of the work, stated here
    tf = flops * (4 * 16) / (4 * 16 + 4 * 2) / dt / 1e12
    return {"batch": batch, "ms_per_step": round(1e3 * dt, 3), "samples_per_s": round(batch / dt, 1),
            "ms_per_step_256_steps_per_epoch": round(1e3 * dt_long, 3),
            "library_launches_per_step": round(launches / (4 * 16 + 4 * 2), 2),
            "flops_per_train_step": int(flops), "algorithmic_TFLOPs": round(tf, 3),
            "step_roofline_frac": round(tf / PEAK_F32_MFMA_TFLOPS, 5)}


def reference_size_set_transformer(dev, steps=30, warmup=5):
    """The notebook's own configuration (...set_transformer.ipynb:304-307, 419-431): 32 neighbourhoods x 50 particles x 12
    features, 6 attention blocks - ~90 launches per step (190 before round 5), bound by launch / dependency latency.  Eager launches
    vs one hipGraph replay per step (SetTransformerDIB(use_graphs=True), the DIB_ENABLE_GRAPHS opt-in)."""
    from dib_amd import SetTransformerDIB
    rng = np.random.default_rng(6)
    xs = torch.from_numpy(rng.standard_normal((32, 50, 12)).astype(np.float32)).to(dev)
    ys = torch.from_numpy((rng.random((32, 1)) > 0.5).astype(np.float32)).to(dev)
    out = {"workload": "set-transformer DIB at the notebook's size: 32 neighbourhoods x 50 particles x 12 features, fp32"}
    for key, graphs in (("eager", False), ("graph_replay", True)):
        st = SetTransformerDIB(device=dev, use_graphs=graphs)
        st.beta_dev.fill_(1e-3)
        for _ in range(warmup):
            st.train_step(xs, ys)
        torch.cuda.synchronize()
        l0 = int(st.lib.dib_launch_count())
        t0 = time.perf_counter()
        for _ in range(steps):
            st.train_step(xs, ys)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        launches = (int(st.lib.dib_launch_count()) - l0) / steps   # (a graph replay issues none from the host)
        D, H, K, P, nfeat = st.bottleneck_dimension, st.number_heads_per_mha, st.key_dim, 50, 12
        per_blk = 3 * 2 * D * H * K * P + 4 * H * K * P * P + 2 * H * K * D * P + 2 * (D * 128 + 128 * D) * P
        fwd = 2 * (nfeat * 5 * 128 + 128 * 128 + 128 * 64) * P + st.number_attention_blocks * per_blk   # per neighbourhood
        tf = 3 * fwd * 32 / dt / 1e12
        out[key] = {"ms_per_step": round(1e3 * dt, 3), "neighbourhoods_per_s": round(32 / dt, 1),
                    "algorithmic_TFLOPs": round(tf, 2), "step_roofline_frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                    "library_launches_per_step": round(launches, 1)}
        out["attention"] = "single-workgroup kernels for P <= 64 (csrc/dib_attn_small.h)"
        out["token_chain"] = "one launch per attention block and direction for the token-wise half (csrc/dib_st_chain.h)"
        del st
    return out


def keras_path_default_batch(dev, epochs=200):
    """The reference's own default run (train.py:30-34: Boolean circuit, 10 scalar features, B = 128, 8 steps per epoch,
    validation every epoch) through DistributedIBNet.fit: a step is ~25 dependent launches of a few microseconds each."""
    import dib_amd
    d = dib_amd.data.fetch_boolean_circuit()
    m = dib_amd.DistributedIBNet(d["feature_dimensionalities"], ENC, INTEG, 1, feature_embedding_dimension=E, device=dev)
    opt = dib_amd.optimizers.get("adam")
    opt.learning_rate = 3e-4
    m.compile(optimizer=opt, loss=d["loss"], metrics=d["metrics"])
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-4, 3.0, 10, 40)
    kw = dict(batch_size=128, callbacks=[cb], verbose=False, validation_data=(d["x_valid"], d["y_valid"]))
    m.fit(d["x_train"], d["y_train"], epochs=3, **kw)
    torch.cuda.synchronize()
    lib = m._engine.lib

    def timed():
        l0 = int(lib.dib_launch_count())
        t0 = time.perf_counter()
        m.fit(d["x_train"], d["y_train"], epochs=epochs, **kw)
        torch.cuda.synchronize()
        # per (training step + validation step) pair
        return (time.perf_counter() - t0) / epochs / 8, (int(lib.dib_launch_count()) - l0) / epochs / 8

    dt, launches = timed()
    # fit evaluates the epoch's 8 validation batches as ONE 1024-row launch set (model.validation_merge_rows: same per-row
    # numbers, History sums in another order); the same run with them evaluated one by one, for the record
    merge = m.validation_merge_rows
    m.validation_merge_rows = 0
    m.fit(d["x_train"], d["y_train"], epochs=3, **kw)
    dt1, launches1 = timed()
    m.validation_merge_rows = merge
    fl = gemm_flops_per_sample(10, 5) * 128                # training step only (the validation forward is ~1/3 more)
    return {"workload": "reference default: Boolean circuit, F = 10, B = 128, 8 train + 8 validation steps per epoch, fit()",
            "us_per_train_plus_validation_step": round(1e6 * dt, 1), "epochs_timed": epochs,
            "library_launches_per_train_plus_validation_step": round(launches, 2),
            "validation": f"the epoch's 8 validation batches in one launch set of {merge} rows (fit.validation_merge_rows)",
            "validation_batches_one_by_one": {"us_per_train_plus_validation_step": round(1e6 * dt1, 1),
                                              "library_launches_per_train_plus_validation_step": round(launches1, 2)},
            "seconds_for_the_reference_11000_epochs": round(dt * 8 * 11000, 1),
            "flops_per_train_step": int(fl), "step_roofline_frac_lower_bound": round(fl / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 6)}


class Workload:
    """One engine + resident dataset + the step of the benchmark for a (feature count, scaling mode)."""

    def __init__(self, n_features, dev, rank, world, dist, scaling, global_batch, dp_buckets, engine=None, n_rows=N_ROWS):
        """engine / n_rows: the CPU test of this class's data-parallel step (tests/test_bench_launcher.py) passes the
        test-only float64 engine and a small dataset and runs the SAME step / dp_breakdown code over gloo."""
        self.F, self.rank, self.world, self.dist = n_features, rank, world, dist
        if engine is None:
            from dib_amd.engine import HipEngine
            engine = HipEngine([1] * n_features, ENC, INTEG, 1, device=dev, init_seed=0)
        self.eng, self.n_rows = engine, int(n_rows)
        self._sync = torch.cuda.synchronize if str(dev).startswith("cuda") else (lambda: None)
        x, y = synthetic(self.n_rows, n_features)
        self.xd, self.yd = self.eng.to_device(x), self.eng.to_device(y)
        self.eng.set_beta(1e-3)
        self.eng.set_lr(3e-4)
        self.dp_buckets = dp_buckets
        self.enc_off, self.enc_cnt = self.eng.part_range(0)
        self.tail_range = self.eng.part_range(3)
        self.set_scaling(scaling, global_batch)

    def set_scaling(self, scaling, global_batch):
        self.scaling = scaling
        if scaling == "strong":   # rank r takes rows [r*B/N, (r+1)*B/N) of each global batch
            self.gb = global_batch
            self.lo = (self.gb * self.rank) // self.world
            self.B = (self.gb * (self.rank + 1)) // self.world - self.lo
        else:                     # weak: every GPU steps through its own 65536-row batches
            self.gb = global_batch * self.world
            self.lo, self.B = 0, global_batch
        self.n_batches = max(1, self.n_rows // global_batch)
        self.stride = global_batch

    def row0(self, i):
        if self.scaling == "strong":
            return (i % self.n_batches) * self.stride + self.lo
        return ((i * self.world + self.rank) % self.n_batches) * self.stride

    ADAM = ("adam", 0.9, 0.999, 1e-7)

    def step(self, i, comm="real"):
        """one training step.  comm: "real" = the product protocol; "none" = the same launches with every collective
        replaced by a no-op (what the step costs a rank before a byte is exchanged); used by dp_breakdown only."""
        eng, dist = self.eng, self.dist
        inv_gb = 1.0 / self.gb
        if dist is None:   # the step's last launch also reduces the partials, accumulates the metrics and applies Adam
            eng.train_step(self.xd, self.yd, None, self.row0(i), self.B, 0, i, "bce_logits", inv_global_batch=inv_gb,
                           optimizer=self.ADAM)
            return
        if self.dp_buckets == 1:  # single all-reduce of the whole flat gradient buffer after the backward
            eng.train_step(self.xd, self.yd, None, self.row0(i), self.B, 0, i, "bce_logits", inv_global_batch=inv_gb)
            if comm == "real":
                dist.all_reduce(eng.grads)
            eng.optimizer_step_part(self.B, -1, self.ADAM, bump=True)
            return
        # gradient buckets, each all-reduced (RCCL over xGMI, async) as soon as it is final: the integration
        # network's under the whole encoder-bank backward; with 3 buckets the encoder front layers' under the last
        # encoder layer's weight gradient, which alone trails the backward; with 2 the whole encoder bank trails it.
        # Each bucket is Adam-stepped as soon as ITS all-reduce has landed (buckets 1-2 while bucket 3 is on the wire).
        pending = []
        issue = (lambda g: pending.append(dist.all_reduce(g, async_op=True))) if comm == "real" else (lambda g: pending.append(None))
        kw = dict(on_encoder_front_grads_ready=issue) if self.dp_buckets == 3 else {}
        eng.train_step(self.xd, self.yd, None, self.row0(i), self.B, 0, i, "bce_logits", inv_global_batch=inv_gb,
                       on_integration_grads_ready=issue, **kw)
        off, cnt = self.tail_range if self.dp_buckets == 3 else (self.enc_off, self.enc_cnt)
        issue(eng.grads[off: off + cnt])
        parts = (1, 2, 3) if self.dp_buckets == 3 else (1, 0)
        for k, (w, part) in enumerate(zip(pending, parts)):
            if w is not None:
                w.wait()
            eng.optimizer_step_part(self.B, part, self.ADAM, bump=k == len(parts) - 1)

    def dp_breakdown(self, steps, dev):
        """N > 1: where a data-parallel step's time goes, so that the scaling record explains itself - the step with the
        collectives replaced by no-ops, each bucket's all-reduce alone (back to back, nothing to overlap with), and the
        step under each bucket protocol.  All max-over-ranks, barrier + synchronize on both sides."""
        dist, eng = self.dist, self.eng
        out = {"per_gpu_batch": self.B, "global_batch": self.gb, "steps": steps}

        def timed(fn, n):
            self._sync(); dist.barrier(); self._sync()
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            self._sync(); dist.barrier(); self._sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return 1e3 * float(t.item()) / n

        keep = self.dp_buckets
        for i in range(2):
            self.step(i, comm="none")
        out["step_ms_no_collectives"] = round(timed(lambda i: self.step(100 + i, comm="none"), steps), 4)
        comm = {}
        for part, name in ((1, "integration"), (2, "encoder_front"), (3, "encoder_last"), (0, "encoder_bank"), (-1, "all")):
            off, cnt = (0, eng.n_params) if part == -1 else eng.part_range(part)
            buf = torch.zeros(cnt, dtype=torch.float32, device=dev)
            for _ in range(3):
                dist.all_reduce(buf)
            ms = timed(lambda i: dist.all_reduce(buf), 20)
            comm[name] = {"floats": int(cnt), "all_reduce_ms": round(ms, 4),
                          "bus_GBps": round(2 * (self.world - 1) / self.world * cnt * 4 / (ms * 1e-3) / 1e9, 2)}
        out["all_reduce_alone"] = comm
        prot = {}
        for nb in (1, 2, 3):
            self.dp_buckets = nb
            for i in range(2):
                self.step(i)
            prot[f"buckets_{nb}"] = round(timed(lambda i: self.step(200 + i), steps), 4)
        self.dp_buckets = keep
        out["step_ms_by_protocol"] = prot
        out["exposed_communication_ms"] = {k: round(v - out["step_ms_no_collectives"], 4) for k, v in prot.items()}
        return out

    def timed_block(self, first_step, steps, dev):
        dist = self.dist
        self._sync()
        if dist is not None:
            dist.barrier()
        self._sync()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(first_step + i)
        self._sync()
        if dist is not None:
            dist.barrier()
        self._sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def measure(self, warmup, steps, blocks, dev):
        for i in range(warmup):
            self.step(i)
        times = [self.timed_block(warmup + b * steps, steps, dev) for b in range(blocks)]
        med = statistics.median(times)
        return med, times


class _ExtrasDeadline:
    """The multi-GPU `extra` measurements run AFTER the headline line has been assembled and under this deadline: a
    collective that never completes in one of them (ranks out of step after an error on one rank) must not take the measured
    headline down with it.  On expiry rank 0 prints the line with what it has and EVERY rank leaves with status 0 (each rank
    arms its own timer; they start within a barrier's skew of each other)."""

    def __init__(self, seconds: float, rank: int, out, extra: dict):
        self.seconds, self.rank, self.out, self.extra = float(seconds), rank, out, extra
        self.timer = threading.Timer(self.seconds, self._expire)
        self.timer.daemon = True

    def _expire(self):
        if self.rank == 0:
            self.out["extra"] = dict(self.extra, error=f"the multi-GPU extras did not finish within {self.seconds:.0f} s; the "
                                                       "headline measurement was complete before they started")
            print(json.dumps(self.out), flush=True)
        os._exit(0)

    def start(self):
        self.timer.start()

    def cancel(self):
        self.timer.cancel()


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-worker":
        _cpu_baseline_worker(int(sys.argv[2]), float(sys.argv[3]))
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps; value = median block")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default, SURVEY 8e): the 65536-row global batch is sharded over the GPUs; weak: 65536 rows per GPU")
    ap.add_argument("--batch", type=int, default=GLOBAL_BATCH, help="global batch (strong) / per-GPU batch (weak)")
    ap.add_argument("--features", type=int, default=64,
                    help="number of scalar features of the synthetic workload: 64 = BASELINE config 3 (the headline), 50 = "
                         "BASELINE config 4 (what the rocprofv3 passes of profiles/*_config4_* wrap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` measurements (other scaling mode, config 4)")
    ap.add_argument("--dp-buckets", type=int, default=3, choices=[1, 2, 3],
                    help="gradient all-reduce buckets: 3 (default) = integration / encoder front layers / last encoder layer, each "
                         "issued as soon as it is final; 2 = integration overlapped, whole encoder bank after the backward; 1 = one")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--extra-timeout", type=float, default=240.0,
                    help="N > 1: seconds the `extra` measurements may take before the headline line is printed without them")
    ap.add_argument("--config5-only", action="store_true",
                    help="run only the BASELINE config-5 set-transformer step (the `extra.config5_set_transformer` object) and "
                         "print it: the command the rocprofv3 passes of profiles/*_config5_* wrap")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE",
                    help="dib_set_tuning(KEY, VALUE) before any layout is created (include/dib_hip.h lists the keys; A/B runs)")
    ap.add_argument("--force-dp-extras", action="store_true",
                    help="run the multi-GPU `extra` measurements (dp_breakdown, other scaling mode) under ONE RCCL rank too "
                         "(python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --force-dp-extras): exercises the code "
                         "path the driver's SCALE run takes")
    ap.add_argument("--only-dp-breakdown", action="store_true",
                    help="with --force-dp-extras: the data-parallel breakdown is the only extra (the per-GPU operating point, "
                         "e.g. --batch 8192 under one RCCL rank)")
    ap.add_argument("--dry-run-backend", default=None, help=argparse.SUPPRESS)  # CPU test of the launcher path (gloo)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        return _respawn_under_launcher(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = None
    if args.dry_run_backend:  # no GPU: prove that N ranks start, rendezvous and agree on the world size
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dry_run_backend)
        t = torch.ones(1)
        dist.all_reduce(t)
        assert dist.get_world_size() == args.gpus and int(t.item()) == args.gpus
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": dist.get_world_size(), "ranks_joined": int(t.item())}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return 0
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback on the product path)"
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run always take the RCCL path (also with 1 rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(dev)
    joined = world
    if dist is not None:  # count the ranks that actually joined the RCCL communicator
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        joined = int(t.item())
        assert joined == args.gpus, f"{joined} RCCL ranks joined, expected {args.gpus}"

    import dib_amd  # noqa: F401
    if args.tuning:
        from dib_amd import _lib as _dib_lib
        for kv in args.tuning:
            k, v = kv.split("=")
            _dib_lib.set_tuning(k, int(v))
    if args.config5_only:
        assert world == 1
        print(json.dumps(config5_set_transformer(dev, steps=max(2, min(args.steps, 6)))), flush=True)
        return 0
    wl = Workload(args.features, dev, rank, world, dist, args.scaling, args.batch, args.dp_buckets)
    flops_per_sample = gemm_flops_per_sample(args.features)
    eng = wl.eng
    med, times = wl.measure(args.warmup, args.steps, args.blocks, dev)

    # one more block with HIP events around every MFMA kernel (roofline of the dominant kernel)
    prof, t_prof = None, None
    if not args.no_kernel_timing:
        eng.profile_enable(True)
        t_prof = wl.timed_block(args.warmup + args.blocks * args.steps, args.steps, dev)
        prof = eng.profile_summary()
        eng.profile_enable(False)

    out = None
    if rank == 0:  # the headline line, assembled before any extra touches the workload object
        gb, B = wl.gb, wl.B
        sps = args.steps * gb / med
        per_gpu_tf = sps * flops_per_sample / 1e12 / world
        out = {"metric": "DIB train samples/sec (fwd+KL+bwd+Adam)", "value": round(sps, 1), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * med / args.steps, 4), "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": ("BASELINE config 3" if args.features == 64 else
                                       "BASELINE config 4 (--features 50)" if args.features == 50 else f"--features {args.features}") +
                                      f": synthetic tabular, 2^20 rows x {args.features} scalar features resident in HBM, "
                                      "posenc [2,4,8,16], encoder [128,128], E=32, integration [256,256], out=1, "
                                      "BCE-from-logits, Adam",
                          "per_gpu_batch": B, "global_batch": gb, "parallelism": f"dp{world}", "rccl_ranks_joined": joined,
                          "dataset_rows": N_ROWS, "params": eng.n_params, "flops_per_sample": flops_per_sample},
               "timing": {"protocol": f"median of {args.blocks} blocks x {args.steps} steps, barrier + synchronize on both "
                                      "sides of every block, max over ranks, no per-kernel events",
                          "blocks_ms_per_step": [round(1e3 * t / args.steps, 4) for t in times]},
               "step_roofline": {"bound": "mfma", "achieved": round(per_gpu_tf, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                                 "unit": "TFLOP/s", "frac": round(per_gpu_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                 "note": "whole-step algorithmic GEMM FLOPs / wall time, per GPU"}}
        if prof:
            out["ms_per_step_kernel_timing"] = round(1e3 * t_prof / args.steps, 4)
            per, rest = per_kernel_roofline(prof, args.features, B, args.steps)
            if per:
                dom = max(per, key=lambda k: per[k]["ms_per_step"])
                out["roofline"] = {"bound": "mfma", "achieved": per[dom]["achieved"], "peak": PEAK_F32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": per[dom]["frac"],
                                   "traffic": HBM_TRAFFIC.get(dom) if (B, args.features) == (GLOBAL_BATCH, 64) else None,
                                   "traffic_source": TRAFFIC_SOURCE,
                                   "kernel": dom, "avg_launch_ms": per[dom]["avg_launch_ms"],
                                   "launches": per[dom]["launches"], "flops_per_launch": per[dom]["flops_per_launch"]}
                out["roofline_by_kernel"] = per
                out["mfma_kernels_ms_per_step"] = round(sum(v["ms_per_step"] for v in per.values()), 4)
            if rest:  # tile shapes the rule picks at other batch sizes (no per-symbol FLOP split for them)
                out["other_timed_kernels_ms_per_step"] = rest
        if "roofline" not in out:
            out["roofline"] = dict(out["step_roofline"], traffic=None)
    extra = {}
    if not args.no_extra and (world > 1 or (args.force_dp_extras and dist is not None)):
        deadline = _ExtrasDeadline(args.extra_timeout, rank, out, extra)
        deadline.start()
        try:  # exposed communication of the headline configuration
            extra["dp_breakdown"] = wl.dp_breakdown(max(5, args.steps // 2), dev)
        except Exception as e:  # noqa: BLE001
            extra["dp_breakdown"] = {"error": f"{type(e).__name__}: {e}"}
        try:  # the other scaling mode, same engine
            if not args.only_dp_breakdown:
                other = "weak" if args.scaling == "strong" else "strong"
                wl.set_scaling(other, args.batch)
                m2, t2 = wl.measure(2, args.steps, 1, dev)
                extra[f"{other}_scaling"] = {"value": round(args.steps * wl.gb / m2, 1), "unit": "samples/s",
                                             "ms_per_step": round(1e3 * m2 / args.steps, 4), "per_gpu_batch": wl.B,
                                             "global_batch": wl.gb}
                wl.set_scaling(args.scaling, args.batch)
        except Exception as e:  # noqa: BLE001
            extra["other_scaling"] = {"error": f"{type(e).__name__}: {e}"}
        # BASELINE config 5 under data parallelism (neighbourhoods sharded over the ranks, gradient all-reduce over RCCL inside
        # SetTransformerDIB.train_step): one neighbourhood of 4096 particles per GPU, i.e. never fewer neighbourhoods than ranks
        try:
            if not args.only_dp_breakdown:
                extra["config5_set_transformer"] = dict(config5_set_transformer(dev, nb=max(4, world), steps=3),
                                                        parallelism=f"dp{world} over neighbourhoods")
        except Exception as e:  # noqa: BLE001
            extra["config5_set_transformer"] = {"error": f"{type(e).__name__}: {e}"}
        deadline.cancel()

    if args.only_dp_breakdown:
        args.no_extra = True
    if rank == 0:
        if world == 1 and not args.no_extra:
            # BASELINE config 4 (amorphous-plasticity radial density, 50 shell features; the notebook and its data are a
            # missing blob in the reference, so x ~ N(0,1) [N, 50] per SURVEY 8d), same step, same batch
            del wl
            torch.cuda.empty_cache()
            w4 = Workload(50, dev, 0, 1, None, "strong", args.batch, args.dp_buckets)
            k4 = max(4, args.steps // 2)
            m4, t4 = w4.measure(8, k4, 3, dev)   # median of 3 blocks, like the headline (8 warm-up steps: the first block after the
                                                 # engine swap measured 2x slow with 3 - fresh workspace pages)
            fl4 = gemm_flops_per_sample(50)
            sps4 = k4 * w4.gb / m4
            extra["config4_F50"] = {"workload": "BASELINE config 4: 50 shell features (synthetic N(0,1)), same architecture",
                                    "value": round(sps4, 1), "unit": "samples/s", "ms_per_step": round(1e3 * m4 / k4, 4),
                                    "steps": k4, "blocks_ms_per_step": [round(1e3 * t / k4, 4) for t in t4], "batch": w4.gb, "params": w4.eng.n_params, "flops_per_sample": fl4,
                                    "step_roofline_frac": round(sps4 * fl4 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
            if not args.no_kernel_timing:   # the same per-kernel table as the headline (HIP events inside the library)
                w4.eng.profile_enable(True)
                tp4 = w4.timed_block(8 + 3 * k4, k4, dev)
                per4, rest4 = per_kernel_roofline(w4.eng.profile_summary(), 50, w4.gb, k4)
                w4.eng.profile_enable(False)
                extra["config4_F50"].update(ms_per_step_kernel_timing=round(1e3 * tp4 / k4, 4), roofline_by_kernel=per4,
                                            mfma_kernels_ms_per_step=round(sum(v["ms_per_step"] for v in per4.values()), 4),
                                            other_timed_kernels_ms_per_step=rest4)
            del w4
            torch.cuda.empty_cache()
            try:
                extra["fit_surface"] = fit_surface(dev, args.batch, med / args.steps)
            except Exception as e:  # noqa: BLE001
                extra["fit_surface"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_extra:
            # BASELINE config 5: per-particle set-transformer DIB, 4096 particles per neighbourhood, 3-D positions
            # (16 derived per-particle features), the notebook's architecture; one step = fwd + KL + BCE + bwd + Adam
            try:
                extra["config5_set_transformer"] = config5_set_transformer(dev)
            except Exception as e:  # noqa: BLE001 - the extra line must never take the headline down
                extra["config5_set_transformer"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                extra["set_transformer_notebook_size"] = reference_size_set_transformer(dev)
            except Exception as e:  # noqa: BLE001
                extra["set_transformer_notebook_size"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                extra["keras_path_default_batch"] = keras_path_default_batch(dev)
            except Exception as e:  # noqa: BLE001
                extra["keras_path_default_batch"] = {"error": f"{type(e).__name__}: {e}"}
            try:   # the reference's default batch (train.py:34) and the chaos notebook's (Chaos_experiments.ipynb:771-821)
                extra["config2_infonce_loop"] = {
                    "workload": "BASELINE config 2 path: custom InfoNCE loop, pendulum layout [2,1,2,1], shared space 64, l2, fp32",
                    "runs": [config2_infonce_loop(dev, 128), config2_infonce_loop(dev, 2048)]}
            except Exception as e:  # noqa: BLE001
                extra["config2_infonce_loop"] = {"error": f"{type(e).__name__}: {e}"}
        if extra:
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
