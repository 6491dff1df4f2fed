"""Step time of the per-particle set-transformer DIB (BASELINE config 5) on one MI355X:
    python tools/set_transformer_bench.py [--batch 32 --particles 50] [--steps 10]
Prints one JSON line: ms/step (fwd + KL + loss + bwd + Adam), neighbourhoods/s, algorithmic GEMM TFLOP/s (3 x forward
FLOPs of oracle.flops_per_neighbourhood: q/k/v/o projections, Q K^T, P V, feed-forward, encoder, head)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fwd_flops(m, P):
    fl, d = 0, m.particle_feature_dimensions * m.number_positional_encoding_frequencies
    for u in m.particle_encoder_arch_spec + [2 * m.bottleneck_dimension]:
        fl += 2 * d * u * P
        d = u
    D, H, K = m.bottleneck_dimension, m.number_heads_per_mha, m.key_dim
    per = 3 * 2 * D * H * K * P + 2 * 2 * H * K * P * P + 2 * H * K * D * P
    d = D
    for u in m.ff_arch_per_block:
        per += 2 * d * u * P
        d = u
    return fl + m.number_attention_blocks * per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--particles", type=int, default=50)
    ap.add_argument("--features", type=int, default=12)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--graphs", type=int, default=0, help="1: one hipGraph replay per step (SetTransformerDIB(use_graphs=True))")
    ap.add_argument("--chain", type=int, default=1, help="0: the layer-by-layer launches instead of the token-chain kernels (A/B)")
    ap.add_argument("--defer", type=int, default=1, help="0: per-block weight-gradient launches instead of the deferred grouped ones (A/B)")
    ap.add_argument("--attn-proj", type=int, default=1, help="0: q / k / v projections as a launch of their own in front of the attention (A/B)")
    ap.add_argument("--attn-bwd-proj", type=int, default=1, help="0: the projections' input gradient as a split-K GEMM per block (A/B)")
    ap.add_argument("--max-slabs", type=int, default=0, help="cap of the weight-gradient slab count in deferred mode (0: the default)")
    ap.add_argument("--defer-target", type=int, default=0, help="workgroups per deferred weight-gradient launch (0: the default)")
    ap.add_argument("--tuning", default="", help="dib_set_tuning keys, e.g. attn_small_waves=8 (A/B)")
    a = ap.parse_args()
    import dib_amd
    from dib_amd import _lib
    for kv in filter(None, a.tuning.split(",")):
        k, v = kv.split("=")
        _lib.set_tuning(k.strip(), int(v))
    m = dib_amd.SetTransformerDIB(particle_feature_dimensions=a.features, attention=os.environ.get("DIB_ST_ATTENTION", "auto"),
                                  use_graphs=bool(a.graphs))
    m.use_chain = bool(a.chain)
    m.defer_wgrads = bool(a.defer)
    m.attention_proj = bool(a.attn_proj)
    m.attention_bwd_proj = bool(a.attn_bwd_proj)
    if a.max_slabs:
        m.deferred_max_slabs = a.max_slabs
    if a.defer_target:
        m.deferred_wgrad_target_wgs = a.defer_target
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((a.batch, a.particles, a.features)).astype(np.float32)).to(m.device)
    y = torch.from_numpy((rng.random((a.batch, 1)) > 0.5).astype(np.float32)).to(m.device)
    m.beta_dev.fill_(1e-3)
    m.lr_dev.fill_(1e-4)
    for _ in range(a.warmup):
        m.train_step(x, y)
    torch.cuda.synchronize()
    n0 = _lib.load_library().dib_launch_count()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.train_step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    launches = (_lib.load_library().dib_launch_count() - n0) / a.steps
    fl = 3 * fwd_flops(m, a.particles) * a.batch
    print(json.dumps({"workload": f"set-transformer DIB, {a.batch} neighbourhoods x {a.particles} particles x {a.features} features",
                      "ms_per_step": round(1e3 * dt, 3), "neighbourhoods_per_s": round(a.batch / dt, 1),
                      "algorithmic_TFLOPs": round(fl / dt / 1e12, 2), "params": m.n_params,
                      "attention": m.attention_impl, "graph_replay": bool(a.graphs), "tuning": a.tuning,
                      "library_launches_per_step": round(launches, 1), "defer_wgrads": bool(a.defer),
                      "defer_target_wgs": a.defer_target or None, "max_slabs": a.max_slabs or None, "attention_proj": bool(a.attn_proj), "attention_bwd_proj": bool(a.attn_bwd_proj)}))


if __name__ == "__main__":
    main()
