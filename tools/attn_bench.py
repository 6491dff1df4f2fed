"""Kernel-level timing of the flash attention entry points (dib_attention_fwd / dib_attention_bwd):
    python tools/attn_bench.py [--batch 4 --particles 4096 --heads 12]
Prints ms per call and the algorithmic TFLOP/s (forward: 2 products of 2*P*P*128 FLOP per (neighbourhood, head); backward: 4)."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--particles", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--stash", type=int, default=1, help="1: forward stashes the score tiles, backward reads them; 0: recompute")
    a = ap.parse_args()
    import dib_amd  # noqa: F401
    from dib_amd._lib import check, load_library
    lib = load_library()
    B, P, H, D = a.batch, a.particles, a.heads, 128
    T, ld = B * P, H * D
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda: (torch.randn((T, ld), generator=g) * 0.5).to(dev)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, dq, dk, dv = (torch.empty_like(q) for _ in range(4))
    lse = torch.empty(B * H * P, device=dev)
    ws = torch.empty(int(lib.dib_attention_bwd_workspace_bytes(B, P, H)) // 4, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale = 1.0 / D ** 0.5
    stash = torch.empty(int(lib.dib_attention_stash_bytes(B, P, H)) // 4, device=dev) if a.stash else None
    sp = p(stash) if a.stash else ctypes.c_void_p(0)
    fwd = lambda: check(lib.dib_attention_fwd(p(q), p(k), p(v), B, P, H, D, ld, scale, p(o), p(lse), sp, st), "fwd")
    bwd = lambda: check(lib.dib_attention_bwd(p(q), p(k), p(v), p(o), p(do), p(lse), sp, B, P, H, D, ld, scale, p(dq), p(dk), p(dv),
                                              p(ws), st), "bwd")
    out = {"B": B, "P": P, "H": H, "score_stash": bool(a.stash)}
    for name, fn, units in (("fwd", fwd, 2), ("bwd", bwd, 4)):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        out[name + "_ms"] = round(ms, 3)
        out[name + "_TFLOPs"] = round(units * 2.0 * P * P * D * B * H / ms / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
