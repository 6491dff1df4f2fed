"""Phase timing of the flash-attention backward (diagnostic).  Needs a library built with -DDIB_ATTN_TIMING:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDIB_ATTN_TIMING <pkg>/csrc/dib_api.hip -o exp/lib_TIMING.so
    DIB_LIB=exp/lib_TIMING.so python tools/attn_phase_timing.py [--batch 4 --particles 4096]
Prints, for wave 0 of workgroup (1, 0, 0), the shader cycles per query tile spent in each phase of the loop."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--particles", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=12)
    a = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(ROOT, os.environ.get("DIB_LIB", "exp/lib_TIMING.so")))
    B, P, H, D = a.batch, a.particles, a.heads, 128
    T, ld = B * P, H * D
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda: (torch.randn((T, ld), generator=g) * 0.5).to(dev)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, dq, dk, dv = (torch.empty_like(q) for _ in range(4))
    lse = torch.empty(B * H * P, device=dev)
    lib.dib_attention_bwd_workspace_bytes.restype = ctypes.c_int64
    ws = torch.empty(int(lib.dib_attention_bwd_workspace_bytes(B, P, H)) // 4, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    i64, f32 = ctypes.c_int64, ctypes.c_float
    scale = 1.0 / D ** 0.5
    assert lib.dib_attention_fwd(p(q), p(k), p(v), B, P, H, D, i64(ld), f32(scale), p(o), p(lse), ctypes.c_void_p(0), st) == 0
    for _ in range(3):
        assert lib.dib_attention_bwd(p(q), p(k), p(v), p(o), p(do), p(lse), ctypes.c_void_p(0), B, P, H, D, i64(ld), f32(scale), p(dq), p(dk), p(dv),
                                     p(ws), st) == 0
    out = (ctypes.c_longlong * 16)()
    assert lib.dib_attn_debug_read(out) == 0
    n = max(1, out[8])
    names = ["barrier A", "S/dP products", "exp + dS + patch", "dV/dK products", "barrier B", "loads + dQ + store", "tile -> LDS"]
    tot = sum(out[i] for i in range(7))
    for i, nm in enumerate(names):
        print(f"{nm:20s} {out[i] / n:9.0f} cycles/tile  {100.0 * out[i] / tot:5.1f} %")
    print(f"{'loop total':20s} {tot / n:9.0f} cycles/tile   (kernel body {out[7]} cycles, {n} tiles; MFMA floor 320 x 64 = 20480)")
    if out[9] > 0:
        print(f"shader clock during the loop: {out[7] / (out[9] / 100e6) / 1e9:.3f} GHz (s_memtime ticks / s_memrealtime at 100 MHz)")


if __name__ == "__main__":
    main()
