"""Phase timing of the fused encoder-bank forward (diagnostic).  The PRODUCT library must be a -DDIB_FUSED_TIMING build
(tools/runs/*.sh copy exp/lib_FTIMING.so over it for the duration of the run):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDIB_FUSED_TIMING <pkg>/csrc/dib_api.hip -o exp/lib_FTIMING.so
Prints, for wave 0 of workgroup (0, 0), the shader cycles per 32-sample tile spent in each phase of the tile loop."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=60, help="training steps before the timers are read (the last forward is reported)")
    a = ap.parse_args()
    import bench
    import dib_amd  # noqa: F401
    from dib_amd._lib import load_library
    lib = load_library()
    dev = torch.device("cuda:0")
    wl = bench.Workload(64, dev, 0, 1, None, "strong", a.batch, 2)
    for i in range(a.steps):
        if i == a.steps - 1 and hasattr(lib, "dib_fused_debug_reset"):
            lib.dib_fused_debug_reset()
        wl.step(i)
    torch.cuda.synchronize()
    if hasattr(lib, "dib_fused_debug_read_tl"):
        tl = (ctypes.c_longlong * 4)()
        assert lib.dib_fused_debug_read_tl(tl) == 0
        big = 0x7fffffffffffffff
        pos_end, fwd_in, fwd_out, nxt_in = tl[0], big - tl[1], tl[2], big - tl[3]
        print(f"timeline (us): last posenc workgroup mark -> first fused-forward wave {(fwd_in - pos_end) / 100.0:.1f}; "
              f"fused forward first wave -> last workgroup done {(fwd_out - fwd_in) / 100.0:.1f}; "
              f"last workgroup done -> first wave of the next kernel {(nxt_in - fwd_out) / 100.0:.1f}")
    out = (ctypes.c_longlong * 16)()
    assert lib.dib_fused_debug_read(out) == 0
    n = max(1, out[10])
    names = ["prefetch + layer 1", "stash h1", "layer 2 + act", "stash h2", "mask bits", "layer 3", "eps, sigma, u, KL",
             "stash mu|logvar, u", "KL sum, loop end"]
    tot = sum(out[i] for i in range(9))
    mfma = {0: 32, 2: 256, 5: 128}
    for i, nm in enumerate(names):
        fl = f"(MFMA floor {mfma[i] * 64})" if i in mfma else ""
        print(f"{nm:22s} {out[i] / n:9.0f} cycles/tile  {100.0 * out[i] / tot:5.1f} %  {fl}")
    for i, nm in ((12, "(pp) barrier wait"), (13, "(pp) input loads"), (14, "(pp) stash h2"), (15, "(pp) KL wave sum")):
        if out[i]:
            print(f"{nm:22s} {out[i] / n:9.0f} cycles/tile")
    tot += sum(out[12:16])
    print(f"{'loop total':22s} {tot / n:9.0f} cycles/tile   ({n} tiles per wave, kernel body {out[9]} cycles; own MFMA floor 416 x 64 = 26624, "
          f"the SIMD's two waves share one pipe)")
    wg = (ctypes.c_longlong * 3072)()
    if hasattr(lib, "dib_fused_debug_read_wg") and lib.dib_fused_debug_read_wg(wg) == 0:
        import numpy as np
        t = np.array(list(wg), dtype=np.int64).reshape(1024, 3)[:256].astype(np.float64) / 100.0   # microseconds
        t0 = t[:, 0].min()
        print(f"per-workgroup (256 persistent workgroups), microseconds after the first workgroup's entry:")
        for nm, col in (("kernel entry", 0), ("weights staged", 1), ("tile loop done", 2)):
            v = t[:, col] - t0
            print(f"  {nm:16s} min {v.min():8.1f}  p10 {np.percentile(v, 10):8.1f}  median {np.median(v):8.1f}  p90 {np.percentile(v, 90):8.1f}  max {v.max():8.1f}")
        d = t[:, 2] - t[:, 1]
        print(f"  loop duration    min {d.min():8.1f}  median {np.median(d):8.1f}  max {d.max():8.1f}")
        wv = (ctypes.c_longlong * 8192)()
        if hasattr(lib, "dib_fused_debug_read_waves") and lib.dib_fused_debug_read_waves(wv) == 0:
            w = np.array(list(wv), dtype=np.int64).reshape(1024, 8)[:256].astype(np.float64) / 100.0 - t0
            print("  loop done per wave index (median / max over the 256 workgroups):")
            for i in range(8):
                print(f"    wave {i}: {np.median(w[:, i]):8.1f} / {w[:, i].max():8.1f}")
        order = np.argsort(t[:, 2])
        print("  slowest workgroups (feature, column):", [(int(i) // 4, int(i) % 4) for i in order[-8:]])
        print("  fastest workgroups (feature, column):", [(int(i) // 4, int(i) % 4) for i in order[:8]])
    if out[11] > 0:
        print(f"shader clock during the loop: {out[9] / (out[11] / 100e6) / 1e9:.3f} GHz (s_memtime ticks / s_memrealtime at 100 MHz)")


if __name__ == "__main__":
    main()
