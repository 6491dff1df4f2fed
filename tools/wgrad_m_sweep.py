"""BASELINE config 4's slow weight gradient: dW[M, 256] = U[B, M]^T @ G[B, 256] through the raw grouped
entry (include/dib_st.h dib_gemm_grouped, mode 2) at B = 65536 - M = 1600 (config 4: 50 features x E = 32) against 1536 /
1664 / 2048 (config 3), each with the row pitch of U equal to M and padded to 2048, at the split counts the layout rule and
the sweep of round 4 used.  Separates the suspects: the 6400-byte row pitch, the half-empty 13th m-tile, the split count
against the 8 XCDs.  One JSON line per case: ms per launch (median of 5 x 10 launches), TFLOP/s, fraction of the fp32-MFMA peak.
    python tools/wgrad_m_sweep.py [--batch 65536]"""
import argparse
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--n", type=int, default=256)
    a = ap.parse_args()
    from dib_amd import _lib
    from dib_amd._gemm_plan import _Gemm, _d
    lib = _lib.load_library()
    dev = torch.device("cuda:0")
    B, N = a.batch, a.n
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    G = torch.randn(B, N, device=dev)
    cases = []
    for M, ld in ((1600, 1600), (1600, 2048), (1600, 1664), (1536, 1536), (1536, 1600), (1664, 1664), (2048, 2048), (1024, 1024)):
        for ns in (16, 19, 24, 32):
            cases.append((M, ld, ns))
    U = torch.randn(B, 2048, device=dev)
    for M, ld, ns in cases:
        rps = (B + ns - 1) // ns
        rps = (rps + 63) // 64 * 64
        ns_eff = (B + rps - 1) // rps
        stride = M * N + N
        stride = (stride + 3) // 4 * 4
        slabs = torch.zeros(ns_eff * stride, device=dev)
        Uv = U.view(-1)[: B * ld]
        g = _Gemm(2, [_d(0, ld, 0, N, 0, N, M, N, B, bias_off=M * N)], Uv, G, slabs, bias_out=slabs, nsplit=ns_eff, rows_per_split=rps,
                  split_stride=stride)
        g.upload(dev)
        for _ in range(3):
            g.run(lib, st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.run(lib, st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ms = statistics.median(ts)
        tf = 2.0 * B * M * N / ms / 1e9
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        print(json.dumps({"M": M, "ld_U": ld, "N": N, "B": B, "splits": ns_eff, "rows_per_split": rps, "workgroups_128x128": tiles * ns_eff,
                          "ms": round(ms, 4), "TFLOPs": round(tf, 1), "frac_of_peak": round(tf / PEAK, 3)}), flush=True)
        del slabs


if __name__ == "__main__":
    main()
