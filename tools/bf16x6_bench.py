"""GPU: the experimental split-bf16 fp32 GEMM (csrc/dib_gemm_bf16x6.h) against the fp32-MFMA kernel on the integration
network's first layer of BASELINE config 3: [65536, 2048] x [2048, 256] + bias + ReLU."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dib_amd  # noqa: E402,F401
from dib_amd._lib import check, load_library  # noqa: E402

lib = load_library()
dev = torch.device("cuda:0")
M, K, N = 65536, 2048, 256
g = torch.Generator(device="cpu").manual_seed(0)
A = torch.randn((M, K), generator=g).to(dev)
W = (torch.randn((K, N), generator=g) / K ** 0.5).to(dev)
b = torch.randn(N, generator=g).to(dev)
C1 = torch.empty((M, N), device=dev)
C2 = torch.empty((M, N), device=dev)
planes = torch.empty(int(lib.dib_split_weights_bytes(K, N)), dtype=torch.uint8, device=dev)
desc = torch.zeros(256, dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())


def fp32():
    check(lib.dib_gemm(0, M, N, K, p(A), K, p(W), N, p(C1), N, p(b), None, 0, 1, p(desc), st), "dib_gemm")


def x6():
    check(lib.dib_split_weights(p(W), K, N, p(planes), st), "split")
    check(lib.dib_gemm_bf16x6(M, N, K, p(A), K, p(planes), p(C2), N, p(b), 1, st), "x6")


def x6_presplit():
    check(lib.dib_gemm_bf16x6(M, N, K, p(A), K, p(planes), p(C2), N, p(b), 1, st), "x6")


for name, fn in (("fp32 MFMA (dib_gemm mode 0)", fp32), ("bf16x6 (incl. weight split)", x6),
                 ("bf16x6 (weights pre-split)", x6_presplit)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:32s} {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s fp32-equivalent")
ref = (A[:2048].double() @ W.double() + b.double()).clamp(min=0)
print("max |err| vs float64 on 2048 rows: fp32 %.3e  bf16x6 %.3e  (max|C| %.2f)" % (
    (C1[:2048].double() - ref).abs().max().item(), (C2[:2048].double() - ref).abs().max().item(), ref.abs().max().item()))
