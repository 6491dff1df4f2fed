"""DenseStack: a plain MLP ([PositionalEncoding] -> Dense(units, act)* -> Dense(out)) on the same hand-written
gfx950 GEMM kernels (dib_gemm through the C ABI), used for the InfoNCE path's output encoder
(reference train.py:184-192: `output_encoder`).  Forward, backward and Keras-Adam all run on the device."""
from __future__ import annotations

import math
from ctypes import c_void_p
from typing import List, Optional, Sequence

import numpy as np
import torch

from ._lib import ACTIVATIONS, check


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


class DenseStack:
    def __init__(self, engine, input_dim: int, units: Sequence[int], output_dim: int, activation: Optional[str] = "relu",
                 use_positional_encoding: bool = True, number_positional_encoding_frequencies: int = 5, seed: int = 0):
        self.eng, self.lib, self.device = engine, engine.lib, engine.device
        self.act = ACTIVATIONS[activation]
        self.n_freq = int(number_positional_encoding_frequencies) if use_positional_encoding else 1
        self.input_dim = int(input_dim)
        dims = [self.input_dim * max(self.n_freq, 1)] + [int(u) for u in units] + [int(output_dim)]
        self.dims = list(zip(dims[:-1], dims[1:]))
        off, self.w_off, self.b_off = 0, [], []
        for i, o in self.dims:
            self.w_off.append(off); off += (i * o + 3) // 4 * 4
            self.b_off.append(off); off += (o + 3) // 4 * 4
        self.n_params = off
        rng = np.random.default_rng(seed)
        flat = np.zeros(off, dtype=np.float32)
        for (i, o), w in zip(self.dims, self.w_off):  # Keras glorot-uniform kernels, zero biases
            lim = math.sqrt(6.0 / (i + o))
            flat[w: w + i * o] = rng.uniform(-lim, lim, i * o).astype(np.float32)
        z = lambda: torch.zeros(off, dtype=torch.float32, device=self.device)
        self.params = torch.from_numpy(flat).to(self.device)
        self.grads, self.adam_m, self.adam_v = z(), z(), z()
        self.t_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.lr_dev = torch.full((1,), 1e-3, dtype=torch.float32, device=self.device)
        self._desc = torch.zeros(256, dtype=torch.uint8, device=self.device)
        self._acts: List[torch.Tensor] = []

    # views
    def kernel(self, l):
        i, o = self.dims[l]
        return self.params[self.w_off[l]: self.w_off[l] + i * o].view(i, o)

    def bias(self, l):
        return self.params[self.b_off[l]: self.b_off[l] + self.dims[l][1]]

    def _gemm(self, mode, M, N, K, A, lda, B, ldb, C, ldc, bias, aux, ldaux, act):
        check(self.lib.dib_gemm(mode, M, N, K, _ptr(A), lda, _ptr(B), ldb, _ptr(C), ldc, _ptr(bias), _ptr(aux), ldaux, act,
                                _ptr(self._desc), self.eng._stream()), "dib_gemm")

    def forward(self, y: torch.Tensor) -> torch.Tensor:
        y = y.to(device=self.device, dtype=torch.float32).contiguous()
        n = y.shape[0]
        if self.n_freq > 1:
            h = torch.empty((n, self.input_dim * self.n_freq), dtype=torch.float32, device=self.device)
            check(self.lib.dib_positional_encoding(_ptr(y), y.stride(0), n, self.input_dim, self.n_freq, _ptr(h),
                                                   self.eng._stream()), "dib_positional_encoding")
        else:
            h = y
        self._acts = [h]
        L = len(self.dims)
        for l, (i, o) in enumerate(self.dims):
            out = torch.empty((n, o), dtype=torch.float32, device=self.device)
            self._gemm(0, n, o, i, h, i, self.kernel(l), o, out, o, self.bias(l), None, 0, self.act if l < L - 1 else 0)
            self._acts.append(out)
            h = out
        return h

    def backward(self, g_out: torch.Tensor) -> None:
        """grads <- d loss / d params given d loss / d output (overwrites self.grads)."""
        g = g_out.contiguous()
        n = g.shape[0]
        for l in reversed(range(len(self.dims))):
            i, o = self.dims[l]
            h_in = self._acts[l]
            gw = self.grads[self.w_off[l]: self.w_off[l] + i * o]
            gb = self.grads[self.b_off[l]: self.b_off[l] + o]
            # wgrad: dW[i,o] = h_in[n,i]^T @ g[n,o], bias gradient = column sums of g
            self._gemm(2, i, o, n, h_in, i, g, o, gw, o, gb, None, 0, 0)
            if l > 0:
                gi = torch.empty((n, i), dtype=torch.float32, device=self.device)
                self._gemm(1, n, i, o, g, o, self.kernel(l), o, gi, i, None, h_in, i, self.act)
                g = gi

    def adam_step(self, lr: float, beta1=0.9, beta2=0.999, eps=1e-7) -> None:
        self.lr_dev.fill_(float(lr))
        check(self.lib.dib_adam_step(_ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                     self.n_params, _ptr(self.lr_dev), _ptr(self.t_dev), beta1, beta2, eps, 1.0,
                                     self.eng._stream()), "dib_adam_step")
