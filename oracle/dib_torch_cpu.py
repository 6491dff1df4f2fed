"""CPU baseline / second checker: the reference graph restated with PyTorch-CPU eager ops.
TEST / BENCH INFRASTRUCTURE ONLY (see oracle/dib_oracle.py header) - never imported by the product.

"CPU restatement of the TF graph (PyTorch-CPU eager), not TensorFlow" (BASELINE.md section 3): the same
per-feature-loop structure as reference models.py:105-122 (F separate Dense chains, concat,
integration MLP), loss + beta*sum KL (models.py:118), autograd backward, Keras-form Adam
(eps=1e-7).  TensorFlow itself is not installable here.  Because the backward is
autograd (not the hand-derived one of dib_oracle.backward) it also cross-checks the numpy oracle.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch

import dib_oracle as orc


def _act(name: Optional[str]):
    return {None: lambda z: z, "linear": lambda z: z, "relu": torch.relu, "tanh": torch.tanh,
            "sigmoid": torch.sigmoid, "leaky_relu": lambda z: torch.nn.functional.leaky_relu(z, 0.2),
            "elu": torch.nn.functional.elu, "softplus": torch.nn.functional.softplus}[name]


class TorchCpuDIB:
    def __init__(self, spec: orc.DIBSpec, params: orc.DIBParams, dtype=torch.float32):
        self.spec, self.dtype = spec, dtype
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True)
        self.enc_W = [[t(w) for w in ws] for ws in params.enc_W]
        self.enc_b = [[t(b) for b in bs] for bs in params.enc_b]
        self.int_W = [t(w) for w in params.int_W]
        self.int_b = [t(b) for b in params.int_b]
        self.m = [torch.zeros_like(p) for p in self.tensors()]
        self.v = [torch.zeros_like(p) for p in self.tensors()]
        self.t = 0
        self.freqs = [float(f) for f in spec.frequencies]

    def tensors(self) -> List[torch.Tensor]:
        out = []
        for ws, bs in zip(self.enc_W, self.enc_b):
            for w, b in zip(ws, bs):
                out += [w, b]
        for w, b in zip(self.int_W, self.int_b):
            out += [w, b]
        return out

    def forward(self, x: torch.Tensor, eps: torch.Tensor):
        """reference models.py:96-123 with eps [B,F,E] injected."""
        s = self.spec
        E = s.feature_embedding_dimension
        act = _act(s.activation_fn)
        feats = torch.split(x, list(s.feature_dimensionalities), dim=-1)
        us, kls = [], []
        for f in range(s.number_features):
            h = feats[f]
            if s.use_positional_encoding:
                h = torch.cat([h] + [torch.sin(fr * h) for fr in self.freqs], -1)
            n = len(self.enc_W[f])
            for l in range(n):
                h = h @ self.enc_W[f][l] + self.enc_b[f][l]
                if l < n - 1:
                    h = act(h)
            mu, lv = h[:, :E], h[:, E:]
            us.append(mu + torch.exp(lv / 2.0) * eps[:, f])
            kls.append(torch.mean(torch.sum(0.5 * (mu * mu + torch.exp(lv) - lv - 1.0), -1)))
        h = torch.cat(us, -1)
        n = len(self.int_W)
        for l in range(n):
            h = h @ self.int_W[l] + self.int_b[l]
            h = act(h) if l < n - 1 else _act(s.output_activation_fn)(h)
        return h, torch.stack(kls)

    def loss(self, kind: str, y: torch.Tensor, pred: torch.Tensor):
        if kind == "bce_logits":
            return torch.nn.functional.binary_cross_entropy_with_logits(pred, y.to(pred.dtype).view_as(pred))
        if kind == "mse":
            return torch.mean((pred - y.to(pred.dtype).view_as(pred)) ** 2)
        if kind == "sparse_cce_logits":
            return torch.nn.functional.cross_entropy(pred, y.view(-1).long())
        raise ValueError(kind)

    def train_step(self, x, y, eps, beta: float, kind: str, lr: float = 3e-4):
        pred, kl = self.forward(x, eps)
        task = self.loss(kind, y, pred)
        total = task + beta * kl.sum()
        ps = self.tensors()
        grads = torch.autograd.grad(total, ps)
        self.t += 1
        b1, b2, e = 0.9, 0.999, 1e-7
        lr_t = lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        with torch.no_grad():
            for p, g, m, v in zip(ps, grads, self.m, self.v):
                m.add_((1 - b1) * (g - m))
                v.add_((1 - b2) * (g * g - v))
                p.sub_(lr_t * m / (torch.sqrt(v) + e))
        return float(task.detach()), kl.detach(), grads
