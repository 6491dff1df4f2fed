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


class _MaskedReLU(torch.autograd.Function):
    """relu(z) whose backward uses a GIVEN subgradient choice (mask) instead of its own `z > 0`.  ReLU' is discontinuous:
    a float32 device and this float64 restatement legitimately disagree about the sign of a pre-activation that sits
    within round-off of 0, and at B = 65536 a handful of the 10^8 units always do - each flips one sample's whole
    contribution to a weight-gradient column (1e-3 of the column's scale).  The full-size parity tests therefore compare
    gradients under the device's own choices and separately prove that those differ from the float64 choices only where
    the pre-activation is at round-off level."""

    @staticmethod
    def forward(ctx, z, mask):
        ctx.save_for_backward(mask)
        return torch.relu(z)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


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

    def _uniform(self) -> bool:
        s = self.spec
        return len(set(s.feature_dimensionalities)) == 1 and s.number_features > 1

    def forward(self, x: torch.Tensor, eps: torch.Tensor, batched: bool = False, reduce_kl: str = "mean", masks=None,
                pre_out=None):
        """reference models.py:96-123 with eps [B,F,E] injected.

        batched=False: the reference's own structure, a Python loop over F separate Dense chains (this is what the CPU
        baseline times).  batched=True (features of equal width only): the SAME arithmetic with the F chains stacked into
        torch.bmm calls - a faster checker for the full-size parity tests; test_torch_cpu_batched_equals_loop pins it on
        the loop form.  reduce_kl: "mean" = models.py:111-112; "sum" = sum over the rows (callers that chunk the batch).
        masks (batched relu only): {"enc": [bool [F,B,units] per hidden layer], "int": [bool [B,units] per hidden layer]} -
        the backward then uses these act' choices (_MaskedReLU).  pre_out: optional dict that receives the float64
        PRE-activations ("enc": [...], "int": [...]) so a caller can check where the choices differ."""
        s = self.spec
        E = s.feature_embedding_dimension
        act = _act(s.activation_fn)
        red = torch.mean if reduce_kl == "mean" else torch.sum
        if batched and self._uniform():
            F, d = s.number_features, s.feature_dimensionalities[0]
            h = x.view(x.shape[0], F, d).permute(1, 0, 2)                       # [F, B, d] (tf.split, models.py:101)
            if s.use_positional_encoding:
                h = torch.cat([h] + [torch.sin(fr * h) for fr in self.freqs], -1)  # models.py:22-23
            n = len(self.enc_W[0])
            for l in range(n):
                W = torch.stack([self.enc_W[f][l] for f in range(F)])           # [F, in, out]
                b = torch.stack([self.enc_b[f][l] for f in range(F)])[:, None, :]
                h = torch.bmm(h, W) + b
                if l < n - 1:
                    if pre_out is not None:
                        pre_out.setdefault("enc", []).append(h.detach())
                    h = _MaskedReLU.apply(h, masks["enc"][l]) if masks is not None else act(h)
            mu, lv = h[..., :E], h[..., E:]                                      # models.py:106
            u = mu + torch.exp(lv / 2.0) * eps.permute(1, 0, 2)                  # models.py:108
            kl = red(torch.sum(0.5 * (mu * mu + torch.exp(lv) - lv - 1.0), -1), -1)  # [F]   models.py:111-112
            h = u.permute(1, 0, 2).reshape(x.shape[0], F * E)                    # tf.concat, models.py:122
        else:
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
                kls.append(red(torch.sum(0.5 * (mu * mu + torch.exp(lv) - lv - 1.0), -1)))
            h = torch.cat(us, -1)
            kl = torch.stack(kls)
        n = len(self.int_W)
        for l in range(n):
            h = h @ self.int_W[l] + self.int_b[l]
            if l < n - 1 and pre_out is not None:
                pre_out.setdefault("int", []).append(h.detach())
            if l < n - 1 and masks is not None:
                h = _MaskedReLU.apply(h, masks["int"][l])
            else:
                h = act(h) if l < n - 1 else _act(s.output_activation_fn)(h)
        return h, kl

    def loss(self, kind: str, y: torch.Tensor, pred: torch.Tensor, reduction: str = "mean"):
        if kind == "bce_logits":
            return torch.nn.functional.binary_cross_entropy_with_logits(pred, y.to(pred.dtype).view_as(pred),
                                                                        reduction=reduction)
        if kind == "mse":
            d = (pred - y.to(pred.dtype).view_as(pred)) ** 2
            return torch.mean(d) if reduction == "mean" else torch.sum(d) / pred.shape[1]
        if kind == "sparse_cce_logits":
            return torch.nn.functional.cross_entropy(pred, y.view(-1).long(), reduction=reduction)
        raise ValueError(kind)

    def loss_and_grads(self, x, y, eps, beta: float, kind: str, chunk: Optional[int] = None, batched: bool = False,
                       want_grads: bool = True, masks=None, boundary=None):
        """L = mean_b loss + beta * sum_f KL_f (models.py:118) and dL/dparams, evaluated in row chunks (the batch mean is
        linear, so the chunk gradients add up exactly): keeps the autograd tape of a 65536-row float64 batch out of
        memory.  Returns (task loss, kl [F] detached, grads list | None, pred [B, out] detached).
        masks: see forward().  boundary: optional dict; with masks given it receives, per hidden layer, the number of units
        whose given choice differs from the float64 `z > 0` and the largest |z| among those ("enc_l0", "int_l1", ...)."""
        assert masks is None or (batched and self._uniform() and self.spec.activation_fn == "relu")
        B = x.shape[0]
        chunk = B if not chunk else int(chunk)
        ps = self.tensors()
        acc = [torch.zeros_like(p) for p in ps] if want_grads else None
        task_sum, kl_sum, preds = 0.0, torch.zeros(self.spec.number_features, dtype=self.dtype), []
        for s0 in range(0, B, chunk):
            sl = slice(s0, min(B, s0 + chunk))
            mk, pre = None, None
            if masks is not None:
                mk = {"enc": [m[:, sl] for m in masks["enc"]], "int": [m[sl] for m in masks["int"]]}
                pre = {} if boundary is not None else None
            with torch.set_grad_enabled(want_grads):
                pred, kl = self.forward(x[sl], eps[sl], batched=batched, reduce_kl="sum", masks=mk, pre_out=pre)
            if pre:
                for net in ("enc", "int"):
                    for l, z in enumerate(pre.get(net, [])):
                        diff = (z > 0) != mk[net][l]
                        cnt, worst = boundary.get(f"{net}_l{l}", (0, 0.0))
                        boundary[f"{net}_l{l}"] = (cnt + int(diff.sum()), max(worst, float(z[diff].abs().max()) if diff.any() else 0.0))
            with torch.set_grad_enabled(want_grads):
                task = self.loss(kind, y[sl], pred, reduction="sum")
                total = (task + beta * kl.sum()) / B
            if want_grads:
                for a, g in zip(acc, torch.autograd.grad(total, ps)):
                    a.add_(g)
            task_sum += float(task.detach())
            kl_sum += kl.detach()
            preds.append(pred.detach())
        return task_sum / B, kl_sum / B, acc, torch.cat(preds, 0)

    def apply_adam(self, grads, lr: float = 3e-4, b1: float = 0.9, b2: float = 0.999, e: float = 1e-7):
        """Keras Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps), eps = 1e-7."""
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        with torch.no_grad():
            for p, g, m, v in zip(self.tensors(), grads, self.m, self.v):
                m.add_((1 - b1) * (g - m))
                v.add_((1 - b2) * (g * g - v))
                p.sub_(lr_t * m / (torch.sqrt(v) + e))

    def train_step(self, x, y, eps, beta: float, kind: str, lr: float = 3e-4, chunk: Optional[int] = None,
                   batched: bool = False):
        task, kl, grads, _ = self.loss_and_grads(x, y, eps, beta, kind, chunk=chunk, batched=batched)
        self.apply_adam(grads, lr)
        return task, kl, grads


# --------------------------------------------------------------------------------------------
# InfoNCE loss + embedding gradients by float64 autograd (checker of dib_infonce_fwd_bwd at working batch sizes; the numpy
# central-difference oracle dib_oracle.infonce_grads_numeric is O(B D) loss evaluations and only practical at B ~ 12).
# reference: utils.py:75-175 (get_scaled_similarity), train.py:203-215 (eval_batch_infonce).  Pinned on the numpy restatement
# dib_oracle.scaled_similarity / infonce_loss by tests/test_oracle_golden.py.
# --------------------------------------------------------------------------------------------
def scaled_similarity_torch(e1: torch.Tensor, e2: torch.Tensor, similarity_type: str, temperature: float) -> torch.Tensor:
    eps = 1e-9
    if similarity_type in ("l2sq", "l2"):
        d2 = torch.clamp((e1 ** 2).sum(-1, keepdim=True) + (e2 ** 2).sum(-1)[None, :] - 2.0 * e1 @ e2.T, min=0.0)  # utils.py:85-90
        sim = -d2 if similarity_type == "l2sq" else -torch.sqrt(d2 + eps)
    elif similarity_type == "l1":
        sim = -(e1[:, None, :] - e2[None, :, :]).abs().sum(-1)
    elif similarity_type == "linf":
        sim = -(e1[:, None, :] - e2[None, :, :]).abs().amax(-1)
    elif similarity_type == "cosine":
        sim = (e1 / e1.norm(dim=-1, keepdim=True)) @ (e2 / e2.norm(dim=-1, keepdim=True)).T
    else:
        raise ValueError("Similarity type not implemented: ", similarity_type)
    return sim / temperature


def infonce_loss_and_grads(e1, e2, similarity_type: str, temperature: float):
    """(loss, dloss/de1, dloss/de2) of train.py:209-214: mean CE(arange(B), S) + mean CE(arange(B), S^T), float64."""
    a = torch.tensor(np.asarray(e1), dtype=torch.float64, requires_grad=True)
    b = torch.tensor(np.asarray(e2), dtype=torch.float64, requires_grad=True)
    S = scaled_similarity_torch(a, b, similarity_type, temperature)
    d = torch.diagonal(S)
    loss = (torch.logsumexp(S, 1) - d).mean() + (torch.logsumexp(S, 0) - d).mean()
    ga, gb = torch.autograd.grad(loss, [a, b])
    return float(loss.detach()), ga.numpy(), gb.numpy()
