"""TEST-ONLY stand-ins for the two device objects dib_amd.infonce.fit_infonce drives - the X model's engine and the Y
encoder - on float64 torch-CPU autograd, so that the loop's HOST composition (row sharding, the all-gather of both
embedding sets, which gradient rows a rank keeps, the 1 / global-batch scaling of the KL term, the two all-reduces, one
Adam per network, beta hand-over) runs under gloo without a GPU and is compared with the single-process float64 oracle
of the whole loop (oracle/infonce_loop_oracle.py).  Lives in tests/ - the product package never imports it."""
import math

import numpy as np
import torch

import dib_oracle as orc
from dib_torch_cpu import TorchCpuDIB, scaled_similarity_torch
from infonce_loop_oracle import YEncoder


class _FlatAdam:
    """Keras-form Adam on the flat float64 gradient buffer `self.grads` of the variables `self.vars` (what
    dib_adam_step does on the device: the product all-reduces `grads` in place, then calls adam_step())."""

    def _init_flat(self, variables):
        self.vars = list(variables)
        n = sum(v.numel() for v in self.vars)
        self.grads = torch.zeros(n, dtype=torch.float64)
        self._m, self._v, self._t, self.lr = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64), 0, 1e-3

    def _store_grads(self, grads):
        self.grads.copy_(torch.cat([g.reshape(-1) for g in grads]))

    def set_lr(self, lr):
        self.lr = float(lr)

    def adam_step(self, lr=None, beta1=0.9, beta2=0.999, eps=1e-7, fused_reduce=False):
        if lr is not None:
            self.lr = float(lr)
        self._t += 1
        lr_t = self.lr * math.sqrt(1.0 - beta2 ** self._t) / (1.0 - beta1 ** self._t)
        g = self.grads
        self._m += (1 - beta1) * (g - self._m)
        self._v += (1 - beta2) * (g * g - self._v)
        upd = lr_t * self._m / (torch.sqrt(self._v) + eps)
        off = 0
        with torch.no_grad():
            for p in self.vars:
                p.sub_(upd[off: off + p.numel()].view_as(p))
                off += p.numel()

    def flat_params(self):
        return torch.cat([p.detach().reshape(-1) for p in self.vars]).numpy().copy()


class CheckerXEngine(_FlatAdam):
    """The subset of HipEngine's interface that fit_infonce uses."""

    def __init__(self, spec: orc.DIBSpec, params: orc.DIBParams):
        self.spec = spec
        self.model = TorchCpuDIB(spec, params, dtype=torch.float64)
        self._init_flat(self.model.tensors())
        self.F, self.E = spec.number_features, spec.feature_embedding_dimension
        self.device, self.beta = torch.device("cpu"), 1.0
        self._gp = None
        self.metrics_acc = torch.zeros(self.F + 3, dtype=torch.float64)   # [0, F): sum over steps of KL_f batch means

    def to_device(self, a, dtype=torch.float32):
        return a.to(dtype) if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a)).to(dtype)

    def set_beta(self, v):
        self.beta = float(v)

    def forward(self, x, row_idx, row0, batch, seed, step, deterministic=False, inference=False, defer_sums=False):
        rows = row_idx.numpy().astype(np.int64)
        assert len(rows) == batch
        xb = torch.tensor(x.numpy().astype(np.float64)[rows])
        eps = torch.tensor(orc.philox_normal_all(seed, step, rows.astype(np.uint32), self.F, self.E), dtype=torch.float64)
        self._ex, self._kl = self.model.forward(xb, eps)          # graph kept for backward_from_pred_grad

    def pred(self, batch):
        return self._ex.detach()

    def g_pred(self, batch):
        self._gp = torch.zeros_like(self._ex.detach())
        return self._gp

    def step_out(self, batch):
        return torch.cat([self._kl.detach() * batch, torch.zeros(3, dtype=torch.float64)])

    def infonce(self, emb_x, emb_y, similarity, temperature, want_grads=True, out_gx=None, out_gy=None, loss_out=None):
        a = emb_x.detach().clone().requires_grad_(True)
        b = emb_y.detach().clone().requires_grad_(True)
        S = scaled_similarity_torch(a, b, similarity, temperature)
        d = torch.diagonal(S)
        loss = (torch.logsumexp(S, 1) - d).mean() + (torch.logsumexp(S, 0) - d).mean()
        if loss_out is not None:   # the product hands a slot of its per-epoch buffer
            loss_out.copy_(loss.detach().reshape(1))
        if not want_grads:
            return (loss_out if loss_out is not None else loss.detach()), None, None
        ga, gb = torch.autograd.grad(loss, [a, b])
        if out_gx is not None:
            out_gx.copy_(ga)
            ga = out_gx
        if out_gy is not None:
            out_gy.copy_(gb)
            gb = out_gy
        return (loss_out if loss_out is not None else loss.detach()), ga, gb

    def step_tail(self, batch, part, flags, inv_global_batch=0.0, optimizer=None, grad_scale=1.0, metrics_acc=None):
        """the subset fit_infonce uses: TAIL_KL | TAIL_METRICS of a validation step (include/dib_hip.h: 2 | 32)"""
        assert flags == (2 | 32) and optimizer is None
        acc = self.metrics_acc if metrics_acc is None else metrics_acc
        acc[: self.F] += self._kl.detach() * batch * inv_global_batch

    def backward_from_pred_grad(self, g_pred, row_idx, row0, batch, seed, step, inv_global_batch=None, finish_flags=0,
                                optimizer=None, metrics_acc=None):
        inv = 1.0 / batch if inv_global_batch is None else inv_global_batch
        # d/dtheta [ sum_rows <emb_x, dL/d emb_x> + beta * sum_rows sum_f KL / global batch ]; _kl is the LOCAL row mean
        obj = (self._ex * g_pred).sum() + self.beta * self._kl.sum() * batch * inv
        self._store_grads(torch.autograd.grad(obj, self.vars))
        if finish_flags & 32:
            acc = self.metrics_acc if metrics_acc is None else metrics_acc
            acc[: self.F] += self._kl.detach() * batch * inv
        if optimizer is not None:
            assert optimizer[0] == "adam"
            self.adam_step(None, *optimizer[1:4])


class CheckerYEncoder(_FlatAdam):
    """The subset of DenseStack's interface that fit_infonce uses, over the oracle's YEncoder."""

    def __init__(self, kernels, biases, activation, use_positional_encoding, number_positional_encoding_frequencies):
        self.enc = YEncoder(kernels, biases, activation, use_positional_encoding, number_positional_encoding_frequencies)
        self._init_flat(self.enc.tensors())

    def forward(self, y, rows=None):
        yb = y.numpy().astype(np.float64)
        if rows is not None:
            yb = yb[rows.numpy().astype(np.int64)]
        self._out = self.enc.forward(torch.tensor(yb))
        self._gbuf = torch.zeros_like(self._out.detach())
        return self._out.detach()

    def output_grad_buffer(self):
        return self._gbuf

    def backward(self, g_out, reduce=True):
        self._store_grads(torch.autograd.grad((self._out * g_out).sum(), self.vars))
