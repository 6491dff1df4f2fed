"""TEST-ONLY engine: implements the engine interface of dib_amd.models on top of the CPU oracle, so
that the host logic (fit loop, callbacks, History accounting, data-parallel sharding + all-reduce)
can be exercised without a GPU.  Lives in tests/ - the product package never imports it."""
import numpy as np
import torch

import dib_oracle as orc
from _helpers import flat_to_params, params_to_flat


class OracleEngine:
    def __init__(self, init_seed=0, **spec_kw):
        self.spec = orc.DIBSpec(**spec_kw)
        self.F, self.E = self.spec.number_features, self.spec.feature_embedding_dimension
        self.dims = list(self.spec.feature_dimensionalities)
        self.out_dim = self.spec.output_dimensionality
        self.device = torch.device("cpu")
        # flat layout: same block order idea as the HIP layout (exact offsets are engine-private)
        self.blocks, off = [], 0
        nle = len(self.spec.feature_encoder_architecture) + 1
        for l in range(nle):
            for f in range(self.F):
                i, o = self.spec.encoder_layer_dims(f)[l]
                self.blocks.append(dict(net=0, layer=l, feature=f, what=0, offset=off, rows=i, cols=o)); off += i * o
            for f in range(self.F):
                o = self.spec.encoder_layer_dims(f)[l][1]
                self.blocks.append(dict(net=0, layer=l, feature=f, what=1, offset=off, rows=1, cols=o)); off += o
        for l, (i, o) in enumerate(self.spec.integration_layer_dims()):
            self.blocks.append(dict(net=1, layer=l, feature=0, what=0, offset=off, rows=i, cols=o)); off += i * o
            self.blocks.append(dict(net=1, layer=l, feature=0, what=1, offset=off, rows=1, cols=o)); off += o
        self.n_params = off
        self.p = orc.glorot_uniform_init(self.spec, init_seed)
        self.params = torch.from_numpy(params_to_flat(self.blocks, self.p, off, np.float64))
        self.grads = torch.zeros(off, dtype=torch.float64)
        self.metrics_acc = torch.zeros(self.F + 3, dtype=torch.float64)
        self.metrics_acc_val = torch.zeros(self.F + 3, dtype=torch.float64)
        self.state = orc.adam_init(self.p)
        self.beta, self.lr = 1.0, 1e-3
        self._pred = None

    # plumbing
    def to_device(self, a, dtype=torch.float32):
        return a.to(dtype) if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a)).to(dtype)

    def set_beta(self, v): self.beta = float(v)
    def get_beta(self): return self.beta
    def set_lr(self, v): self.lr = float(v)
    def get_flat_params(self): return params_to_flat(self.blocks, self.p, self.n_params, np.float64)

    def set_flat_params(self, flat):
        self.p = flat_to_params(self.blocks, np.asarray(flat, dtype=np.float64), self.spec)

    def _rows(self, row_idx, row0, batch):
        return row_idx.numpy().astype(np.int64) if row_idx is not None else np.arange(row0, row0 + batch)

    def _fwd(self, x, row_idx, row0, batch, seed, step):
        rows = self._rows(row_idx, row0, batch)
        xb = x.numpy().astype(np.float64)[rows]
        eps = orc.philox_normal_all(seed, step, rows.astype(np.uint32), self.F, self.E)
        return rows, xb, orc.forward(self.spec, self.p, xb, eps)

    def forward(self, x, row_idx, row0, batch, seed, step, deterministic=False, inference=False):
        _, _, c = self._fwd(x, row_idx, row0, batch, seed, step)
        self._pred, self._kl = c.pred, c.kl

    def pred(self, batch): return torch.from_numpy(self._pred)
    def step_out(self, batch): return torch.from_numpy(np.concatenate([self._kl * batch, [0, 0, batch]]))

    def _account(self, c, task, yb, kind, batch, inv):
        acc = self.metrics_acc.numpy()
        acc[: self.F] += c.kl * batch * inv
        acc[self.F] += task * batch + self.beta * c.kl.sum() * batch
        if kind != "mse":
            acc[self.F + 1] += orc.accuracy(kind, yb, c.pred) * batch
        acc[self.F + 2] += batch

    def part_range(self, part):
        """same bucket numbering as include/dib_hip.h: 0 encoder bank, 1 integration, 2 encoder front layers, 3 last layer"""
        split = min(b["offset"] for b in self.blocks if b["net"] == 1)
        last = max(b["layer"] for b in self.blocks if b["net"] == 0)
        tail = min(b["offset"] for b in self.blocks if b["net"] == 0 and b["layer"] == last)
        return {0: (0, split), 1: (split, self.n_params - split), 2: (0, tail), 3: (tail, split - tail)}[part]

    fused_optimizer_tail = True   # same engine contract as HipEngine: train_step(optimizer=...) applies the update itself

    def train_step(self, x, y, row_idx, row0, batch, seed, step, loss_kind, inv_global_batch=None, accumulate=True,
                   on_integration_grads_ready=None, on_encoder_front_grads_ready=None, optimizer=None):
        inv = 1.0 / batch if inv_global_batch is None else inv_global_batch
        rows, xb, c = self._fwd(x, row_idx, row0, batch, seed, step)
        yb = y.numpy()[rows]
        task, g, _ = orc.backward(self.spec, self.p, xb, yb, c, self.beta, loss_kind,
                                  loss_scale_rows=int(round(1.0 / inv)))
        self.grads.copy_(torch.from_numpy(params_to_flat(self.blocks, g, self.n_params, np.float64)))
        self._gstruct = g
        if on_integration_grads_ready is not None:  # bucket protocol of the product engine: same hooks, same order
            off, cnt = self.part_range(1)
            on_integration_grads_ready(self.grads[off: off + cnt])
        if on_encoder_front_grads_ready is not None:
            off, cnt = self.part_range(2)
            on_encoder_front_grads_ready(self.grads[off: off + cnt])
        if accumulate:
            self._account(c, task, yb, loss_kind, batch, inv)
        if optimizer is not None:
            assert on_integration_grads_ready is None and on_encoder_front_grads_ready is None
            if optimizer[0] == "adam":
                self.adam_step(*optimizer[1:4])
            else:
                self.sgd_step()

    def optimizer_step_part(self, batch, part, optimizer, bump):
        """the optimizer on ONE gradient bucket; every call of a step sees the same Adam step count, `bump` advances it"""
        off, cnt = (0, self.n_params) if part == -1 else self.part_range(part)   # -1: the whole buffer (one-bucket protocol)
        keep = np.ones(self.n_params, dtype=bool)
        keep[off: off + cnt] = False
        flat = lambda q: params_to_flat(self.blocks, q, self.n_params, np.float64)
        before = [flat(self.p), flat(self.state.m), flat(self.state.v)]
        t0 = self.state.t
        if optimizer[0] == "adam":
            self.adam_step(*optimizer[1:4])
        else:
            self.sgd_step()
        after = [flat(self.p), flat(self.state.m), flat(self.state.v)]
        for a, b in zip(after, before):
            a[keep] = b[keep]
        self.p = flat_to_params(self.blocks, after[0], self.spec)
        self.state.m = flat_to_params(self.blocks, after[1], self.spec)
        self.state.v = flat_to_params(self.blocks, after[2], self.spec)
        self.state.t = t0 + 1 if (bump and optimizer[0] == "adam") else t0

    def eval_step(self, x, y, row_idx, row0, batch, seed, step, loss_kind, inv_global_batch=None, metrics_acc=None):
        inv = 1.0 / batch if inv_global_batch is None else inv_global_batch
        rows, xb, c = self._fwd(x, row_idx, row0, batch, seed, step)
        yb = y.numpy()[rows]
        task, _ = orc.loss_and_grad(loss_kind, yb, c.pred)
        if metrics_acc is None:
            self._account(c, task, yb, loss_kind, batch, inv)
        else:   # (the product's fit: validation sums go to an accumulator of their own)
            keep, self.metrics_acc = self.metrics_acc, metrics_acc
            try:
                self._account(c, task, yb, loss_kind, batch, inv)
            finally:
                self.metrics_acc = keep

    def adam_step(self, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        g = flat_to_params(self.blocks, self.grads.numpy() * grad_scale, self.spec)
        orc.adam_keras_step(self.p, g, self.state, lr=self.lr, b1=beta1, b2=beta2, eps=eps)

    def sgd_step(self, grad_scale=1.0):
        g = flat_to_params(self.blocks, self.grads.numpy() * grad_scale, self.spec)
        for p, gg in zip(self.p.tensors(), g.tensors()):
            p -= self.lr * gg

    def read_metrics(self, reset=True):
        m = self.metrics_acc.numpy().copy()
        if reset:
            self.metrics_acc.zero_()
        return m

    def read_metrics_pair(self, reset=True):
        a, b = self.metrics_acc.numpy().copy(), self.metrics_acc_val.numpy().copy()
        if reset:
            self.metrics_acc.zero_()
            self.metrics_acc_val.zero_()
        return a, b

    def encode_feature(self, f, x_f):
        return torch.from_numpy(orc.encode_feature(self.spec, self.p, f, np.asarray(x_f, dtype=np.float64)))

    def mi_sandwich_bounds(self, enc_out, seed, step, feature):
        e = enc_out.numpy() if hasattr(enc_out, "numpy") else np.asarray(enc_out)
        E = e.shape[1] // 2
        u = orc.mi_sandwich_sample_u(e[:, :E], e[:, E:], seed, step, feature)
        return orc.mi_sandwich_bounds_batch(e[:, :E], e[:, E:], u)
