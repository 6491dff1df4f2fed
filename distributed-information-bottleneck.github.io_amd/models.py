"""Drop-in mirror of the reference's `models.py` surface on the MI355X-native path.

    DistributedIBNet / compile / fit / InfoBottleneckAnnealingCallback /
    SaveCompressionMatricesCallback / PositionalEncoding

Same names, argument meaning and History contract as the TensorFlow/Keras reference
(reference models.py:12-186, train.py:138-178); the device math is hand-written HIP for gfx950
reached through the C ABI (include/dib_hip.h) - see engine.py.  Nothing here falls back to CPU.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import losses as _losses
from . import optimizers as _optimizers


class PositionalEncoding:
    """reference models.py:12-23: concat([x] + [sin(f*x) for f in frequencies], -1).
    Parameter-free; on the hot path it is fused into the encoder-bank kernels - this host class
    only carries the frequency list (and evaluates on numpy for small host-side uses)."""

    def __init__(self, frequencies):
        self.frequencies = np.asarray(frequencies)

    def __call__(self, inputs):
        x = np.asarray(inputs)
        return np.concatenate([x] + [np.sin(f * x) for f in self.frequencies], -1)

    call = __call__


class _BetaVariable:
    """Stands in for `tf.Variable(1., trainable=False)` (reference models.py:86): device-resident
    scalar read by the kernels, so annealing never re-launches or re-captures anything."""

    def __init__(self, model, value=1.0):
        self._model = model
        self._host = np.float32(value)

    def assign(self, value):
        self._host = np.float32(value)
        if self._model._engine is not None:
            self._model._engine.set_beta(float(self._host))
        return self

    def value(self):
        return self._host

    def numpy(self):
        return self._host

    def __float__(self):
        return float(self._host)

    def __mul__(self, other):
        return self._host * other

    __rmul__ = __mul__

    def __repr__(self):
        return f"<beta={float(self._host):.6g}>"


class _FeatureEncoder:
    """`model.feature_encoders[f]`: callable [N, d_f] -> [N, 2E] (mu | logvar), deterministic
    (reference models.py:73-78,183; visualization.py:31)."""

    def __init__(self, model, index):
        self._model, self.index = model, index

    def __call__(self, x_f):
        eng = self._model._ensure_engine()
        return eng.encode_feature(self.index, x_f).detach().cpu().numpy()


class History:
    """Keras History: `.history[k]` = list of per-epoch floats (reference train.py:169-172)."""

    def __init__(self):
        self.history: Dict[str, List[float]] = {}
        self.epoch: List[int] = []
        self.model = None


class Callback:
    """Keras callback protocol subset used by the reference (on_epoch_begin/on_epoch_end, .model)."""

    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class DistributedIBNet:
    """Distributed IB model: one Gaussian bottleneck per input feature (reference models.py:26-123).

    Constructor arguments are exactly those of the reference (models.py:56-66).  Extra keyword-only
    knobs (`noise_seed`, `init_seed`, `shuffle_seed`, `device`) make runs reproducible: the
    reference samples eps from an unseeded stateful generator, here eps is a counter-based function
    of (noise_seed, step, dataset row, feature, dim).
    """

    def __init__(self,
                 feature_dimensionalities,
                 feature_encoder_architecture,
                 integration_network_architecture,
                 output_dimensionality,
                 use_positional_encoding=True,
                 number_positional_encoding_frequencies=5,
                 activation_fn='relu',
                 feature_embedding_dimension=32,
                 output_activation_fn=None,
                 *, noise_seed: int = 0, init_seed: int = 0, shuffle_seed: int = 0, device: Optional[str] = None):
        self.feature_dimensionalities = [int(d) for d in feature_dimensionalities]
        self.number_features = len(self.feature_dimensionalities)
        self.feature_encoder_architecture = [int(u) for u in feature_encoder_architecture]
        self.integration_network_architecture = [int(u) for u in integration_network_architecture]
        self.output_dimensionality = int(output_dimensionality)
        self.use_positional_encoding = bool(use_positional_encoding)
        self.number_positional_encoding_frequencies = int(number_positional_encoding_frequencies)
        self.activation_fn = activation_fn if activation_fn not in ("None", "") else None
        self.feature_embedding_dimension = int(feature_embedding_dimension)
        self.output_activation_fn = output_activation_fn
        # reference models.py:70
        self.positional_encoding_frequencies = 2 ** np.arange(1, self.number_positional_encoding_frequencies)
        self.noise_seed, self.init_seed, self.shuffle_seed = int(noise_seed), int(init_seed), int(shuffle_seed)
        self._device = device
        self._engine = None
        # data-parallel gradient all-reduce buckets (fit under torch.distributed): 3 = integration / encoder front layers /
        # last encoder layer, each issued as soon as it is final (default); 2 = integration / encoder bank; 1 = one all-reduce
        self.dp_small_batch_rows = 1024   # per-rank batches up to this many rows use ONE gradient bucket (see fit)
        self.validation_merge_rows = 1024  # fit: full validation batches are evaluated together up to this many rows (0: one by one)
        self.dp_buckets = int(os.environ.get("DIB_DP_BUCKETS", "3"))
        if self.dp_buckets not in (1, 2, 3):
            raise ValueError(f"DIB_DP_BUCKETS={self.dp_buckets}: 1, 2 or 3")
        self.beta = _BetaVariable(self, 1.0)
        self.feature_encoders = [_FeatureEncoder(self, f) for f in range(self.number_features)]
        self.optimizer = None
        self.loss = None
        self.metrics_names: List[str] = []
        self.losses: list = []
        self.history = None
        self.stop_training = False
        self._step = 0
        self._pending_weights = None

    # ---- engine -----------------------------------------------------------------------------
    def _spec_kwargs(self):
        return dict(feature_dimensionalities=self.feature_dimensionalities,
                    feature_encoder_architecture=self.feature_encoder_architecture,
                    integration_network_architecture=self.integration_network_architecture,
                    output_dimensionality=self.output_dimensionality,
                    use_positional_encoding=self.use_positional_encoding,
                    number_positional_encoding_frequencies=self.number_positional_encoding_frequencies,
                    activation_fn=self.activation_fn,
                    feature_embedding_dimension=self.feature_embedding_dimension,
                    output_activation_fn=self.output_activation_fn)

    def _make_engine(self):
        """The device engine of this model: HipEngine, which raises without a GPU or libdib_hip.so - there is no CPU path."""
        from .engine import HipEngine
        return HipEngine(**self._spec_kwargs(), device=self._device, init_seed=self.init_seed)

    def _ensure_engine(self):
        if self._engine is None:
            self._engine = self._make_engine()
            self._engine.set_beta(float(self.beta.value()))
            if self._pending_weights is not None:
                self._engine.set_flat_params(self._pending_weights)
                self._pending_weights = None
        return self._engine

    def build(self, input_shape):
        """reference models.py:88-94."""
        assert input_shape[-1] == np.sum(self.feature_dimensionalities)
        self._ensure_engine()

    # ---- weights ----------------------------------------------------------------------------
    def get_flat_weights(self) -> np.ndarray:
        return self._ensure_engine().get_flat_params()

    def set_flat_weights(self, flat) -> None:
        self._ensure_engine().set_flat_params(np.asarray(flat, dtype=np.float32))

    def param_blocks(self):
        """[{net, layer, feature, what, offset, rows, cols}] describing the flat buffer."""
        return self._ensure_engine().blocks

    @property
    def trainable_variables(self):
        """Views into the flat parameter buffer, Keras order: per feature encoder (kernel, bias)*, then
        the integration network (reference train.py:196)."""
        eng = self._ensure_engine()
        by = {(b["net"], b["layer"], b["feature"], b["what"]): b for b in eng.blocks}
        out = []
        for f in range(self.number_features):
            for l in range(len(self.feature_encoder_architecture) + 1):
                for what in (0, 1):
                    b = by[(0, l, f, what)]
                    v = eng.params[b["offset"]: b["offset"] + b["rows"] * b["cols"]]
                    out.append(v.view(b["rows"], b["cols"]) if what == 0 else v)
        for l in range(len(self.integration_network_architecture) + 1):
            for what in (0, 1):
                b = by[(1, l, 0, what)]
                v = eng.params[b["offset"]: b["offset"] + b["rows"] * b["cols"]]
                out.append(v.view(b["rows"], b["cols"]) if what == 0 else v)
        return out

    def count_params(self) -> int:
        return int(self._ensure_engine().n_params)

    # ---- call ---------------------------------------------------------------------------------
    def __call__(self, inputs, training=None):
        """reference models.py:96-123: returns the prediction [B, out]; noise is always on (the
        reference's call has no `training` switch).  Sets `self.losses = [beta * sum_f KL_f]`
        (models.py:118) and `self.last_kl` [F] (nats)."""
        eng = self._ensure_engine()
        x = eng.to_device(np.asarray(inputs, dtype=np.float32) if not isinstance(inputs, torch.Tensor) else inputs)
        if x.dim() == 1:
            x = x.view(1, -1)
        B = x.shape[0]
        eng.forward(x, None, 0, B, self.noise_seed, self._step, inference=True)  # not differentiable: forward_autograd is
        self._step += 1
        kl = eng.step_out(B)[: self.number_features].clone() / B
        self.last_kl = kl
        self.losses = [kl.sum() * float(self.beta.value())]
        return eng.pred(B).clone()

    call = __call__

    # ---- autograd bridge for custom PyTorch training loops ------------------------------------------
    def forward_autograd(self, inputs, row_ids=None):
        """Differentiable forward for hand-written loops (the reference's custom-loop contract, train.py:196-220:
        `model(x)`, `model.losses`, `model.trainable_variables` under a gradient tape).

        Returns `(prediction [B, out], kl_loss)` where `kl_loss = beta * sum_f KL_f` (models.py:118) - both are
        autograd-connected to `model.flat_parameters` (a leaf tensor aliasing the engine's flat parameter buffer):

            pred, kl_loss = model.forward_autograd(x)
            loss = my_loss(pred, y) + kl_loss
            loss.backward()                       # runs the HIP backward kernels; fills model.flat_parameters.grad
            torch_optimizer.step()                # any torch.optim optimizer over [model.flat_parameters]

        `row_ids` (int tensor [B]) keys the noise per sample; default arange(B).  The batch-mean convention of the
        KL term matches the reference: the caller's loss should be a mean over the batch."""
        eng = self._ensure_engine()
        x = eng.to_device(np.asarray(inputs, dtype=np.float32) if not isinstance(inputs, torch.Tensor) else inputs)
        if x.dim() == 1:
            x = x.view(1, -1)
        idx = None if row_ids is None else eng.to_device(row_ids, dtype=torch.int32)
        step = self._step
        self._step += 1
        pred, kl_loss = _DIBFunction.apply(self.flat_parameters, self, x, idx, step)
        self.losses = [kl_loss]
        return pred, kl_loss

    @property
    def flat_parameters(self) -> torch.Tensor:
        """The engine's flat fp32 parameter buffer as an autograd leaf (shares memory: optimizer updates are seen by
        the kernels directly)."""
        eng = self._ensure_engine()
        if getattr(self, "_flat_leaf", None) is None or self._flat_leaf.data_ptr() != eng.params.data_ptr():
            self._flat_leaf = eng.params.detach().requires_grad_(True)
        return self._flat_leaf

    def predict(self, x, batch_size=None, verbose=0):
        eng = self._ensure_engine()
        xd = eng.to_device(x)
        n = xd.shape[0]
        bs = int(batch_size or 32768)
        outs = []
        for s0 in range(0, n, bs):
            b = min(bs, n - s0)
            eng.forward(xd, None, s0, b, self.noise_seed, (1 << 30) + s0 // bs, inference=True)
            outs.append(eng.pred(b).clone())
        return torch.cat(outs, 0).cpu().numpy()

    # ---- compile / fit ---------------------------------------------------------------------
    def compile(self, optimizer='adam', loss=None, metrics=None, **_):
        """reference train.py:138-142."""
        self.optimizer = _optimizers.get(optimizer)
        self.loss = _losses.get(loss)
        self.metrics_names = list(metrics or [])
        for m in self.metrics_names:
            if m not in ("accuracy", "acc"):
                raise ValueError(f"unsupported metric {m!r} (supported: 'accuracy')")
        return self

    def _dist(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_rank(), dist.get_world_size()
        return None, 0, 1

    def _optimizer_tuple(self):
        opt = self.optimizer
        return ("adam", opt.beta_1, opt.beta_2, opt.epsilon) if opt.name == "adam" else ("sgd",)

    def _optimizer_step(self, eng):
        opt = self.optimizer
        eng.set_lr(opt.learning_rate)
        if opt.name == "adam":
            eng.adam_step(opt.beta_1, opt.beta_2, opt.epsilon)
        else:
            eng.sgd_step()

    def _epoch_logs(self, acc: np.ndarray, nsteps: int, prefix: str) -> Dict[str, float]:
        """History accounting (reference models.py:115,121; train.py:169-172):
        loss = sample-weighted mean of (task + beta*sum KL); add_metric scalars (KL{f}, beta) =
        unweighted mean over steps."""
        F = self.number_features
        rows = max(acc[F + 2], 1.0)
        logs = {prefix + "loss": acc[F] / rows}
        if self.metrics_names:
            logs[prefix + "accuracy"] = acc[F + 1] / rows
        for f in range(F):
            logs[f"{prefix}KL{f}"] = acc[f] / max(nsteps, 1)
        logs[prefix + "beta"] = float(self.beta.value())
        return logs

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
            shuffle=True, initial_epoch=0, **_):
        """reference train.py:157-166 `model.fit(x, y, epochs=, shuffle=True, batch_size=, callbacks=,
        verbose=, validation_data=)` -> History.

        Data parallel: when torch.distributed is initialised every rank calls fit() with the same
        arguments; each global batch is split row-wise across ranks, flat gradients are
        all-reduced (RCCL over xGMI with the nccl backend) and every rank applies the same update.
        """
        if self.optimizer is None or self.loss is None:
            raise RuntimeError("call compile(optimizer=..., loss=...) before fit()")
        eng = self._ensure_engine()
        dist, rank, world = self._dist()
        F = self.number_features
        x_np = np.asarray(x, dtype=np.float32)
        if x_np.ndim == 1:
            x_np = x_np[:, None]
        y_np = np.asarray(y, dtype=np.float32)
        if y_np.ndim == 1:
            y_np = y_np[:, None]
        assert x_np.shape[-1] == int(np.sum(self.feature_dimensionalities)), "input width != sum(feature dims)"
        n = x_np.shape[0]
        bs = int(batch_size or 32)
        xd, yd = eng.to_device(x_np), eng.to_device(y_np)
        if validation_data is not None:
            xv_np = np.asarray(validation_data[0], dtype=np.float32)
            yv_np = np.asarray(validation_data[1], dtype=np.float32)
            if xv_np.ndim == 1:
                xv_np = xv_np[:, None]
            if yv_np.ndim == 1:
                yv_np = yv_np[:, None]
            xvd, yvd = eng.to_device(xv_np), eng.to_device(yv_np)
        cbs = list(callbacks or [])
        for cb in cbs:
            cb.set_model(self) if hasattr(cb, "set_model") else setattr(cb, "model", self)
        hist = History()
        hist.model = self
        self.history = hist
        self.stop_training = False
        kind = self.loss.kind
        for cb in cbs:
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin()

        def reduce_metrics(acc_np):
            if dist is None:
                return acc_np
            t = torch.from_numpy(acc_np.copy()).to(eng.metrics_acc.device)
            dist.all_reduce(t)
            return t.cpu().numpy()

        # Optional hipGraph replay of the whole step (DIB_ENABLE_GRAPHS=1; needs the device-resident noise step counter;
        # single-process only).  Bit-identical to the eager launch sequence (tests).  OFF by default: measured on MI355X
        # for the reference's default Boolean-circuit run (B = 128, ~30 launches/step) the replay is ~8 % SLOWER than
        # eager (4.09 vs 3.76 ms/epoch, tools/small_batch_bench.py) - the step is bound by the latency of its dependent
        # kernels, not by launch overhead.
        n_full = n // bs
        use_graphs = (dist is None and hasattr(eng, "capture_step_graph") and bs <= 8192 and n_full * epochs >= 64
                      and os.environ.get("DIB_ENABLE_GRAPHS", "0") == "1")
        graphs = {}
        if use_graphs:
            eng.enable_step_counter(self._step)
            eng.set_lr(self.optimizer.learning_rate)
            opt_args = ((self.optimizer.beta_1, self.optimizer.beta_2, self.optimizer.epsilon)
                        if self.optimizer.name == "adam" else ())
            graphs["train"] = eng.capture_step_graph(xd, yd, bs, kind, 1.0 / bs, self.noise_seed, True,
                                                     self.optimizer.name, opt_args)
            eng.set_step_counter(self._step)
        elif getattr(eng, "step_dev", None) is not None:
            eng.set_step_counter(self._step)

        def epoch_order(epoch):
            """row order of `epoch` on the device (int32): Keras reshuffles every epoch; seeded by (shuffle_seed, epoch)"""
            order = (np.random.default_rng([self.shuffle_seed, epoch]).permutation(n) if shuffle else np.arange(n)).astype(np.int32)
            return eng.to_device(order, dtype=torch.int32)

        next_order = None
        try:
            for epoch in range(initial_epoch, epochs):
                for cb in cbs:
                    cb.on_epoch_begin(epoch)
                eng.set_beta(float(self.beta.value()))
                order_dev = next_order if next_order is not None else epoch_order(epoch)
                next_order = None
                nsteps = 0
                for s0 in range(0, n, bs):
                    gb = min(bs, n - s0)  # last partial batch is kept (Keras)
                    if use_graphs and gb == bs:
                        g, stage = graphs["train"]
                        eng.set_lr(self.optimizer.learning_rate)
                        stage.copy_(order_dev[s0: s0 + bs])
                        g.replay()  # fwd + loss + bwd + optimizer + metrics + step-counter bump
                        self._step += 1
                        nsteps += 1
                        continue
                    lo = (gb * rank) // world
                    hi = (gb * (rank + 1)) // world
                    # Every rank issues the SAME collectives every step, rows or no rows (a tail batch with fewer rows than
                    # ranks leaves some ranks empty: they contribute zeros).  Gradient buckets of the layer-major flat buffer
                    # (DESIGN 6), each all-reduced (RCCL, async) as soon as it is final:
                    #   1 integration network   - under the whole encoder-bank backward
                    #   2 encoder front layers  - under the last encoder layer's weight gradient      (dp_buckets == 3)
                    #   3 last encoder layer    - the only exposed one                                (dp_buckets == 3)
                    #   0 = 2 + 3 as one bucket after the backward                                    (dp_buckets == 2)
                    pending = []
                    nb = self.dp_buckets if (dist is not None and hasattr(eng, "part_range")) else 1
                    if nb > 1 and gb // world <= self.dp_small_batch_rows:
                        # small per-rank batches (<= 1024 rows: a handful of kernels per step, one grouped launch for every weight
                        # gradient): the backward is a handful of launches with nothing to hide an all-reduce under - one
                        # bucket after it, and the launches (hence the bits) of the single-process step.  Decided from the
                        # GLOBAL batch: identical on every rank.
                        nb = 1
                    issue = lambda g: pending.append(dist.all_reduce(g, async_op=True))
                    if dist is None and getattr(eng, "fused_optimizer_tail", False):
                        # one process: the step's LAST launch reduces the gradient partials, sums the KL / loss partials,
                        # accumulates the History metrics and applies the optimizer (csrc/dib_tail.h)
                        eng.set_lr(self.optimizer.learning_rate)
                        eng.train_step(xd, yd, order_dev[s0: s0 + gb], 0, gb, self.noise_seed, self._step, kind,
                                       inv_global_batch=1.0 / gb, optimizer=self._optimizer_tuple())
                        self._step += 1
                        nsteps += 1
                        if getattr(eng, "step_dev", None) is not None:
                            eng.set_step_counter(self._step)
                        continue
                    if hi > lo:
                        eng.train_step(xd, yd, order_dev[s0 + lo: s0 + hi], 0, hi - lo, self.noise_seed, self._step, kind,
                                       inv_global_batch=1.0 / gb, on_integration_grads_ready=issue if nb >= 2 else None,
                                       **(dict(on_encoder_front_grads_ready=issue) if nb == 3 else {}))
                    else:
                        eng.grads.zero_()
                        for part in ((1,) if nb == 2 else (1, 2) if nb == 3 else ()):
                            off, cnt = eng.part_range(part)
                            issue(eng.grads[off: off + cnt])
                    if nb >= 2:
                        off, cnt = eng.part_range(3 if nb == 3 else 0)
                        issue(eng.grads[off: off + cnt])
                        if hasattr(eng, "optimizer_step_part"):
                            # each bucket is stepped as soon as ITS all-reduce has landed: buckets 1 (and 2) are updated on
                            # the compute stream while the last bucket is still on the wire; every launch reads the same
                            # Adam step count, the last one advances it
                            eng.set_lr(self.optimizer.learning_rate)
                            parts = (1, 2, 3) if nb == 3 else (1, 0)
                            for k, (w, part) in enumerate(zip(pending, parts)):
                                w.wait()
                                eng.optimizer_step_part(max(hi - lo, 1), part, self._optimizer_tuple(), bump=k == len(parts) - 1)
                        else:
                            for w in pending:
                                w.wait()
                            self._optimizer_step(eng)
                    else:
                        if dist is not None:
                            dist.all_reduce(eng.grads)
                        self._optimizer_step(eng)
                    self._step += 1
                    nsteps += 1
                    if getattr(eng, "step_dev", None) is not None:
                        eng.set_step_counter(self._step)  # eager step under the device counter: keep it in sync
                # host work under the device's queue: the next epoch's permutation (12 ms of numpy for 2^20 rows - with it at
                # the top of the epoch the device idled 7 % of a config-3 epoch, bench.py extra.fit_surface) and its upload are
                # done while this epoch's steps are still executing; the read-back below is the epoch's only synchronisation
                if epoch + 1 < epochs and not self.stop_training:
                    next_order = epoch_order(epoch + 1)
                # ONE synchronisation per epoch: the validation pass is enqueued right behind the training steps - its History
                # sums go to an accumulator of their own - and both are read back together (round 5 read the training sums
                # first: the device idled through that round trip and the host work behind it, every epoch)
                vsteps = 0
                early = None
                if getattr(self, "syncs_per_epoch", 1) == 2:   # A/B switch: round 5's order (training sums read before the validation pass)
                    early = eng.read_metrics_pair()[0]
                if validation_data is not None:
                    nv = xvd.shape[0]
                    if getattr(eng, "step_dev", None) is not None:
                        eng.set_step_counter((1 << 31) + epoch)  # validation noise stream, same key as the eager path
                    # Validation batches do not depend on each other - no state changes between them, one noise key per epoch
                    # (rows keyed by their dataset index) - and every History quantity is linear in per-row sums: k full
                    # batches are evaluated as ONE launch set of k * bs rows with the per-batch 1 / bs scaling and counted as k
                    # steps.  Same per-row numbers, sums in another order (fp32 rounding).  At the reference's default (8
                    # validation batches of 128 rows per epoch, train.py:30-34) that is 3 launches per epoch instead of 24.
                    kmerge = max(1, self.validation_merge_rows // bs) if world == 1 else 1
                    s0 = 0
                    while s0 < nv:
                        gb = min(bs, nv - s0)
                        nb = min(kmerge, (nv - s0) // bs) if gb == bs else 1
                        rows = gb * nb
                        lo = (rows * rank) // world
                        hi = (rows * (rank + 1)) // world
                        if hi > lo:
                            eng.eval_step(xvd, yvd, None, s0 + lo, hi - lo, self.noise_seed, (1 << 31) + epoch, kind,
                                          inv_global_batch=1.0 / gb, metrics_acc=eng.metrics_acc_val)
                        vsteps += nb
                        s0 += rows
                    if getattr(eng, "step_dev", None) is not None:
                        eng.set_step_counter(self._step)
                train_sums, val_sums = eng.read_metrics_pair()
                if early is not None:
                    train_sums = early
                logs = self._epoch_logs(reduce_metrics(train_sums), nsteps, "")
                if validation_data is not None:
                    logs.update(self._epoch_logs(reduce_metrics(val_sums), vsteps, "val_"))
                for k, v in logs.items():
                    hist.history.setdefault(k, []).append(float(v))
                hist.epoch.append(epoch)
                if verbose and rank == 0:
                    kls = sum(logs[f"KL{f}"] for f in range(F))
                    print(f"Epoch {epoch + 1}/{epochs} - loss: {logs['loss']:.4f} - sumKL: {kls:.4f} nats - "
                          f"beta: {logs['beta']:.3e}" + (f" - val_loss: {logs['val_loss']:.4f}" if 'val_loss' in logs else ""))
                for cb in cbs:
                    cb.on_epoch_end(epoch, logs)
                if self.stop_training:
                    break
            for cb in cbs:
                if hasattr(cb, "on_train_end"):
                    cb.on_train_end()
        finally:
            if graphs:   # the captured step is dropped with `graphs`: its pinned workspace may be evicted again
                torch.cuda.synchronize(eng.device)   # no replay may be in flight when the graph object dies
                graphs.clear()
                eng.release_step_graph(bs)
        return hist

    def evaluate(self, x, y, batch_size=None, verbose=0, return_dict=True):
        eng = self._ensure_engine()
        if self.loss is None:
            raise RuntimeError("call compile() before evaluate()")
        xd = eng.to_device(np.asarray(x, dtype=np.float32))
        y_np = np.asarray(y, dtype=np.float32)
        yd = eng.to_device(y_np[:, None] if y_np.ndim == 1 else y_np)
        n, bs = xd.shape[0], int(batch_size or 32)
        eng.read_metrics()
        steps = 0
        # full batches are evaluated together like fit's validation pass (same per-row numbers; see validation_merge_rows)
        kmerge, s0 = max(1, self.validation_merge_rows // bs), 0
        while s0 < n:
            b = min(bs, n - s0)
            nb = min(kmerge, (n - s0) // bs) if b == bs else 1
            eng.eval_step(xd, yd, None, s0, b * nb, self.noise_seed, (1 << 31) - 1, self.loss.kind, inv_global_batch=1.0 / b)
            steps += nb
            s0 += b * nb
        return self._epoch_logs(eng.read_metrics(), steps, "")


class _DIBFunction(torch.autograd.Function):
    """torch.autograd bridge: forward = dib_encoder_bank_fwd + dib_integration_fwd, backward = the HIP backward
    kernels with the caller's dL/dpred injected (engine.backward_from_pred_grad)."""

    @staticmethod
    def forward(ctx, flat_params, model, x, idx, step):
        eng = model._ensure_engine()
        B = x.shape[0]
        eng.forward(x, idx, 0, B, model.noise_seed, step)
        ctx.model, ctx.idx, ctx.step, ctx.B = model, idx, step, B
        ctx.beta = float(model.beta.value())
        # the activations the backward kernels re-read live in the engine's workspace for B rows: remember which forward
        # wrote them so that backward() can refuse to run on a workspace another forward has overwritten since
        ctx.ws_gen = eng._ws_gen.get(B) if hasattr(eng, "_ws_gen") else None
        kl = eng.step_out(B)[: model.number_features].sum() / B
        return eng.pred(B).clone(), (kl * ctx.beta).reshape(())

    @staticmethod
    def backward(ctx, g_pred, g_kl):
        model = ctx.model
        eng = model._ensure_engine()
        if ctx.ws_gen is not None and eng._ws_gen.get(ctx.B) != ctx.ws_gen:
            raise RuntimeError(
                f"forward_autograd: another forward pass with batch size {ctx.B} ran between forward_autograd() and "
                "backward(); it overwrote the stashed activations of this graph node.  Call backward() before the next "
                "model(x) / predict / evaluate / forward_autograd of the same batch size.")
        # d/dparams of (beta * sum KL) scaled by the incoming gradient of the KL-loss output - as a device scalar, no host sync
        if g_kl is not None:
            eng.beta_dev.copy_((g_kl.detach().to(torch.float32) * ctx.beta).reshape(1))
        else:
            eng.set_beta(0.0)
        eng.backward_from_pred_grad(g_pred.contiguous(), ctx.idx, 0, ctx.B, model.noise_seed, ctx.step,
                                    inv_global_batch=1.0 / ctx.B)
        eng.set_beta(float(model.beta.value()))
        return eng.grads.clone(), None, None, None, None


class InfoBottleneckAnnealingCallback(Callback):
    """Logarithmically ramp beta during training (reference models.py:125-149).

    on_epoch_begin: beta <- exp(log b0 + max(epoch - n_pre, 0)/n_anneal * (log b1 - log b0)),
    evaluated in float32 like the TF ops of the reference (models.py:147-149).
    """

    def __init__(self, beta_start, beta_end, number_pretraining_epochs, number_annealing_epochs):
        super().__init__()
        self.beta_start = beta_start
        self.beta_end = beta_end
        self.number_pretraining_epochs = number_pretraining_epochs
        self.number_annealing_epochs = number_annealing_epochs

    def beta_at(self, epoch) -> np.float32:
        f32 = np.float32
        frac = f32(max(epoch - self.number_pretraining_epochs, 0)) / f32(self.number_annealing_epochs)
        return f32(np.exp(np.log(f32(self.beta_start)) + frac * (np.log(f32(self.beta_end)) - np.log(f32(self.beta_start)))))

    def on_epoch_begin(self, epoch, logs=None):
        self.model.beta.assign(self.beta_at(epoch))


class SaveCompressionMatricesCallback(Callback):
    """Save per-feature compression-scheme matrices during training (reference models.py:152-186;
    defects A1/A2 of SURVEY App. A fixed by intent: the working copy is train.py:253-261).

    Every `save_frequency` epochs: for each feature, encode a sample of the validation values
    deterministically, Bhattacharyya matrix -> exp(-D) -> PNG named
    feature_{f}_log10beta_{log10(beta):.3f}.png (models.py:181).  The matrices are also kept in
    `self.matrices[(epoch, f)]`.
    """

    def __init__(self, save_frequency, x_processed, x_raw, outdir, save_png=True):
        super().__init__()
        self.save_frequency = save_frequency
        self.x_processed = np.asarray(x_processed)
        self.x_raw = np.asarray(x_raw)
        self.outdir = outdir
        self.save_png = save_png
        self.matrices = {}

    def on_epoch_end(self, epoch, logs=None):
        if (epoch % self.save_frequency) != 0:
            return
        from . import visualization
        dist, rank, _ = self.model._dist()
        if rank != 0:
            return
        beta_value = float(self.model.beta.value())
        dims = self.model.feature_dimensionalities
        idx = np.cumsum(dims)[:-1]
        features_split = np.split(self.x_processed, idx, axis=-1)
        features_split_raw = np.split(self.x_raw, idx, axis=-1)
        os.makedirs(self.outdir, exist_ok=True)
        for feature_ind in range(self.model.number_features):
            out_fname = os.path.join(self.outdir, f'feature_{feature_ind}_log10beta_{np.log10(beta_value):.3f}.png')
            mat = visualization.save_compression_matrices(
                self.model.feature_encoders[feature_ind], features_split[feature_ind],
                out_fname if self.save_png else None, inp_features_raw=features_split_raw[feature_ind],
                model=self.model)
            self.matrices[(epoch, feature_ind)] = mat


class InfoPerFeatureCallback(Callback):
    """Information (nats) in each compression channel during training (reference models.py:188-223; the kwarg-name
    defect A3 of SURVEY App. A fixed by intent).

    Every `save_frequency` epochs, for each feature: `utils.estimate_mi_sandwich_bounds` on that feature's columns of
    the validation inputs; appends `[infonce_lower, loo_upper]` to `self.bounds` (same flat order as the reference:
    feature-major within an evaluation).  `validation_x` is the validation input matrix [N, sum d_f] (the reference
    passes a tf.data.Dataset of (x, y) and strips y, models.py:210)."""

    def __init__(self, save_frequency, validation_x, info_bound_batch_size=1024, info_bound_number_batches=8):
        super().__init__()
        self.save_frequency = save_frequency
        self.validation_x = np.asarray(validation_x, dtype=np.float32)
        self.bounds = []
        self.epochs = []
        self.info_bound_batch_size = info_bound_batch_size
        self.info_bound_number_batches = info_bound_number_batches

    def on_epoch_end(self, epoch, logs=None):
        if (epoch % self.save_frequency) != 0:
            return
        from . import utils
        idx = np.cumsum(self.model.feature_dimensionalities)[:-1]
        split = np.split(self.validation_x, idx, axis=-1)
        for feature_ind in range(self.model.number_features):
            lower, upper = utils.estimate_mi_sandwich_bounds(
                self.model.feature_encoders[feature_ind], split[feature_ind],
                evaluation_batch_size=self.info_bound_batch_size,
                number_evaluation_batches=self.info_bound_number_batches, seed=epoch)
            self.bounds.append([lower, upper])
        self.epochs.append(epoch)
