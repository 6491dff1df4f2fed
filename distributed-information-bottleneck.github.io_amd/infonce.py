"""The reference's custom InfoNCE training loop (train.py:180-289) on the MI355X-native engine.

X is encoded by the DistributedIBNet (its output is the shared-space embedding), Y by a DenseStack MLP
(train.py:184-192); the loss is the symmetric InfoNCE over the in-batch similarity matrix (train.py:203-215) plus
beta * sum KL (models.py:118).  Epoch bookkeeping follows the reference: full batches from a repeating shuffled
stream, epoch boundaries at round(steps_per_epoch * epoch) (train.py:222-236), beta updated with numpy maths at each
boundary (train.py:248), validation with noise on over number_full_validation_batches + 1 batches (train.py:230-234,
262-268).  The batch order is tf.data's `repeat().shuffle(min(n, 10_000)).batch(B)` (train.py:226-227: a 10 000-element shuffle
buffer over the repeating SEQUENTIAL stream - `_BatchStream`); the validation dataset object is iterated afresh at every epoch
boundary (train.py:262: a new iterator = the source restarts at row 0, the buffer is refilled and reshuffled).  One difference,
by intent: `infonce_space_dimensionality` (a typo'd attribute in the reference) is the
`--infonce_shared_dimensionality` flag.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .dense import DenseStack


_IDX_BLOCK = 256          # training steps whose row indices travel to the device in one upload
SHUFFLE_BUFFER = 10_000   # train.py:227,234: shuffle(min(dataset_length, 10_000))


class _BatchStream:
    """tf.data's `from_tensor_slices(...).repeat().shuffle(min(n, 10_000)).batch(batch_size)` (train.py:226-227, 233-234) as a
    stream of dataset ROW INDICES: the source is the endless sequential stream 0, 1, ..., n-1, 0, 1, ... (repeat() comes BEFORE
    shuffle(), so passes over the data blend into each other); the shuffle buffer is filled with its first min(n, 10_000)
    elements; every draw emits a uniformly chosen slot of the buffer and refills that slot with the next element of the source.
    So the element emitted as draw t comes from source positions < t + buffer: for a dataset larger than the buffer - the
    pendulum's 240 000 time-ordered rows (data.py:122-123) - a batch's in-batch negatives all lie in a 10 000-row window
    (+ batch) of the sequential order, and no source position is ever emitted twice or skipped.

    Slots are drawn from numpy's default_rng(seed).integers(0, buffer) - one draw per emitted element, so the stream does not
    depend on how many elements a call asks for (oracle/infonce_loop_oracle.BatchStream draws them one by one).  TensorFlow's
    own Philox-seeded slot choice cannot be reproduced (and is unseeded in the reference): the DISTRIBUTION is tf.data's, the
    random numbers are this project's."""

    def __init__(self, n: int, batch_size: int, seed, buffer_size: int = SHUFFLE_BUFFER):
        self.n, self.bs, self.rng = int(n), int(batch_size), np.random.default_rng(seed)
        self.nbuf = min(self.n, int(buffer_size))
        self.buf = np.arange(self.nbuf, dtype=np.int64) % self.n      # source positions 0 .. nbuf-1
        self.pos = self.nbuf                                          # next source position

    def draw(self, m: int) -> np.ndarray:
        """the next m emitted row indices (vectorised form of m sequential draw-emit-refill steps)"""
        slots = self.rng.integers(0, self.nbuf, size=m)
        incoming = (self.pos + np.arange(m, dtype=np.int64)) % self.n
        self.pos += m
        # a slot drawn again inside this call holds the element an EARLIER draw of the call refilled it with
        # (stable sort of 16-bit keys = numpy's radix sort: 4 x faster than the 64-bit merge sort at 500 000 draws)
        order = np.argsort(slots.astype(np.uint16) if self.nbuf <= 65536 else slots, kind="stable")
        ss = slots[order]
        again = np.flatnonzero(ss[1:] == ss[:-1]) + 1
        out = self.buf[slots]
        out[order[again]] = incoming[order[again - 1]]
        # one write per slot: its LAST refill of this call (numpy leaves the result of an advanced-index assignment with repeated
        # indices unspecified)
        last = np.r_[ss[1:] != ss[:-1], True] if m else np.zeros(0, dtype=bool)
        self.buf[ss[last]] = incoming[order[last]]
        return out

    def next(self) -> np.ndarray:
        return self.draw(self.bs)

    def next_batches(self, k: int) -> np.ndarray:
        """[k, batch_size]: the next k batches in one call (one host-to-device upload for k steps)"""
        return self.draw(k * self.bs).reshape(k, self.bs)


def _dist():
    """torch.distributed if a multi-rank process group is up (one process per GPU, RCCL), else None."""
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None


def infonce_data_parallel(emb_x: torch.Tensor, emb_y: torch.Tensor, infonce_fn, dist=None):
    """Symmetric InfoNCE of the GLOBAL batch when its rows are sharded over ranks (the negatives are in-batch, so the
    [B,B] similarity needs every rank's embeddings - SURVEY 8(f) rank 1).  Each rank all-gathers both embedding sets
    (rank order = row order of the global batch), evaluates loss and embedding gradients of the whole batch with
    `infonce_fn(emb_x_all, emb_y_all) -> (loss, g_x_all, g_y_all)` (redundantly: B^2 D work, B <= a few thousand) and keeps
    the gradient rows of its own shard.  Back-propagating those through the local encoders and all-reducing (sum) the
    parameter gradients gives exactly the single-process gradient.  Returns (loss, g_x_local, g_y_local)."""
    if dist is None:
        return infonce_fn(emb_x, emb_y)
    world, rank, b = dist.get_world_size(), dist.get_rank(), emb_x.shape[0]
    ex = torch.empty((world * b, emb_x.shape[1]), dtype=emb_x.dtype, device=emb_x.device)
    ey = torch.empty((world * b, emb_y.shape[1]), dtype=emb_y.dtype, device=emb_y.device)
    dist.all_gather_into_tensor(ex, emb_x.contiguous())
    dist.all_gather_into_tensor(ey, emb_y.contiguous())
    loss, gx, gy = infonce_fn(ex, ey)
    sl = slice(rank * b, (rank + 1) * b)
    return loss, (None if gx is None else gx[sl]), (None if gy is None else gy[sl])


def _epoch_mean(values, scalar: bool):
    """np.mean(running) / np.mean(running, axis=0) of train.py:271-274; device tensors are stacked and reduced on the device
    (one read-back per epoch boundary instead of one per step)."""
    if values and isinstance(values[0], torch.Tensor):
        m = torch.stack(values).double().mean(0).cpu().numpy()
        return float(m.reshape(-1)[0]) if scalar else m
    return np.mean(values) if scalar else np.mean(values, axis=0)


class _LossSlots:
    """1-element views of chunked device buffers: dib_infonce_fwd_bwd writes each step's loss into its own slot, the epoch
    mean stacks the views (no per-step allocation; a full chunk is simply replaced - the views keep it alive)."""

    def __init__(self, device, dtype=torch.float32, chunk: int = 4096):
        self.device, self.dtype, self.chunk, self.buf, self.used = device, dtype, chunk, None, chunk

    def next(self) -> torch.Tensor:
        if self.used >= self.chunk:
            self.buf, self.used = torch.empty(self.chunk, dtype=self.dtype, device=self.device), 0
        v = self.buf[self.used: self.used + 1]
        self.used += 1
        return v


def run_custom_loop(*, dataset_length: int, validation_set_length: int, batch_size: int, number_pretraining_epochs: int,
                    number_annealing_epochs: int, beta_start: float, beta_end: float, train_step, validation_step,
                    assign_beta, epoch_callback=None, epoch_mean=None) -> Dict[str, np.ndarray]:
    """Host bookkeeping of the reference's custom loop (train.py:222-279) around the device step functions:
    `train_step(step_num)` / `validation_step(epoch_num, batch_number)` -> (loss_infonce, kl), `assign_beta(v)` =
    model.beta.assign.  Epoch e is closed after the step whose number is round(steps_per_epoch * e) (numpy half-to-even
    rounding; when several epochs round to one step the first wins, train.py:245-247), beta comes from the float64 numpy
    formula of train.py:248, the validation pass draws number_full_validation_batches + 1 full batches (train.py:231-234) and
    the loop ends one step short of the last boundary (`take(epoch_steps[-1])`), so number_epochs - 1 epochs are recorded.
    `epoch_mean(series name, values)` (optional) replaces the plain mean of a series' per-step values: a step function that
    accumulates on the device returns placeholders and supplies the mean here.
    Pinned on the reference's own statements executed: tests/test_oracle_golden.py::test_custom_loop_accounting_*."""
    number_epochs = number_pretraining_epochs + number_annealing_epochs
    boundaries = np.round((dataset_length / batch_size) * np.arange(number_epochs)).astype(np.int32)
    first_epoch_at = {}
    for e, s in enumerate(boundaries.tolist()):
        first_epoch_at.setdefault(s, e)
    n_val_batches = validation_set_length // batch_size + 1
    log_span = np.log(beta_end) - np.log(beta_start)
    series = dict(beta=[], kl=[], loss_infonce=[], kl_validation=[], loss_infonce_validation=[])
    pending = dict(kl=[], loss_infonce=[], kl_validation=[], loss_infonce_validation=[])
    for step_num in range(int(boundaries[-1])):
        loss, kl = train_step(step_num)
        pending['loss_infonce'].append(loss)
        pending['kl'].append(kl)
        epoch_num = first_epoch_at.get(step_num)
        if epoch_num is None:
            continue
        next_beta = np.exp(np.log(beta_start) + float(max(epoch_num - number_pretraining_epochs, 0)) / number_annealing_epochs * log_span)
        series['beta'].append(next_beta)
        assign_beta(next_beta)
        if epoch_callback is not None:
            epoch_callback(epoch_num)
        for vb in range(n_val_batches):
            loss, kl = validation_step(epoch_num, vb)
            pending['loss_infonce_validation'].append(loss)
            pending['kl_validation'].append(kl)
        for k, vals in pending.items():
            series[k].append(epoch_mean(k, vals) if epoch_mean is not None else _epoch_mean(vals, scalar=k.startswith('loss')))
            pending[k] = []
    out = {k: np.asarray(v) for k, v in series.items()}
    out['beta'] = np.float32(out['beta'])               # train.py:272
    return out


def fit_infonce(model, x_train, y_train, x_valid, y_valid, *, batch_size: int, number_pretraining_epochs: int,
                number_annealing_epochs: int, beta_start: float, beta_end: float, learning_rate: float,
                y_encoder_architecture=(128, 128), shared_dimensionality: int = 64, similarity: str = 'l2',
                temperature: float = 1.0, use_positional_encoding: bool = True,
                number_positional_encoding_frequencies: int = 5, activation_fn: Optional[str] = 'relu', seed: int = 0,
                epoch_callback=None, output_encoder: Optional[DenseStack] = None,
                shuffle_buffer: int = SHUFFLE_BUFFER) -> Dict[str, np.ndarray]:
    """Returns dict(beta, kl [epochs-1, F] nats, loss_infonce, kl_validation, loss_infonce_validation) - the series the
    reference builds at train.py:237-279 (before its conversion to bits) - plus kl_total / kl_total_validation.
    The reference's own KL series is `kl_loss / model.beta` with kl_loss = model.losses, the one-element list
    [beta * sum_f KL_f] (models.py:118): ITS series are the row sums kl_total; the per-feature columns are this project's
    superset.  `output_encoder`: a pre-built Y encoder (default: a fresh DenseStack seeded with seed + 1).  `shuffle_buffer`: the
    reference's 10 000 (train.py:227); tools/stream_effect.py passes the dataset length to measure what the buffer does."""
    eng = model._ensure_engine()
    assert model.output_dimensionality == shared_dimensionality, "model output must be the shared embedding space"
    F = model.number_features
    xd, yd = eng.to_device(np.asarray(x_train, dtype=np.float32)), eng.to_device(np.asarray(y_train, dtype=np.float32))
    xvd, yvd = eng.to_device(np.asarray(x_valid, dtype=np.float32)), eng.to_device(np.asarray(y_valid, dtype=np.float32))
    yenc = output_encoder if output_encoder is not None else DenseStack(
        eng, yd.shape[1], list(y_encoder_architecture), shared_dimensionality, activation_fn,
        use_positional_encoding, number_positional_encoding_frequencies, seed=seed + 1)
    model.output_encoder = yenc
    n, nv = xd.shape[0], xvd.shape[0]
    # data parallel (one process per GPU): every rank draws the same global batch (same seed) and takes its row shard;
    # embeddings are all-gathered for the in-batch negatives, parameter gradients all-reduced (sum)
    dist = _dist()
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
    assert batch_size % world == 0, "batch_size must be divisible by the number of ranks"
    B = batch_size // world
    shard = slice(rank * B, (rank + 1) * B)
    n_val_batches = nv // batch_size + 1
    # batch order (see _BatchStream).  Row indices go to the device a block of steps at a time - one upload per _IDX_BLOCK
    # training steps / per validation pass instead of one per step.
    stream = _BatchStream(n, batch_size, seed, shuffle_buffer)
    block_steps = max(8, min(_IDX_BLOCK, (1 << 16) // batch_size))
    draw_block = lambda: np.ascontiguousarray(stream.next_batches(block_steps)[:, shard]).astype(np.int32)
    # the NEXT block is drawn on a worker thread while the device runs this one (numpy's sort / gather release the GIL): 100 ns per
    # drawn row is 0.2 ms per step at B = 2048 - half a step - if the launch thread pays it
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)
    train_idx = dict(first=0, dev=None, next=pool.submit(draw_block))
    val_idx = dict(epoch=None, dev=None)

    def train_rows(step_num):
        k = step_num - train_idx["first"]
        if train_idx["dev"] is None or k >= train_idx["dev"].shape[0]:
            blk = train_idx["next"].result()
            train_idx["next"] = pool.submit(draw_block)
            train_idx["first"], train_idx["dev"], k = step_num, eng.to_device(blk, dtype=torch.int32), 0
        return train_idx["dev"][k]

    def validation_rows(epoch_num, vb):
        if val_idx["epoch"] != epoch_num:      # `for ... in tf_dataset_validation` (train.py:262): a fresh iterator per pass
            blk = _BatchStream(nv, batch_size, [seed + 7, epoch_num], shuffle_buffer).next_batches(n_val_batches)
            val_idx["epoch"] = epoch_num
            val_idx["dev"] = eng.to_device(np.ascontiguousarray(blk[:, shard]).astype(np.int32), dtype=torch.int32)
        return val_idx["dev"][vb]

    # the learning rate is constant over this loop (train.py:128-129): one device scalar per network, set once
    eng.set_lr(learning_rate)
    yenc.set_lr(learning_rate)

    from . import _lib
    adam = ("adam", 0.9, 0.999, 1e-7)                             # tf.keras.optimizers.Adam defaults (train.py:128-129)
    slots = _LossSlots(eng.device, eng.metrics_acc.dtype)
    # single process: the per-feature KL of every step is accumulated on the device by the step's LAST launch (metrics_acc[f] +=
    # KL_f sum / B, dib_step_tail) - training and validation into accumulators of THIS loop, read once per epoch boundary.
    # (Not eng.metrics_acc: that is the accumulator model.fit's History is read from, and the loop ends one step short of the
    # last boundary - the steps after the last recorded epoch would stay in it and pollute the first epoch of a later fit.)
    kl_acc = dict(kl=torch.zeros_like(eng.metrics_acc), kl_validation=torch.zeros_like(eng.metrics_acc))

    def eval_batch(xs, ys, idx, training, step):
        if dist is None:
            # launches of a training step: encoder bank, integration network, Y encoder (gather + 3 layers), 3 InfoNCE kernels
            # (which write dL/d(embedding) where the two backward passes read it), the two backward chains, and ONE tail each
            # (partials + KL sums + accumulation + Adam + counter bump) - no torch kernel in the loop
            # (row-tile batches: the Y encoder's forward rides in the integration network's grid, its dgrad chain in the
            # integration backward's - 9 launches per training step)
            comp = yenc.companion_forward(ys, rows=idx) if hasattr(yenc, "companion_forward") else None
            ckw = {} if comp is None else dict(companion=comp)
            eng.forward(xs, idx, 0, B, model.noise_seed, step, inference=not training, defer_sums=True, **ckw)   # noise always on (train.py:263-265)
            emb_y = yenc.forward(ys, rows=idx) if comp is None else yenc.companion_output()
            loss, gx, gy = eng.infonce(eng.pred(B), emb_y, similarity, temperature, want_grads=training,
                                       out_gx=eng.g_pred(B) if training else None,
                                       out_gy=yenc.output_grad_buffer() if training else None, loss_out=slots.next())
            if training:
                compb = yenc.companion_backward(gy) if comp is not None else None
                eng.backward_from_pred_grad(gx, idx, 0, B, model.noise_seed, step, inv_global_batch=1.0 / batch_size,
                                            finish_flags=_lib.TAIL_KL | _lib.TAIL_METRICS, optimizer=adam, metrics_acc=kl_acc["kl"],
                                            **({} if compb is None else dict(companion=compb)))
                yenc.backward(gy, reduce=False, **({} if compb is None else dict(dgrad_done=True)))
                yenc.adam_step(fused_reduce=True)                  # one Keras Adam over all variables (train.py:196,219)
            else:
                eng.step_tail(B, -1, _lib.TAIL_KL | _lib.TAIL_METRICS, 1.0 / batch_size, metrics_acc=kl_acc["kl_validation"])
            return loss, None
        eng.forward(xs, idx, 0, B, model.noise_seed, step, inference=not training)  # noise always on (train.py:263-265)
        emb_x = eng.pred(B)
        emb_y = yenc.forward(ys, rows=idx)                         # gathered straight into the encoder's workspace
        loss, gx, gy = infonce_data_parallel(
            emb_x, emb_y, lambda a, b: eng.infonce(a, b, similarity, temperature, want_grads=training), dist)
        kl = eng.step_out(B)[:F] * (1.0 / B)                       # kl_loss / beta (train.py:220), per feature
        dist.all_reduce(kl)
        kl /= world
        if training:
            eng.backward_from_pred_grad(gx, idx, 0, B, model.noise_seed, step, inv_global_batch=1.0 / batch_size)
            yenc.backward(gy)
            dist.all_reduce(eng.grads)
            dist.all_reduce(yenc.grads)
            eng.adam_step()                                        # one Keras Adam over all variables (train.py:196,219)
            yenc.adam_step()
        return loss, kl

    def epoch_mean(name, values):
        if dist is None and name in kl_acc:                        # mean over the epoch's steps of the per-step batch means
            acc = kl_acc[name]
            m = acc[:F].double().cpu().numpy() / max(len(values), 1)
            acc.zero_()
            return m
        return _epoch_mean(values, scalar=name.startswith('loss'))

    eng.set_beta(float(model.beta.value()))                        # the first step runs at the constructor's beta (models.py:86)
    try:
        out = run_custom_loop(
            dataset_length=n, validation_set_length=nv, batch_size=batch_size, number_pretraining_epochs=number_pretraining_epochs,
            number_annealing_epochs=number_annealing_epochs, beta_start=beta_start, beta_end=beta_end,
            train_step=lambda step_num: eval_batch(xd, yd, train_rows(step_num), True, step_num),
            validation_step=lambda epoch_num, vb: eval_batch(xvd, yvd, validation_rows(epoch_num, vb), False,
                                                             (1 << 31) + epoch_num * 1024 + vb),
            assign_beta=model.beta.assign,
            epoch_callback=(lambda e: epoch_callback(e, model)) if epoch_callback is not None else None, epoch_mean=epoch_mean)
    finally:   # also when a step or the caller's epoch_callback raises: no worker thread / pending draw left behind
        train_idx["next"].cancel()
        pool.shutdown(wait=True, cancel_futures=True)
    out['kl_total'], out['kl_total_validation'] = out['kl'].sum(-1), out['kl_validation'].sum(-1)
    return out
