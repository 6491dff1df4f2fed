"""CPU oracle of the reference's CUSTOM TRAINING LOOP (InfoNCE path, BASELINE config 2).  TEST INFRASTRUCTURE ONLY
(see oracle/dib_oracle.py header) - never imported by the product.

Restates reference train.py:180-289 in float64:
  * train.py:184-192  the Y ("output") encoder: [PositionalEncoding] -> Dense(units, act)* -> Dense(shared dim)
  * train.py:196      ONE list of trainable variables (X model + Y encoder) ...
  * train.py:201-220  ... one GradientTape over both networks, symmetric InfoNCE (mean CE of S rows + mean CE of S^T rows,
                      labels arange(B)) + sum(model.losses) [= beta * sum_f KL_f, models.py:118], ONE optimizer.apply_gradients
                      (Keras Adam), returns (loss_infonce, kl_loss / beta)
  * train.py:222-236  full batches only from a repeating shuffled stream; epoch boundaries at round(steps_per_epoch * arange(E))
  * train.py:240-250  step loop over epoch_steps[-1] steps; at a boundary the numpy (float64) beta formula, model.beta.assign
  * train.py:262-279  validation with noise ON over number_full_validation_batches + 1 batches, epoch means, reset

PARITY PINNING STATUS
  * The epoch / series ACCOUNTING (`run_loop`) is pinned on the reference's own loop statements, lifted from train.py:236-279
    and EXECUTED (tests/golden/make_golden_infonce_loop.py -> tests/golden/infonce_loop.npz; checked by
    tests/test_oracle_golden.py): same step function stubs in, same series out, incl. the first step running at the
    constructor's beta = 1 (models.py:86), non-integer steps_per_epoch with banker's rounding, E - 1 recorded epochs.
  * The similarity functions are pinned on utils.py:75-175 executed (tests/golden/models_forward.npz, round 2).
  * tf.data's shuffle-buffer order, tf.random.normal's stream and Keras Adam internals cannot be executed here (no
    TensorFlow): the batch order follows tf.data's documented shuffle-buffer ALGORITHM (`BatchStream`: buffer of min(n, 10 000)
    over the repeating sequential stream, uniform slot, refill; the validation dataset re-iterated from row 0 at every
    boundary) with numpy's random numbers in place of TensorFlow's, the noise is the counter-based Philox stream shared by
    product and oracle, Adam is the Keras form restated from its documentation - parity unpinned for those three, as
    everywhere else in this repo.
  * One deliberate superset: the reference's `kl_loss / model.beta` is the ONE-element list [sum_f KL_f] (model.losses holds a
    single add_loss term), so its kl series are [epochs-1, 1]; product and oracle record the per-feature vector [epochs-1, F]
    whose row sums are the reference's series (`kl_total`).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

import dib_oracle as orc
from dib_torch_cpu import TorchCpuDIB, _act, scaled_similarity_torch


class BatchStream:
    """tf_dataset.repeat().shuffle(min(n, 10_000)).batch(batch_size) (train.py:226-227, 233-234), literally, one element at a
    time: the source is the sequential, repeating stream of dataset rows 0, 1, ..., n-1, 0, 1, ...; tf.data's shuffle holds a
    buffer of min(n, 10_000) elements, filled from the source in order; each output element is a uniformly chosen slot of the
    buffer, and that slot is refilled with the next source element.  (TensorFlow's own random numbers are not reproducible
    here - and unseeded in the reference: slot j of draw t = default_rng(seed).integers(0, buffer), the t-th call.)"""

    def __init__(self, n: int, batch_size: int, seed, buffer_size: int = 10_000):
        self.n, self.bs = int(n), int(batch_size)
        self.rng = np.random.default_rng(seed)
        self.buffer: List[int] = [i % self.n for i in range(min(self.n, int(buffer_size)))]
        self.source_position = len(self.buffer)

    def next(self) -> np.ndarray:
        batch = []
        for _ in range(self.bs):
            slot = int(self.rng.integers(0, len(self.buffer)))
            batch.append(self.buffer[slot])
            self.buffer[slot] = self.source_position % self.n
            self.source_position += 1
        return np.asarray(batch, dtype=np.int64)


def beta_at_boundary(epoch_num: int, beta_start: float, beta_end: float, number_pretraining_epochs: int,
                     number_annealing_epochs: int) -> float:
    """train.py:248 - numpy float64 maths (unlike the Keras callback, models.py:147-149, which is float32)."""
    return float(np.exp(np.log(beta_start) + float(max(epoch_num - number_pretraining_epochs, 0)) / number_annealing_epochs *
                        (np.log(beta_end) - np.log(beta_start))))


def run_loop(*, dataset_length: int, validation_set_length: int, batch_size: int, number_pretraining_epochs: int,
             number_annealing_epochs: int, beta_start: float, beta_end: float,
             train_step: Callable[[int], tuple], validation_step: Callable[[int, int], tuple],
             assign_beta: Callable[[float], None], on_boundary: Optional[Callable[[int], None]] = None) -> Dict[str, np.ndarray]:
    """The bookkeeping of train.py:222-279 around abstract step functions.
    train_step(step_num) -> (loss_infonce, kl) ; validation_step(epoch_num, batch_number) -> (loss_infonce, kl);
    assign_beta(value) = model.beta.assign.  kl may be a scalar or a vector (mean is taken over axis 0, train.py:276-277).
    on_boundary(epoch_num) (not in the reference): called after an epoch's series entries are recorded - the trajectory tests
    re-synchronise the checker's state with the device's there (tests/test_gpu_trajectories.py)."""
    number_epochs = number_pretraining_epochs + number_annealing_epochs
    steps_per_epoch = dataset_length / batch_size                                            # train.py:224
    number_full_validation_batches = validation_set_length // batch_size                     # train.py:231
    epoch_steps = np.round(steps_per_epoch * np.arange(number_epochs)).astype(np.int32)      # train.py:236
    series = dict(beta=[], loss_infonce=[], loss_infonce_validation=[], kl=[], kl_validation=[])
    run_l, run_lv, run_k, run_kv = [], [], [], []
    for step_num in range(int(epoch_steps[-1])):                                             # tf_dataset.take(epoch_steps[-1])
        l, k = train_step(step_num)
        run_l.append(l)
        run_k.append(k)
        if step_num in epoch_steps:                                                          # train.py:245
            epoch_num = int(np.where(epoch_steps == step_num)[0][0])
            next_beta = beta_at_boundary(epoch_num, beta_start, beta_end, number_pretraining_epochs, number_annealing_epochs)
            series["beta"].append(next_beta)
            assign_beta(next_beta)
            for vb in range(number_full_validation_batches + 1):                             # .take(n_full + 1), train.py:234
                lv, kv = validation_step(epoch_num, vb)
                run_lv.append(lv)
                run_kv.append(kv)
            series["loss_infonce"].append(np.mean(run_l))
            series["loss_infonce_validation"].append(np.mean(run_lv))
            series["kl"].append(np.mean(run_k, axis=0))
            series["kl_validation"].append(np.mean(run_kv, axis=0))
            run_l, run_lv, run_k, run_kv = [], [], [], []
            if on_boundary is not None:
                on_boundary(epoch_num)
    out = {k: np.asarray(v) for k, v in series.items()}
    out["beta"] = np.float32(out["beta"])                                                    # train.py:272
    return out


class YEncoder:
    """train.py:184-192 in float64 torch: kernels [in, out] (Keras), y = act(x @ W + b), linear last layer."""

    def __init__(self, kernels: Sequence[np.ndarray], biases: Sequence[np.ndarray], activation: Optional[str],
                 use_positional_encoding: bool, number_positional_encoding_frequencies: int, dtype=torch.float64):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True)
        self.W, self.b = [t(w) for w in kernels], [t(b) for b in biases]
        self.act = _act(activation)
        # train.py:186-187: 2**np.arange(1, n) -> n - 1 frequencies
        self.freqs = [float(f) for f in 2 ** np.arange(1, number_positional_encoding_frequencies)] if use_positional_encoding else []

    def tensors(self) -> List[torch.Tensor]:
        return [t for wb in zip(self.W, self.b) for t in wb]

    def forward(self, y: torch.Tensor) -> torch.Tensor:
        h = y
        if self.freqs:
            h = torch.cat([h] + [torch.sin(fr * h) for fr in self.freqs], -1)                # models.py:22-23
        for l, (w, b) in enumerate(zip(self.W, self.b)):
            h = h @ w + b
            if l < len(self.W) - 1:
                h = self.act(h)
        return h


class InfoNCELoopOracle:
    """eval_batch_infonce (train.py:201-220) + the loop, float64, with the product's noise / batch-order conventions:
    eps keyed by (noise_seed, step key, DATASET row, feature, dim); training step key = step_num, validation step key =
    2^31 + 1024 * epoch_num + batch_number; ONE BatchStream(seed) for the training rows of the whole run; the validation
    dataset object is iterated anew at every boundary (`for ... in tf_dataset_validation`, train.py:262): a fresh
    BatchStream([seed + 7, epoch_num]) per validation pass, source restarting at row 0."""

    def __init__(self, spec: orc.DIBSpec, x_params: orc.DIBParams, y_encoder: YEncoder, similarity: str, temperature: float,
                 learning_rate: float, noise_seed: int, dtype=torch.float64):
        """dtype: float64 is the checker; float32 exists only to MEASURE how far two float32-accurate runs of this chaotic
        recursion (Adam on ReLU networks) drift apart, which is what the trajectory tolerances are derived from."""
        self.spec, self.yenc, self.dtype = spec, y_encoder, dtype
        self.model = TorchCpuDIB(spec, x_params, dtype=dtype)
        self.similarity, self.temperature, self.lr, self.noise_seed = similarity, float(temperature), float(learning_rate), noise_seed
        self.beta = np.float32(1.0)                                                          # models.py:86: tf.Variable(1.)
        self.vars = self.model.tensors() + self.yenc.tensors()                               # train.py:196
        self.m = [torch.zeros_like(p) for p in self.vars]
        self.v = [torch.zeros_like(p) for p in self.vars]
        self.t = 0

    def assign_beta(self, value: float) -> None:
        self.beta = np.float32(value)                                                        # float32 tf.Variable

    def load_state(self, params: Sequence[np.ndarray], m: Sequence[np.ndarray], v: Sequence[np.ndarray], t: int) -> None:
        """Overwrite every variable, both Adam moments and the Adam step count (order of self.vars: X model tensors in
        TorchCpuDIB.tensors() order, then the Y encoder's kernel / bias pairs)."""
        assert len(params) == len(m) == len(v) == len(self.vars)
        with torch.no_grad():
            for dst, src in zip(self.vars, params):
                dst.copy_(torch.as_tensor(np.asarray(src), dtype=self.dtype).reshape(dst.shape))
            for dst, src in zip(self.m, m):
                dst.copy_(torch.as_tensor(np.asarray(src), dtype=self.dtype).reshape(dst.shape))
            for dst, src in zip(self.v, v):
                dst.copy_(torch.as_tensor(np.asarray(src), dtype=self.dtype).reshape(dst.shape))
        self.t = int(t)

    def _adam(self, grads, b1=0.9, b2=0.999, e=1e-7) -> None:
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        with torch.no_grad():
            for p, g, m, v in zip(self.vars, grads, self.m, self.v):
                m.add_((1 - b1) * (g - m))
                v.add_((1 - b2) * (g * g - v))
                p.sub_(lr_t * m / (torch.sqrt(v) + e))

    def eval_batch(self, x: np.ndarray, y: np.ndarray, rows: np.ndarray, step_key: int, training: bool):
        """-> (loss_infonce float, kl [F] nats).  x, y: whole dataset arrays; rows: dataset row indices of the batch."""
        s = self.spec
        xb = torch.tensor(x[rows], dtype=self.dtype)
        yb = torch.tensor(y[rows], dtype=self.dtype)
        eps = torch.tensor(orc.philox_normal_all(self.noise_seed, step_key, rows.astype(np.uint32), s.number_features,
                                                 s.feature_embedding_dimension), dtype=self.dtype)
        with torch.set_grad_enabled(training):
            ex, kl = self.model.forward(xb, eps)                                             # models.py:96-123
            ey = self.yenc.forward(yb)
            S = scaled_similarity_torch(ex, ey, self.similarity, self.temperature)           # utils.py:131-175
            d = torch.diagonal(S)
            loss_infonce = (torch.logsumexp(S, 1) - d).mean() + (torch.logsumexp(S, 0) - d).mean()  # train.py:209-214
            loss = loss_infonce + float(self.beta) * kl.sum()                                # train.py:215-216 (models.py:118)
        if training:
            self._adam(torch.autograd.grad(loss, self.vars))                                 # train.py:218-219
        return float(loss_infonce.detach()), kl.detach().numpy().copy()

    def fit(self, x_train, y_train, x_valid, y_valid, *, batch_size: int, number_pretraining_epochs: int,
            number_annealing_epochs: int, beta_start: float, beta_end: float, seed: int = 0,
            on_boundary: Optional[Callable[[int], None]] = None) -> Dict[str, np.ndarray]:
        x_train, y_train, x_valid, y_valid = [np.asarray(a, dtype=np.float32).astype(np.float64)
                                              for a in (x_train, y_train, x_valid, y_valid)]
        stream = BatchStream(len(x_train), batch_size, seed)
        vstreams: Dict[int, BatchStream] = {}

        def validation_rows(ep: int) -> np.ndarray:
            if ep not in vstreams:
                vstreams.clear()
                vstreams[ep] = BatchStream(len(x_valid), batch_size, [seed + 7, ep])
            return vstreams[ep].next()

        out = run_loop(
            dataset_length=len(x_train), validation_set_length=len(x_valid), batch_size=batch_size,
            number_pretraining_epochs=number_pretraining_epochs, number_annealing_epochs=number_annealing_epochs,
            beta_start=beta_start, beta_end=beta_end,
            train_step=lambda step: self.eval_batch(x_train, y_train, stream.next(), step, True),
            validation_step=lambda ep, vb: self.eval_batch(x_valid, y_valid, validation_rows(ep), (1 << 31) + ep * 1024 + vb, False),
            assign_beta=self.assign_beta, on_boundary=on_boundary)
        out["kl_total"] = out["kl"].sum(-1)                                                  # the reference's [sum_f KL_f] series
        out["kl_total_validation"] = out["kl_validation"].sum(-1)
        return out
