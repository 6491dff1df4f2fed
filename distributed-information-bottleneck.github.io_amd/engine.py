"""HipEngine: device state + launch sequencing of the Distributed-IB step on one MI355X.

PyTorch is used only as plumbing (device memory, streams, torch.distributed); every FLOP of the
path runs in libdib_hip.so (hand-written HIP for gfx950) through the C ABI in include/dib_hip.h.
There is no CPU / eager fallback: constructing a HipEngine without a GPU or without the library
raises.
"""
from __future__ import annotations

import ctypes
import math
from ctypes import byref, c_int, c_int64, c_void_p
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import ACTIVATIONS, LOSS_KINDS, check


def _ptr(t: Optional[torch.Tensor]):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


class HipEngine:
    """Owns the flat parameter / gradient / Adam buffers and the activation workspace.

    The architecture arguments mirror DistributedIBNet.__init__ (reference models.py:56-66).
    """

    def __init__(self, feature_dimensionalities: Sequence[int], feature_encoder_architecture: Sequence[int],
                 integration_network_architecture: Sequence[int], output_dimensionality: int,
                 use_positional_encoding: bool = True, number_positional_encoding_frequencies: int = 5,
                 activation_fn: Optional[str] = "relu", feature_embedding_dimension: int = 32,
                 output_activation_fn: Optional[str] = None, device: Optional[str] = None, init_seed: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs an AMD GPU (torch.cuda.is_available() is False); "
                               "the Distributed-IB path has no CPU fallback")
        self.lib = _lib.load_library()
        if activation_fn not in ACTIVATIONS or output_activation_fn not in ACTIVATIONS:
            raise ValueError(f"unsupported activation {activation_fn!r}/{output_activation_fn!r}; "
                             f"supported: {sorted(k for k in ACTIVATIONS if isinstance(k, str))}")
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.F = len(feature_dimensionalities)
        self.E = int(feature_embedding_dimension)
        self.dims = [int(d) for d in feature_dimensionalities]
        self.sum_d = sum(self.dims)
        self.out_dim = int(output_dimensionality)
        self.enc_units = [int(u) for u in feature_encoder_architecture]
        self.int_units = [int(u) for u in integration_network_architecture]
        ci = lambda xs: (c_int * max(1, len(xs)))(*xs)
        handle = c_void_p()
        check(self.lib.dib_layout_create(self.F, ci(self.dims), len(self.enc_units), ci(self.enc_units), self.E,
                                         len(self.int_units), ci(self.int_units), self.out_dim,
                                         1 if use_positional_encoding else 0,
                                         int(number_positional_encoding_frequencies), ACTIVATIONS[activation_fn],
                                         ACTIVATIONS[output_activation_fn], byref(handle)), "dib_layout_create")
        self.layout = handle
        self.n_params = int(self.lib.dib_layout_param_count(self.layout))
        n_alloc = (self.n_params + 3) // 4 * 4
        with torch.cuda.device(self.device):
            z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.device)
            self.params, self.grads, self.adam_m, self.adam_v = z(n_alloc), z(n_alloc), z(n_alloc), z(n_alloc)
            self.beta_dev = torch.ones(1, dtype=torch.float32, device=self.device)  # reference models.py:86
            self.lr_dev = torch.full((1,), 1e-3, dtype=torch.float32, device=self.device)
            self._lr_host: Optional[float] = 1e-3
            self.t_dev = z(1, torch.int64)
            # History accumulators: training | validation, one tensor (one device-to-host copy reads both: fit synchronises
            # once per epoch - the validation pass is enqueued behind the training steps without reading anything back)
            self._metrics_stride = (self.F + 3 + 3) // 4 * 4
            self._metrics2 = z(2 * self._metrics_stride)
            self.metrics_acc = self._metrics2[: self.F + 3]
            self.metrics_acc_val = self._metrics2[self._metrics_stride: self._metrics_stride + self.F + 3]
            tb = int(self.lib.dib_layout_table_bytes(self.layout))
            self._tables = torch.zeros(tb, dtype=torch.uint8, device=self.device)
            check(self.lib.dib_layout_upload_tables(self.layout, _ptr(self._tables), self._stream()),
                  "dib_layout_upload_tables")
            torch.cuda.synchronize(self.device)
        self._ws: Dict[int, torch.Tensor] = {}      # batch -> step workspace, insertion order = recency
        self._ws_pinned = set()                     # batch sizes whose workspace a captured graph points into
        self._ws_gen: Dict[int, int] = {}           # batch -> number of forwards that (re)wrote its activations
        self._scratch: Optional[torch.Tensor] = None
        self._fused_head: Dict[int, bool] = {}      # loss kind -> dib_output_head_fused_supported
        self._infonce_ws: Dict[int, torch.Tensor] = {}   # batch -> workspace of dib_infonce_fwd_bwd
        self.blocks = self._query_blocks()
        self.set_flat_params(self.glorot_uniform(init_seed))

    # ---- plumbing -----------------------------------------------------------------------------
    def _stream(self):
        return c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "layout", None):
                self.lib.dib_layout_destroy(self.layout)
                self.layout = None
        except Exception:
            pass

    _WS_KEEP = 4  # unpinned step workspaces kept (train, tail, validation, validation tail)

    def workspace(self, batch: int) -> torch.Tensor:
        """Step workspace for `batch` rows: true LRU over the most recent batch sizes.  A workspace whose raw pointer
        is baked into a captured hipGraph is PINNED (capture_step_graph) and never evicted - freeing it would let the
        caching allocator hand the memory to another tensor while graph replays still write to it."""
        ws = self._ws.pop(batch, None)
        if ws is None:
            nbytes = int(self.lib.dib_workspace_bytes(self.layout, batch))
            if nbytes <= 0:
                raise _lib.DibError(f"dib_workspace_bytes({batch}) -> {nbytes}")
            # include/dib_hip.h contract: the split-batch gradient slabs that no launch writes must read as zero
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            check(self.lib.dib_workspace_init(self.layout, batch, _ptr(ws), self._stream()), "dib_workspace_init")
            unpinned = [b for b in self._ws if b not in self._ws_pinned]
            while len(unpinned) >= self._WS_KEEP:
                self._ws.pop(unpinned.pop(0))  # least recently used first
            self._ws_gen[batch] = self._ws_gen.get(batch, 0) + 1  # a fresh buffer holds nobody's activations
        self._ws[batch] = ws  # (re)insert at the most-recent end
        return ws

    def scratch(self, nbytes: int) -> torch.Tensor:
        """Grow-only scratch for the evaluation helpers (encode_feature, MI bounds): they never touch the step
        workspaces, so callbacks cannot evict or clobber a training workspace (or one a graph / autograd node owns)."""
        n = (int(nbytes) + 3) // 4
        if self._scratch is None or self._scratch.numel() < n:
            self._scratch = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._scratch

    def ws_view(self, batch: int, which: int, numel: int) -> torch.Tensor:
        off = int(self.lib.dib_workspace_offset(self.layout, batch, which))
        if off < 0:
            raise _lib.DibError(f"dib_workspace_offset -> {off}")
        return self.workspace(batch)[off // 4: off // 4 + numel]

    def _query_blocks(self) -> List[dict]:
        out = []
        off, rows, cols = c_int64(), c_int(), c_int()
        for net, nl in ((0, len(self.enc_units) + 1), (1, len(self.int_units) + 1)):
            for layer in range(nl):
                for f in (range(self.F) if net == 0 else [0]):
                    for what in (0, 1):
                        check(self.lib.dib_layout_param_block(self.layout, net, layer, f, what, byref(off), byref(rows),
                                                              byref(cols)), "dib_layout_param_block")
                        out.append(dict(net=net, layer=layer, feature=f, what=what, offset=off.value, rows=rows.value,
                                        cols=cols.value))
        return out

    def glorot_uniform(self, seed: int) -> np.ndarray:
        """Keras Dense defaults: kernel U(+-sqrt(6/(in+out))), bias zeros."""
        rng = np.random.default_rng(seed)
        flat = np.zeros(self.params.numel(), dtype=np.float32)
        for b in self.blocks:
            if b["what"] == 0:
                lim = math.sqrt(6.0 / (b["rows"] + b["cols"]))
                n = b["rows"] * b["cols"]
                flat[b["offset"]: b["offset"] + n] = rng.uniform(-lim, lim, size=n).astype(np.float32)
        return flat

    def set_flat_params(self, flat: np.ndarray) -> None:
        flat = np.asarray(flat, dtype=np.float32).reshape(-1)
        buf = np.zeros(self.params.numel(), dtype=np.float32)
        buf[: min(len(flat), len(buf))] = flat[: len(buf)]
        self.params.copy_(torch.from_numpy(buf))

    def get_flat_params(self) -> np.ndarray:
        return self.params.detach().cpu().numpy().copy()

    def get_flat_grads(self) -> np.ndarray:
        return self.grads.detach().cpu().numpy().copy()

    def reset_optimizer(self) -> None:
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.t_dev.zero_()

    def set_beta(self, v: float) -> None:
        self.beta_dev.fill_(float(v))

    def get_beta(self) -> float:
        return float(self.beta_dev.item())

    def set_lr(self, v: float) -> None:
        """device learning rate; a repeated value costs nothing (fit sets it every step - a 4 us fill kernel per step of the
        reference's default 140 us B = 128 step otherwise).  `lr_dev` is written ONLY here: the host-side cache below is
        valid by construction (code that must write the device scalar itself calls invalidate_lr_cache() afterwards)."""
        if self._lr_host != float(v):
            self.lr_dev.fill_(float(v))
            self._lr_host = float(v)

    def invalidate_lr_cache(self) -> None:
        self._lr_host = None

    def to_device(self, a, dtype=torch.float32) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(device=self.device, dtype=dtype).contiguous()

    # ---- the step -------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, row_idx: Optional[torch.Tensor], row0: int, batch: int, seed: int, step: int,
                deterministic: bool = False, inference: bool = False, hidden_only: bool = False,
                defer_sums: bool = False, companion=None) -> None:
        """reference models.py:96-123 -> ws[U], ws[PRED], KL local sums in ws[STEP_OUT].  inference=True: no backward
        follows (validation / predict), the fused forward skips the stashes it would write for it.  defer_sums=True: the KL
        column sums are left to the step's tail launch (step_tail with TAIL_KL).  companion: DenseStack.companion_forward(...)
        - the custom loop's output encoder runs in the integration network's grid (dib_integration_fwd_and_mlp_fwd)."""
        self.encoder_forward(x, row_idx, row0, batch, seed, step, deterministic, inference, defer_sums)
        ws, st = self.workspace(batch), self._stream()
        if companion is not None:
            assert not hidden_only
            check(self.lib.dib_integration_fwd_and_mlp_fwd(self.layout, batch, _ptr(self.params), _ptr(ws), *companion, st),
                  "dib_integration_fwd_and_mlp_fwd")
        elif hidden_only:  # the output layer is evaluated by the fused head together with the loss
            check(self.lib.dib_integration_fwd_hidden(self.layout, batch, _ptr(self.params), _ptr(ws), st),
                  "dib_integration_fwd_hidden")
        else:
            check(self.lib.dib_integration_fwd(self.layout, batch, _ptr(self.params), _ptr(ws), st), "dib_integration_fwd")

    def encoder_forward(self, x: torch.Tensor, row_idx: Optional[torch.Tensor], row0: int, batch: int, seed: int, step: int,
                        deterministic: bool = False, inference: bool = False, defer_sums: bool = False) -> None:
        """reference models.py:101-115 -> ws[ENC_OUT], ws[U], the KL partials (dib_encoder_bank_fwd)"""
        ws = self.workspace(batch)
        self._ws_gen[batch] += 1
        check(self.lib.dib_encoder_bank_fwd(self.layout, _ptr(x), x.stride(0), _ptr(row_idx), int(row0), batch,
                                            _ptr(self.params), int(seed), int(step) & 0xFFFFFFFF,
                                            (_lib.FWD_DETERMINISTIC if deterministic else 0) | (_lib.FWD_INFERENCE if inference else 0)
                                            | (_lib.FWD_DEFER_SUMS if defer_sums else 0), _ptr(ws), self._stream()),
              "dib_encoder_bank_fwd")

    def loss(self, loss_kind: str, y: torch.Tensor, row_idx, row0: int, batch: int, inv_global_batch: float,
             defer_sums: bool = False) -> None:
        ws = self.workspace(batch)
        check(self.lib.dib_loss_fwd_bwd(self.layout, LOSS_KINDS[loss_kind], _ptr(y), y.stride(0), _ptr(row_idx),
                                        int(row0), batch, float(inv_global_batch), _lib.HEAD_DEFER_SUMS if defer_sums else 0,
                                        _ptr(ws), self._stream()),
              "dib_loss_fwd_bwd")

    def step_tail(self, batch: int, part: int, flags: int, inv_global_batch: float = 0.0, optimizer=None,
                  grad_scale: float = 1.0, metrics_acc: Optional[torch.Tensor] = None) -> None:
        """ONE launch for the end of a step (include/dib_hip.h dib_step_tail): any of bucket finalize, the fused head's
        weight-gradient reduce, KL / loss sums, metric accumulation, the optimizer on the bucket and the step-count bump.
        optimizer: ("adam", beta_1, beta_2, epsilon) or ("sgd",) - adds TAIL_ADAM / TAIL_SGD to `flags`."""
        b1, b2, eps = 0.9, 0.999, 1e-7
        if optimizer is not None:
            if optimizer[0] == "adam":
                flags |= _lib.TAIL_ADAM
                b1, b2, eps = (float(v) for v in optimizer[1:4])
            elif optimizer[0] == "sgd":
                flags |= _lib.TAIL_SGD
            else:
                raise ValueError(f"optimizer {optimizer[0]!r}")
        check(self.lib.dib_step_tail(self.layout, batch, int(part), int(flags), _ptr(self.params), _ptr(self.grads),
                                     _ptr(self.adam_m), _ptr(self.adam_v), _ptr(self.lr_dev), _ptr(self.t_dev), b1, b2, eps,
                                     float(grad_scale), _ptr(self.beta_dev), float(inv_global_batch),
                                     _ptr(self.metrics_acc if metrics_acc is None else metrics_acc),
                                     _ptr(self.workspace(batch)), self._stream()), "dib_step_tail")

    def optimizer_step_part(self, batch: int, part: int, optimizer, bump: bool) -> None:
        """the optimizer on ONE gradient bucket (after its all-reduce): the data-parallel fit steps buckets 1 and 2 while
        bucket 3 is still on the wire; the launch with bump=True (the last) advances Adam's step count."""
        self.step_tail(batch, part, _lib.TAIL_BUMP if (bump and optimizer[0] == "adam") else 0, optimizer=optimizer)

    def part_range(self, part: int):
        """(offset, count) of gradient bucket `part` in the flat buffers (include/dib_hip.h): 0 = encoder bank,
        1 = integration network, 2 = encoder layers before the last, 3 = last encoder layer (0 = 2 + 3)."""
        off, cnt = c_int64(), c_int64()
        check(self.lib.dib_layout_part_range(self.layout, part, byref(off), byref(cnt)), "dib_layout_part_range")
        return off.value, cnt.value

    def backward(self, row_idx, row0: int, batch: int, seed: int, step: int, inv_global_batch: float,
                 on_integration_grads_ready=None, hidden_only: bool = False, on_encoder_front_grads_ready=None,
                 finish_flags: int = 0, optimizer=None, integration_done: bool = False, companion=None,
                 metrics_acc: Optional[torch.Tensor] = None) -> None:
        """Backward pass, with the hooks of the data-parallel bucket protocol:
        `on_integration_grads_ready(grads_slice)` is called as soon as the integration network's gradients (bucket 1) are
        final - right after dib_integration_bwd - so their all-reduce runs under the whole encoder-bank backward;
        `on_encoder_front_grads_ready(grads_slice)` (three-bucket protocol, needs the first hook too) is called when the
        gradients of the encoder layers before the last (bucket 2) are final; the last layer's weight gradient (bucket 3)
        is computed after it, under that all-reduce, and is the only part left for the caller to reduce afterwards.
        Every bucket is finalised by ONE dib_step_tail launch; the LAST of them also carries `finish_flags` (the step's
        deferred KL / loss sums, the metric accumulation - into `metrics_acc`, default the model's History accumulator) and,
        without hooks, the optimizer (`optimizer`, see step_tail)."""
        ws = self.workspace(batch)
        st = self._stream()
        FIN = _lib.TAIL_FINALIZE
        head = _lib.TAIL_HEAD_WGRAD if hidden_only else 0   # the fused head left its weight-gradient partials to the tail
        if on_integration_grads_ready is None and (integration_done or not hidden_only):
            # no bucket protocol: the rest of the backward pass in one entry - for small batches the dgrad chains are one
            # launch each and ALL weight gradients one grouped launch - then ONE tail launch
            assert on_encoder_front_grads_ready is None
            bflags = _lib.BWD_INTEGRATION_DONE if integration_done else 0
            if companion is not None:   # DenseStack.companion_backward(...): the output encoder's dgrad chain in the same grid
                check(self.lib.dib_backward_and_mlp_bwd(self.layout, batch, _ptr(self.params), _ptr(self.grads), _ptr(self.beta_dev),
                                                        float(inv_global_batch), bflags, _ptr(ws), *companion, st),
                      "dib_backward_and_mlp_bwd")
            else:
                check(self.lib.dib_backward(self.layout, batch, _ptr(self.params), _ptr(self.grads), _ptr(self.beta_dev),
                                            float(inv_global_batch), bflags, _ptr(ws), st), "dib_backward")
            self.step_tail(batch, -1, FIN | head | finish_flags | (_lib.TAIL_BUMP if optimizer and optimizer[0] == "adam" else 0),
                           inv_global_batch, optimizer=optimizer, metrics_acc=metrics_acc)
            return
        assert companion is None, "a companion pass rides on dib_backward (no bucket hooks)"
        if not integration_done:   # (dib_integration_head_step already ran the integration network's backward)
            fn = self.lib.dib_integration_bwd_hidden if hidden_only else self.lib.dib_integration_bwd
            check(fn(self.layout, batch, _ptr(self.params), _ptr(self.grads), _ptr(ws), st), "dib_integration_bwd")
        if on_integration_grads_ready is not None:
            assert optimizer is None, "the data-parallel caller steps the optimizer after its all-reduces"
            self.step_tail(batch, 1, FIN | head)
            off, cnt = self.part_range(1)
            on_integration_grads_ready(self.grads[off: off + cnt])
        # (row_idx / row0 / seed / step are not needed by the device backward: eps * sigma = ws[U] - mu)
        if on_encoder_front_grads_ready is not None:
            assert on_integration_grads_ready is not None, "the three-bucket protocol needs both hooks"
            for stage, part in ((1, 2), (2, 3)):
                check(self.lib.dib_encoder_bank_bwd_stage(self.layout, batch, _ptr(self.params), _ptr(self.grads),
                                                          _ptr(self.beta_dev), float(inv_global_batch), stage, _ptr(ws), st),
                      "dib_encoder_bank_bwd_stage")
                self.step_tail(batch, part, FIN | (finish_flags if stage == 2 else 0), inv_global_batch, metrics_acc=metrics_acc)
                if stage == 1:
                    off, cnt = self.part_range(2)
                    on_encoder_front_grads_ready(self.grads[off: off + cnt])
            return
        check(self.lib.dib_encoder_bank_bwd(self.layout, batch, _ptr(self.params), _ptr(self.grads),
                                            _ptr(self.beta_dev), float(inv_global_batch), _ptr(ws), st), "dib_encoder_bank_bwd")
        if on_integration_grads_ready is not None:
            self.step_tail(batch, 0, FIN | finish_flags, inv_global_batch, metrics_acc=metrics_acc)
        else:
            self.step_tail(batch, -1, FIN | head | finish_flags | (_lib.TAIL_BUMP if optimizer and optimizer[0] == "adam" else 0),
                           inv_global_batch, optimizer=optimizer, metrics_acc=metrics_acc)

    def accumulate_metrics(self, batch: int, inv_global_batch: float) -> None:
        check(self.lib.dib_metrics_accumulate(self.layout, batch, _ptr(self.beta_dev), float(inv_global_batch),
                                              _ptr(self.metrics_acc), _ptr(self.workspace(batch)), self._stream()),
              "dib_metrics_accumulate")

    def _head_fused(self, kind: int) -> bool:
        fused_head = self._fused_head.get(kind)
        if fused_head is None:
            fused_head = self._fused_head[kind] = bool(self.lib.dib_output_head_fused_supported(self.layout, kind))
        return fused_head

    fused_optimizer_tail = True   # train_step(optimizer=...) applies the optimizer in the step's last launch

    def train_step(self, x, y, row_idx, row0: int, batch: int, seed: int, step: int, loss_kind: str,
                   inv_global_batch: Optional[float] = None, accumulate: bool = True,
                   on_integration_grads_ready=None, on_encoder_front_grads_ready=None, optimizer=None) -> None:
        """fwd + loss + bwd for the local rows; grads (partial sums over local rows / B_global) land in
        self.grads, ready for the data-parallel all-reduce(sum) and the optimizer step.  optimizer=("adam", b1, b2, eps) /
        ("sgd",) (single process only): the update is applied by the step's LAST launch, which also reduces the gradient
        partials, sums the KL / loss partials and accumulates the History metrics (csrc/dib_tail.h)."""
        inv = 1.0 / batch if inv_global_batch is None else inv_global_batch
        kind = LOSS_KINDS[loss_kind]
        fused_head = self._head_fused(kind)
        finish = _lib.TAIL_KL | (_lib.TAIL_LOSS_HEAD if fused_head else _lib.TAIL_LOSS) | (_lib.TAIL_METRICS if accumulate else 0)
        if fused_head:
            # encoder bank, then the integration network's whole share in one entry: hidden layers, output Dense(1) + loss and
            # their backward down to dL/du (one launch of 16-row tiles in the row-tile regime), hidden weight gradients
            self.encoder_forward(x, row_idx, row0, batch, seed, step, defer_sums=True)
            single = on_integration_grads_ready is None   # no bucket protocol: every weight gradient of the step in dib_backward
            check(self.lib.dib_integration_head_step(self.layout, kind, _ptr(y), y.stride(0), _ptr(row_idx), int(row0), batch,
                                                     float(inv), _lib.HEAD_DEFER_SUMS | (_lib.HEAD_DEFER_WGRAD if single else 0),
                                                     _ptr(self.params), _ptr(self.grads),
                                                     _ptr(self.workspace(batch)), self._stream()), "dib_integration_head_step")
            self.backward(row_idx, row0, batch, seed, step, inv, on_integration_grads_ready, hidden_only=True,
                          on_encoder_front_grads_ready=on_encoder_front_grads_ready, finish_flags=finish, optimizer=optimizer,
                          integration_done=True)
            return
        self.forward(x, row_idx, row0, batch, seed, step, defer_sums=True)
        self.loss(loss_kind, y, row_idx, row0, batch, inv, defer_sums=True)
        self.backward(row_idx, row0, batch, seed, step, inv, on_integration_grads_ready, hidden_only=False,
                      on_encoder_front_grads_ready=on_encoder_front_grads_ready, finish_flags=finish, optimizer=optimizer)

    def eval_step(self, x, y, row_idx, row0: int, batch: int, seed: int, step: int, loss_kind: str,
                  inv_global_batch: Optional[float] = None, metrics_acc: Optional[torch.Tensor] = None) -> None:
        """validation: noise stays ON and the KL term is included (reference train.py:263-265).  metrics_acc: where the step's
        History sums go (default: the training accumulator; fit passes `metrics_acc_val`)."""
        inv = 1.0 / batch if inv_global_batch is None else inv_global_batch
        kind = LOSS_KINDS[loss_kind]
        fused_head = self._head_fused(kind)
        if fused_head:
            self.encoder_forward(x, row_idx, row0, batch, seed, step, inference=True, defer_sums=True)
            check(self.lib.dib_integration_head_step(self.layout, kind, _ptr(y), y.stride(0), _ptr(row_idx), int(row0), batch,
                                                     float(inv), _lib.HEAD_DEFER_SUMS | _lib.HEAD_NO_GRAD, _ptr(self.params), None,
                                                     _ptr(self.workspace(batch)), self._stream()), "dib_integration_head_step")
        else:
            self.forward(x, row_idx, row0, batch, seed, step, inference=True, defer_sums=True)
            self.loss(loss_kind, y, row_idx, row0, batch, inv, defer_sums=True)
        self.step_tail(batch, -1, _lib.TAIL_KL | (_lib.TAIL_LOSS_HEAD if fused_head else _lib.TAIL_LOSS) | _lib.TAIL_METRICS, inv,
                       metrics_acc=metrics_acc)

    # ---- hipGraph capture of a whole step (launch-bound small-batch regime) -----------------------
    def enable_step_counter(self, value: int = 0) -> None:
        """Key the noise with a DEVICE-resident step counter (dib_layout_set_step_counter) so that a captured step
        can be replayed; from now on the by-value `step` arguments are ignored by the kernels."""
        if getattr(self, "step_dev", None) is None:
            self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
            check(self.lib.dib_layout_set_step_counter(self.layout, _ptr(self.step_dev)), "dib_layout_set_step_counter")
        self.set_step_counter(value)

    def set_step_counter(self, value: int) -> None:
        v = int(value) & 0xFFFFFFFF
        self.step_dev.fill_(v - (1 << 32) if v >= (1 << 31) else v)  # uint32 bit pattern in an int32 tensor

    def capture_step_graph(self, x, y, batch: int, loss_kind: str, inv_global_batch: float, seed: int, train: bool,
                           optimizer: str = "adam", opt_args=(0.9, 0.999, 1e-7)):
        """Capture fwd + loss [+ bwd + optimizer] + metric accumulation + step-counter bump for `batch` rows whose
        dataset indices are read from a fixed staging buffer.  Returns (graph, idx_stage): copy the batch's int32 row
        indices into idx_stage, then graph.replay().  Everything the step reads that changes between replays (beta,
        lr, Adam t, noise step, row indices) lives in device memory."""
        assert getattr(self, "step_dev", None) is not None, "call enable_step_counter() first"
        idx_stage = torch.zeros(batch, dtype=torch.int32, device=self.device)
        self.workspace(batch)
        self._ws_pinned.add(batch)  # the graph bakes this workspace's pointer in: never evict it
        # Eager warm-up with EXACTLY the launch sequence that is captured below (the fused output head, the hidden-only
        # integration passes, the optimizer and the counter bump included): every kernel's module load and every
        # hipFuncSetAttribute happens outside the capture.  All state the warm-up touches is restored afterwards.
        saved = [t.clone() for t in (self.metrics_acc, self.grads, self.params, self.adam_m, self.adam_v, self.t_dev,
                                     self.step_dev)]

        def body():
            if train:
                self.train_step(x, y, idx_stage, 0, batch, seed, 0, loss_kind, inv_global_batch,
                                optimizer=("adam", *opt_args) if optimizer == "adam" else ("sgd",))
            else:
                self.eval_step(x, y, idx_stage, 0, batch, seed, 0, loss_kind, inv_global_batch)
            self.step_dev.add_(1)

        def restore():
            for t, v in zip((self.metrics_acc, self.grads, self.params, self.adam_m, self.adam_v, self.t_dev, self.step_dev),
                            saved):
                t.copy_(v)

        body()
        restore()
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        restore()   # (capture does not execute, but keep the contract obvious: the caller's state is untouched)
        return graph, idx_stage

    def release_step_graph(self, batch: int) -> None:
        """The caller has dropped every graph captured for `batch` rows: its workspace may be evicted again (it re-enters
        the LRU as the most recent entry).  Without this, pinned workspaces accumulate for the lifetime of the engine."""
        self._ws_pinned.discard(batch)
        unpinned = [b for b in self._ws if b not in self._ws_pinned]
        while len(unpinned) > self._WS_KEEP:
            self._ws.pop(unpinned.pop(0))

    def adam_step(self, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0) -> None:
        check(self.lib.dib_adam_step(_ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                     self.n_params, _ptr(self.lr_dev), _ptr(self.t_dev), beta1, beta2, eps,
                                     float(grad_scale), self._stream()), "dib_adam_step")

    def sgd_step(self, grad_scale=1.0) -> None:
        check(self.lib.dib_sgd_step(_ptr(self.params), _ptr(self.grads), self.n_params, _ptr(self.lr_dev),
                                    float(grad_scale), self._stream()), "dib_sgd_step")

    def read_metrics(self, reset: bool = True) -> np.ndarray:
        m = self.metrics_acc.detach().cpu().numpy().astype(np.float64)
        if reset:
            self.metrics_acc.zero_()
        return m

    def read_metrics_pair(self, reset: bool = True):
        """(training sums, validation sums) with ONE device-to-host copy (= one synchronisation) and one reset."""
        both = self._metrics2.detach().cpu().numpy().astype(np.float64)
        if reset:
            self._metrics2.zero_()
        n, st = self.F + 3, self._metrics_stride
        return both[:n].copy(), both[st: st + n].copy()

    def profile_enable(self, on: bool) -> None:
        check(self.lib.dib_profile_enable(1 if on else 0), "dib_profile_enable")

    def profile_summary(self) -> dict:
        """{kernel symbol: (total ms, launches)} since profile_enable(True); synchronises."""
        return profile_summary(self.lib)


    # ---- views / helpers ---------------------------------------------------------------------
    def pred(self, batch: int) -> torch.Tensor:
        return self.ws_view(batch, _lib.WS_PRED, batch * self.out_dim).view(batch, self.out_dim)

    def u(self, batch: int) -> torch.Tensor:
        return self.ws_view(batch, _lib.WS_U, batch * self.F * self.E).view(batch, self.F * self.E)

    def enc_out(self, batch: int) -> torch.Tensor:
        # stored feature-major [F][B][2E] on the device; presented as [B, F, 2E]
        return self.ws_view(batch, _lib.WS_ENC_OUT, batch * self.F * 2 * self.E).view(self.F, batch, 2 * self.E) \
            .permute(1, 0, 2)

    def enc_h(self, batch: int, layer: int) -> torch.Tensor:
        """post-activation output of encoder hidden layer `layer` as stashed by the training forward: [F, B, units]."""
        w = self.enc_units[layer]
        return self.ws_view(batch, _lib.WS_ENC_H0 + layer, batch * self.F * w).view(self.F, batch, w)

    def int_h(self, batch: int, layer: int) -> torch.Tensor:
        """post-activation output of integration hidden layer `layer`: [B, units]."""
        w = self.int_units[layer]
        return self.ws_view(batch, _lib.WS_INT_H0 + layer, batch * w).view(batch, w)

    def g_u(self, batch: int) -> torch.Tensor:
        return self.ws_view(batch, _lib.WS_G_U, batch * self.F * self.E).view(batch, self.F * self.E)

    def g_pred(self, batch: int) -> torch.Tensor:
        """dL/d(model output) [batch, out_dim]: what backward_from_pred_grad consumes (a view of the step workspace)."""
        return self.ws_view(batch, _lib.WS_G_PRED, batch * self.out_dim).view(batch, self.out_dim)

    def step_out(self, batch: int) -> torch.Tensor:
        return self.ws_view(batch, _lib.WS_STEP_OUT, self.F + 3)

    def encode_feature(self, f: int, x_f) -> torch.Tensor:
        """model.feature_encoders[f](x_f): deterministic [N, 2E] (reference models.py:183, visualization.py:31)."""
        xf = self.to_device(x_f).reshape(-1, self.dims[f])
        n = xf.shape[0]
        out = torch.empty((n, 2 * self.E), dtype=torch.float32, device=self.device)
        ws = self.scratch(int(self.lib.dib_workspace_bytes(self.layout, n)))
        check(self.lib.dib_encode_deterministic(self.layout, int(f), _ptr(xf), n, _ptr(self.params), _ptr(out),
                                                _ptr(ws), self._stream()), "dib_encode_deterministic")
        return out

    def bhattacharyya(self, mu1, lv1, mu2, lv2) -> torch.Tensor:
        mu1, lv1, mu2, lv2 = [self.to_device(t) for t in (mu1, lv1, mu2, lv2)]
        n, d = mu1.shape
        m = mu2.shape[0]
        out = torch.empty((n, m), dtype=torch.float32, device=self.device)
        check(self.lib.dib_bhattacharyya(_ptr(mu1), _ptr(lv1), n, _ptr(mu2), _ptr(lv2), m, d, _ptr(out),
                                         self._stream()), "dib_bhattacharyya")
        return out

    def infonce(self, emb_x: torch.Tensor, emb_y: torch.Tensor, similarity: str = "l2", temperature: float = 1.0,
                want_grads: bool = True, out_gx: Optional[torch.Tensor] = None, out_gy: Optional[torch.Tensor] = None,
                loss_out: Optional[torch.Tensor] = None):
        """Symmetric InfoNCE (reference train.py:201-214, utils.py:131-175) -> (loss [1] device tensor, g_x, g_y).
        out_gx / out_gy: contiguous [B, D] tensors to receive the gradients (e.g. the model's dL/d(output) workspace view and
        the Y encoder's gradient buffer: the training loop then needs no copies).  loss_out: a 1-element float32 device tensor
        to receive the loss (a slot of the caller's per-epoch buffer) instead of a fresh allocation."""
        emb_x, emb_y = emb_x.contiguous(), emb_y.contiguous()
        b, d = emb_x.shape
        ws = self._infonce_ws.get(b)
        if ws is None:
            if len(self._infonce_ws) >= 4:
                self._infonce_ws.pop(next(iter(self._infonce_ws)))
            ws = self._infonce_ws[b] = torch.empty(int(self.lib.dib_infonce_workspace_bytes(b)) // 4, dtype=torch.float32,
                                                   device=self.device)
        loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=self.device)
        assert loss.numel() == 1 and loss.dtype == torch.float32 and loss.device == self.device
        gx = gy = None
        if want_grads:
            for o in (out_gx, out_gy):
                assert o is None or (o.is_contiguous() and tuple(o.shape) == (b, d) and o.dtype == torch.float32)
            gx = out_gx if out_gx is not None else torch.empty_like(emb_x)
            gy = out_gy if out_gy is not None else torch.empty_like(emb_y)
        check(self.lib.dib_infonce_fwd_bwd(_ptr(emb_x), _ptr(emb_y), b, d, _lib.SIMILARITIES[similarity], float(temperature),
                                           _ptr(gx), _ptr(gy), _ptr(loss), _ptr(ws), self._stream()), "dib_infonce_fwd_bwd")
        return loss, gx, gy

    def backward_from_pred_grad(self, g_pred: torch.Tensor, row_idx, row0: int, batch: int, seed: int, step: int,
                                inv_global_batch: Optional[float] = None, finish_flags: int = 0, optimizer=None,
                                companion=None, metrics_acc: Optional[torch.Tensor] = None) -> None:
        """Backward of the model given dL/d(model output) from a custom loss (reference train.py:216-219): the
        beta*KL term (models.py:118) is added inside the encoder-bank backward.  Gradients land in self.grads.
        finish_flags / optimizer: what the backward's last launch also does (step_tail: e.g. TAIL_KL | TAIL_METRICS after a
        forward(defer_sums=True), and the optimizer update of a single-process loop); TAIL_METRICS accumulates into `metrics_acc`
        (a custom loop passes its OWN accumulator: the default is the one model.fit's History is read from)."""
        inv = 1.0 / batch if inv_global_batch is None else inv_global_batch
        dst = self.g_pred(batch)
        if not (g_pred.data_ptr() == dst.data_ptr() and g_pred.shape == dst.shape and g_pred.stride() == dst.stride()):
            dst.copy_(g_pred)   # (a custom loss may have written its gradient straight into the view)
        self.backward(row_idx, row0, batch, seed, step, inv, finish_flags=finish_flags, optimizer=optimizer, companion=companion,
                      metrics_acc=metrics_acc)

    def mi_sandwich_bounds(self, enc_out: torch.Tensor, seed: int, step: int, feature: int):
        """(InfoNCE lower, leave-one-out upper) in nats for one batch enc_out [N, 2E] (reference utils.py:36-62)."""
        enc_out = self.to_device(enc_out)
        n, e2 = enc_out.shape
        ws = torch.empty(int(self.lib.dib_mi_workspace_bytes(n, e2 // 2)) // 8, dtype=torch.float64, device=self.device)
        rows = torch.empty((2, n), dtype=torch.float64, device=self.device)
        check(self.lib.dib_mi_sandwich_rows(_ptr(enc_out), n, e2 // 2, int(seed), int(step) & 0xFFFFFFFF, int(feature),
                                            _ptr(rows[0]), _ptr(rows[1]), _ptr(ws), self._stream()),
              "dib_mi_sandwich_rows")
        m = rows.mean(dim=1).cpu().numpy()
        return float(m[0]), float(m[1])

    def eps(self, row_idx, row0: int, batch: int, seed: int, step: int) -> torch.Tensor:
        out = torch.empty((batch, self.F, self.E), dtype=torch.float32, device=self.device)
        check(self.lib.dib_philox_normal_fill(_ptr(out), _ptr(row_idx), int(row0), batch, self.F, self.E, int(seed),
                                              int(step) & 0xFFFFFFFF, self._stream()), "dib_philox_normal_fill")
        return out


PROFILE_CATEGORIES = 17  # include/dib_hip.h: DIB_PROFILE_CATEGORIES


def profile_summary(lib) -> dict:
    """{kernel symbol: (total ms, launches)} of the library's live HIP-event timing since dib_profile_enable(1)."""
    ms = (ctypes.c_double * PROFILE_CATEGORIES)()
    cnt = (c_int * PROFILE_CATEGORIES)()
    check(lib.dib_profile_summary(ms, cnt), "dib_profile_summary")
    bk = lambda mode, ni, nj: 64 if (ni, nj) == (2, 2) or (mode, ni, nj) == (2, 1, 2) else 32  # csrc/dib_api.hip launch_gemm_t
    names = [f"dib_gemm_kernel<{mode}, {ni}, {nj}, {bk(mode, ni, nj)}>"
             for mode in (0, 1, 2) for ni in (1, 2) for nj in (1, 2)]
    names += ["dib_fused_encoder_fwd_kernel", "dib_fused_encoder_bwd_kernel", "other", "dib_attn_fwd_kernel", "dib_attn_bwd_kernel"]
    return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(names) if cnt[i]}
