"""DenseStack: a plain MLP ([PositionalEncoding] -> Dense(units, act)* -> Dense(out)) on the same hand-written
gfx950 GEMM kernels, used for the InfoNCE path's output encoder (reference train.py:184-192: `output_encoder`).
Forward, backward and Keras-Adam all run on the device.

Per batch size a plan holds the activations, the gradient buffers and the descriptor tables of `dib_gemm_grouped`
(include/dib_st.h), built and uploaded once: a step is 3 launches per layer with no per-call descriptor traffic.  Weight
gradients contract over the batch in <= 32 row slabs (one partial parameter buffer each) summed in a fixed order by
`dib_reduce_splits` - the first version ran every weight gradient as ONE workgroup per 128 x 128 output tile walking the
whole batch: 265 us per layer at the chaos notebook's batch of 2048 (profiles/r03m_*)."""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from ._gemm_plan import _Gemm, _d, _ptr
from ._lib import ACTIVATIONS, SYNC_WORDS as _SYNC_WORDS, check


class _MlpDesc(ctypes.Structure):
    """include/dib_hip.h dib_mlp_desc: where the layers of a plain MLP sit in its flat parameter buffer"""
    _fields_ = [("w_off", ctypes.c_int64 * 4), ("b_off", ctypes.c_int64 * 4), ("n_hidden", ctypes.c_int32),
                ("width", ctypes.c_int32 * 4), ("in_dim", ctypes.c_int32), ("n_freq", ctypes.c_int32), ("act", ctypes.c_int32)]


assert ctypes.sizeof(_MlpDesc) == 96


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
        self._plans: Dict[int, dict] = {}
        self._last: Optional[dict] = None
        # batches <= 2048 rows: the whole layer chain of 16 rows in one workgroup (dib_mlp_small_fwd / _bwd, csrc/dib_small.h)
        self._desc = None
        if 2 <= len(self.dims) <= 4:
            d = _MlpDesc()
            for l in range(len(self.dims)):
                d.w_off[l], d.b_off[l], d.width[l] = self.w_off[l], self.b_off[l], self.dims[l][1]
            d.n_hidden, d.in_dim, d.n_freq, d.act = len(self.dims) - 1, self.input_dim, max(self.n_freq, 1), self.act
            self._desc = d

    # views
    def kernel(self, l):
        i, o = self.dims[l]
        return self.params[self.w_off[l]: self.w_off[l] + i * o].view(i, o)

    def bias(self, l):
        return self.params[self.b_off[l]: self.b_off[l] + self.dims[l][1]]

    def _plan(self, n: int) -> dict:
        pl = self._plans.get(n)
        if pl is not None:
            return pl
        L = len(self.dims)
        off, o = {}, 0

        def take(name, cnt):
            nonlocal o
            off[name] = o
            o = (o + int(cnt) + 3) // 4 * 4

        take("y", n * self.input_dim)
        take("a0", n * self.dims[0][0])                      # (positionally encoded) input
        for l, (_, wo) in enumerate(self.dims):
            take(f"a{l + 1}", n * wo)                        # post-activation output of layer l
            take(f"g{l + 1}", n * wo)                        # dL/d(pre-activation of layer l) (= dL/d output for the last)
        ws = torch.zeros(o, dtype=torch.float32, device=self.device)
        # weight gradients: batch slabs of >= 64 rows (multiple of 32), <= 32 of them, fixed-order reduce
        nsplit = max(1, min(32, n // 64))
        rps = ((n + nsplit - 1) // nsplit + 31) // 32 * 32
        nsplit = (n + rps - 1) // rps
        slabs = torch.zeros(nsplit * self.n_params, dtype=torch.float32, device=self.device) if nsplit > 1 else None
        gt = slabs if nsplit > 1 else self.grads
        g = {}
        for l, (wi, wo) in enumerate(self.dims):
            act = self.act if l < L - 1 else 0
            g[f"fwd{l}"] = _Gemm(0, [_d(off[f"a{l}"], wi, self.w_off[l], wo, off[f"a{l + 1}"], wo, n, wo, wi,
                                        bias_off=self.b_off[l])], ws, self.params, ws, bias=self.params, act=act)
            if l > 0:   # dL/d(pre-activation of layer l-1) = (g_l @ W_l^T) * act'(a_l)
                g[f"dgrad{l}"] = _Gemm(1, [_d(off[f"g{l + 1}"], wo, self.w_off[l], wo, off[f"g{l}"], wi, n, wi, wo,
                                              aux_off=off[f"a{l}"], ldaux=wi)], ws, self.params, ws, aux=ws, act=self.act)
        # every layer's weight gradient in ONE grouped launch (independent products; the dgrad chain runs first)
        g["wgrad_all"] = _Gemm(2, [_d(off[f"a{l}"], wi, off[f"g{l + 1}"], wo, self.w_off[l], wo, wi, wo, n, bias_off=self.b_off[l])
                                  for l, (wi, wo) in enumerate(self.dims)], ws, ws, gt, bias_out=gt, nsplit=nsplit,
                               rows_per_split=rps, split_stride=self.n_params)
        for gg in g.values():
            gg.upload(self.device)
        if len(self._plans) >= 4:                             # train batch, validation batch, their tails
            self._plans.pop(next(iter(self._plans)))
        pl = self._plans[n] = dict(n=n, ws=ws, off=off, g=g, nsplit=nsplit, slabs=slabs, small=False)
        if self._desc is not None:
            nh = L - 1
            ptrs = lambda pre: (ctypes.c_void_p * 3)(*[_ptr(ws, off[f"{pre}{l + 1}"]).value if l < nh else None for l in range(3)])
            pl.update(h_ptrs=ptrs("a"), g_ptrs=ptrs("g"))
        return pl

    def _view(self, pl, name, rows, cols):
        o = pl["off"][name]
        return pl["ws"][o: o + rows * cols].view(rows, cols)

    def forward(self, y: torch.Tensor, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[n, input_dim] -> [n, output_dim].  The result is a view into the plan's workspace: valid until the next forward
        with the same batch size.  rows (int32 / int64 device tensor): encode y[rows] - gathered straight into the workspace."""
        y = y.to(device=self.device, dtype=torch.float32)
        if rows is not None:
            if rows.dim() != 1:
                raise ValueError("rows must be a 1-D index tensor")
            # the kernels read contiguous int32 indices; int64 / strided / host index tensors are converted, not refused
            rows = rows.to(device=y.device, dtype=torch.int32).contiguous()
            if y.stride(1) != 1:
                y = y.contiguous()
        n = y.shape[0] if rows is None else int(rows.shape[0])
        pl = self._plan(n)
        st = self.eng._stream()
        L = len(self.dims)
        # (asked per call: "small_batch" is a run-time tuning key; the backward follows the forward's choice)
        pl["small"] = bool(self._desc is not None and y.dim() == 2 and y.stride(1) == 1
                           and self.lib.dib_mlp_small_supported(ctypes.byref(self._desc), n))
        if pl["small"]:   # gather + positional encoding + every layer: one launch
            check(self.lib.dib_mlp_small_fwd(ctypes.byref(self._desc), _ptr(self.params), _ptr(y), y.stride(0),
                                             _ptr(rows) if rows is not None else None, n, _ptr(pl["ws"], pl["off"]["a0"]),
                                             pl["h_ptrs"], _ptr(pl["ws"], pl["off"][f"a{L}"]), st), "dib_mlp_small_fwd")
            self._last = pl
            return self._view(pl, f"a{L}", n, self.dims[-1][1])
        if rows is not None:   # gather (+ positional encoding) in one launch, straight into the first layer's operand
            check(self.lib.dib_positional_encoding_rows(_ptr(y), y.stride(0), _ptr(rows), n, self.input_dim, self.n_freq,
                                                        _ptr(pl["ws"], pl["off"]["a0"]), st), "dib_positional_encoding_rows")
        elif self.n_freq > 1:
            self._view(pl, "y", n, self.input_dim).copy_(y)
            check(self.lib.dib_positional_encoding(_ptr(pl["ws"], pl["off"]["y"]), self.input_dim, n, self.input_dim, self.n_freq,
                                                   _ptr(pl["ws"], pl["off"]["a0"]), st), "dib_positional_encoding")
        else:
            self._view(pl, "a0", n, self.input_dim).copy_(y)
        for l in range(len(self.dims)):
            pl["g"][f"fwd{l}"].run(self.lib, st)
        self._last = pl
        return self._view(pl, f"a{len(self.dims)}", n, self.dims[-1][1])

    def companion_forward(self, y: torch.Tensor, rows: Optional[torch.Tensor] = None):
        """The argument tuple of `dib_integration_fwd_and_mlp_fwd` for forward(y, rows) - the caller hands it to the X model's
        forward (HipEngine.forward(companion=...)), which runs both networks in one grid; the result is then
        `companion_output()`.  None when this batch does not take the row-tile kernels (the caller calls forward() itself)."""
        y = y.to(device=self.device, dtype=torch.float32)
        n = y.shape[0] if rows is None else int(rows.shape[0])
        if rows is not None and (rows.dtype != torch.int32 or rows.device != y.device or rows.dim() != 1 or not rows.is_contiguous()):
            raise ValueError("rows must be a contiguous 1-D int32 tensor on the stack's device")
        if self._desc is None or y.dim() != 2 or y.stride(1) != 1 or n < 1 \
                or not self.lib.dib_mlp_small_supported(ctypes.byref(self._desc), n):
            return None
        pl = self._plan(n)
        pl["small"] = True
        self._last = pl
        self._keep = (y, rows)   # alive until the launch that reads them has been issued
        return (ctypes.byref(self._desc), _ptr(self.params), _ptr(y), y.stride(0), _ptr(rows) if rows is not None else None, n,
                _ptr(pl["ws"], pl["off"]["a0"]), pl["h_ptrs"], _ptr(pl["ws"], pl["off"][f"a{len(self.dims)}"]))

    def companion_output(self) -> torch.Tensor:
        pl = self._last
        return self._view(pl, f"a{len(self.dims)}", pl["n"], self.dims[-1][1])

    def companion_backward(self, g_out: torch.Tensor):
        """The argument tuple of `dib_backward_and_mlp_bwd` for the dgrad chain of backward(g_out) (HipEngine.backward(companion=...));
        follow with backward(g_out, dgrad_done=True) for the weight gradients.  None if the last forward did not take the
        row-tile kernels."""
        pl = self._last
        assert pl is not None and g_out.shape[0] == pl["n"], "backward follows a forward with the same batch"
        if not pl["small"]:
            return None
        n, L = pl["n"], len(self.dims)
        dst = self._view(pl, f"g{L}", n, self.dims[-1][1])
        if not (g_out.data_ptr() == dst.data_ptr() and g_out.shape == dst.shape and g_out.stride() == dst.stride()):
            dst.copy_(g_out)
        return (ctypes.byref(self._desc), _ptr(self.params), _ptr(pl["ws"], pl["off"][f"g{L}"]), pl["h_ptrs"], pl["g_ptrs"], n)

    def output_grad_buffer(self) -> torch.Tensor:
        """[n, output_dim] view that backward() reads d loss / d output from (of the last forward's plan): a loss kernel can
        write its gradient there directly."""
        pl = self._last
        assert pl is not None
        return self._view(pl, f"g{len(self.dims)}", pl["n"], self.dims[-1][1])

    def backward(self, g_out: torch.Tensor, reduce: bool = True, dgrad_done: bool = False) -> None:
        """grads <- d loss / d params given d loss / d output of the last forward (overwrites self.grads).  reduce=False: the
        batch-slab partials are left for adam_step(fused_reduce=True), which sums them in the optimizer's own launch.
        dgrad_done=True: the dgrad chain already ran as the companion of the X model's backward (companion_backward)."""
        pl = self._last
        assert pl is not None and g_out.shape[0] == pl["n"], "backward follows a forward with the same batch"
        n, L, st = pl["n"], len(self.dims), self.eng._stream()
        dst = self._view(pl, f"g{L}", n, self.dims[-1][1])
        if not (g_out.data_ptr() == dst.data_ptr() and g_out.shape == dst.shape and g_out.stride() == dst.stride()):
            dst.copy_(g_out)
        if pl["nsplit"] == 1:
            self.grads.zero_()
        if dgrad_done:
            assert pl["small"]
        elif pl["small"]:                                # the whole dgrad chain: one launch
            check(self.lib.dib_mlp_small_bwd(ctypes.byref(self._desc), _ptr(self.params), _ptr(pl["ws"], pl["off"][f"g{L}"]),
                                             pl["h_ptrs"], pl["g_ptrs"], n, st), "dib_mlp_small_bwd")
        else:
            for l in reversed(range(1, L)):
                pl["g"][f"dgrad{l}"].run(self.lib, st)   # dL/d(pre-activation of layer l-1)
        pl["g"]["wgrad_all"].run(self.lib, st)       # dW_l[i,o] = a_l^T @ g_{l+1}, bias gradients = column sums of g_{l+1}, all l
        self._unreduced = pl if (pl["nsplit"] > 1 and not reduce) else None
        if pl["nsplit"] > 1 and reduce:
            check(self.lib.dib_reduce_splits(_ptr(pl["slabs"]), self.n_params, pl["nsplit"], self.n_params, _ptr(self.grads), st),
                  "dib_reduce_splits")

    def set_lr(self, lr: float) -> None:
        self.lr_dev.fill_(float(lr))

    def adam_step(self, lr: Optional[float] = None, beta1=0.9, beta2=0.999, eps=1e-7, fused_reduce: bool = False) -> None:
        """lr None: the learning rate last set (set_lr) - a loop with a constant rate sets it once.  fused_reduce=True: one
        launch sums the slabs a backward(reduce=False) left, applies Keras-Adam and bumps the step count
        (dib_reduce_adam_step) instead of reduce + Adam + bump."""
        if lr is not None:
            self.lr_dev.fill_(float(lr))
        if fused_reduce:
            pl = getattr(self, "_unreduced", None)
            if getattr(self, "_sync", None) is None:
                self._sync = torch.zeros(_SYNC_WORDS, dtype=torch.int32, device=self.device)
            check(self.lib.dib_reduce_adam_step(_ptr(pl["slabs"]) if pl is not None else None, pl["nsplit"] if pl is not None else 0,
                                                self.n_params, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m),
                                                _ptr(self.adam_v), self.n_params, _ptr(self.lr_dev), _ptr(self.t_dev), beta1,
                                                beta2, eps, 1.0, _ptr(self._sync), self.eng._stream()), "dib_reduce_adam_step")
            self._unreduced = None
            return
        assert getattr(self, "_unreduced", None) is None, "backward(reduce=False) must be followed by adam_step(fused_reduce=True)"
        check(self.lib.dib_adam_step(_ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                     self.n_params, _ptr(self.lr_dev), _ptr(self.t_dev), beta1, beta2, eps, 1.0,
                                     self.eng._stream()), "dib_adam_step")
