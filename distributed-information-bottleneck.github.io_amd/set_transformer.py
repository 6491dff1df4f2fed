"""Per-particle Distributed-IB set transformer on the MI355X-native path (BASELINE config 5).

Host-side mirror of the reference notebook
    complex_systems/InfoDecomp_Amorphous_plasticity_per_particle_measurements_and_set_transformer.ipynb, code cell 8:
`particle_encoder` (PositionalEncoding -> Dense(128, LeakyReLU(0.1)) x2 -> Dense(2*32), shared by all particles), the
`train_step` bottleneck (logvar - 3, reparameterised sample, KL summed over (particle, dim) and averaged over the batch),
`set_transformer` (6 x [MultiHeadAttention(12, 128)(x, x, x) -> Add -> LayerNormalization -> Dense(128, relu),
Dense(32, relu) -> Add -> LayerNormalization], tf.reduce_mean over the particle axis, Dense(256, LeakyReLU(0.1)), Dense(1)),
BCE-from-logits + beta * KL, Keras Adam, the linear learning-rate warm-up and the per-STEP log ramp of beta of the
notebook's training loop.  Same names and argument meaning as the notebook's variables.

Every FLOP runs in libdib_hip.so through the C ABI of include/dib_st.h: all matrix products (encoder, q/k/v/output
projections, the per-(neighbourhood, head) Q K^T, P V and their four backward products, feed-forward, head, every weight
gradient) on the grouped fp32-MFMA GEMM (`dib_gemm_grouped`, 12 x batch groups per launch for the attention products),
softmax / Add+LayerNorm / mean-pool / reparameterisation+KL / loss / Adam as HBM-bound row kernels.  PyTorch only owns the
device memory.  There is no CPU fallback.

Parameters live in one flat fp32 buffer in Keras variable-creation order (`param_shapes`, the order of
`particle_encoder.trainable_variables + set_transformer.trainable_variables`), every block 16-byte aligned.
"""
from __future__ import annotations

import ctypes
import math
import os
from ctypes import c_void_p
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._gemm_plan import DESC, _Gemm, _SkinnyKGemm, _d, _ptr, _ptr8
from ._lib import check

ACT_NONE, ACT_RELU, ACT_LEAKY01 = 0, 1, _lib.ACT_LEAKY_RELU_01


class _BlockDesc(ctypes.Structure):
    """include/dib_st.h dib_st_block_desc: element offsets of one attention block's token-wise chain in the flat parameter buffer"""
    _fields_ = [("o_w", ctypes.c_int64), ("o_b", ctypes.c_int64), ("ln1_g", ctypes.c_int64), ("ln1_b", ctypes.c_int64),
                ("ln2_g", ctypes.c_int64), ("ln2_b", ctypes.c_int64), ("ff_w", ctypes.c_int64 * 3), ("ff_b", ctypes.c_int64 * 3),
                ("n_ff", ctypes.c_int32), ("ff_width", ctypes.c_int32 * 3), ("D", ctypes.c_int32), ("HK", ctypes.c_int32),
                ("eps", ctypes.c_float), ("act", ctypes.c_int32)]


assert ctypes.sizeof(_BlockDesc) == 128
LOSS_BCE_LOGITS = 0
# Test switch: take train_step's collective branch even on a ONE-rank process group, so that the RCCL calls themselves run
# on the single GPU the test box has (tests/_dp_gpu_st_worker.py).  A 1-rank sum all-reduce is the identity.
_FORCE_DP_BRANCH = False


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


def convert_to_per_particle_feature_set(particle_positions, types, number_particles_to_use=60):
    """Notebook cell 6: per-particle features (x, x^2, r, log r, log x^2, unit vector, one-hot type), nearest particles
    first.  Host-side data preparation (NumPy), done once per dataset."""
    pos = np.asarray(particle_positions, dtype=np.float32)
    types = np.asarray(types).astype(np.int32)
    one_hot = np.eye(2, dtype=np.float32)[types - 1]
    radii = np.sqrt(np.sum(np.square(pos), -1, keepdims=True) + np.float32(1e-10)).astype(np.float32)
    unit = pos / radii
    feats = np.concatenate([pos, pos ** 2, radii, np.log(radii + np.float32(1e-3)), np.log(pos ** 2 + np.float32(1e-3)),
                            unit, one_hot], -1).astype(np.float32)
    if number_particles_to_use > 0:
        order = np.argsort(np.squeeze(radii, -1), kind="stable")
        feats = feats[order][:number_particles_to_use]
    return feats


class _SplitKGemm:
    """A skinny product C[M, N] = A[M, K] @ W (N = the model width, 32; K = heads * key_dim = 1536) with FEW row tiles: the
    output has ceil(M / 64) workgroups' worth of tiles and each would walk all of K with one tile of prefetch - at the
    notebook's size (1600 tokens) 25 workgroups x 48 dependent k-tiles = 84 us for 0.16 GFLOP, the two slowest launches of
    the whole step.  Here the contraction is cut into `ksplit` chunks that run as extra GROUPS of the same grouped launch
    (one partial slab each), summed in a fixed order by dib_reduce_splits: deterministic, no new kernel."""

    def __init__(self, gemm: "_Gemm", partial, partial_off, n, nslabs, out, out_off, mode: str = "store"):
        """mode: "store" out = sum of the slabs; "add" out += sum (a residual branch's gradient joins the one already there:
        one launch instead of reduce + add); "defer" no reduce here - the consumer sums the slabs itself
        (dib_add_layernorm_fwd's b_slabs)."""
        self.gemm, self.partial, self.partial_off, self.n, self.nslabs, self.out, self.out_off, self.mode = \
            gemm, partial, partial_off, n, nslabs, out, out_off, mode

    def upload(self, device):
        self.gemm.upload(device)

    def run(self, lib, stream):
        self.gemm.run(lib, stream)
        if self.mode == "defer":
            return
        fn = lib.dib_reduce_splits_add if self.mode == "add" else lib.dib_reduce_splits
        check(fn(_ptr(self.partial, self.partial_off), self.n, self.nslabs, self.n, _ptr(self.out, self.out_off), stream),
              "dib_reduce_splits")


class SetTransformerDIB:
    """`particle_encoder` + `set_transformer` + `train_step` of the notebook as one device-resident object."""

    def __init__(self, particle_feature_dimensions: int = 12, number_positional_encoding_frequencies: int = 5,
                 particle_encoder_arch_spec: Sequence[int] = (128, 128), bottleneck_dimension: int = 32, key_dim: int = 128,
                 number_heads_per_mha: int = 12, number_attention_blocks: int = 6,
                 ff_arch_per_block: Sequence[int] = (128, 32), final_processing_arch: Sequence[int] = (256,),
                 output_dimensionality: int = 1, logvar_initialization: float = -3.0, layer_norm_epsilon: float = 1e-3,
                 *, init_seed: int = 0, noise_seed: int = 0, device: Optional[str] = None, attention: str = "auto",
                 attention_score_stash_bytes: int = 64 << 30, use_graphs: Optional[bool] = None):
        """use_graphs: replay the whole training step (copy-in, ~190 launches, Adam, noise-step bump) as one captured hipGraph
        per (batch, particles) shape - the notebook's own configuration, 32 neighbourhoods x 50 particles, is bound by launch
        and dependency latency, not by arithmetic.  Default: the DIB_ENABLE_GRAPHS=1 opt-in shared with DistributedIBNet.fit.
        Single-process only (the data-parallel step has collectives between its launches).
        attention_score_stash_bytes: flash attention keeps the raw [P, P] score tiles of every block for the backward
        (4 tile products per tile pair instead of 5, include/dib_st.h) when all blocks' tiles of a batch shape fit this budget
        (4 x 4096 particles: 19.3 GB of the 288); above it - or with 0 - the backward recomputes them.
        attention: "flash" = dib_attention_fwd/bwd (probabilities never in HBM; key_dim must be 128), "gemm" = the products as
        grouped GEMMs with the [P, P] probabilities stashed in HBM (any key_dim), "auto" (per batch shape, key_dim == 128):
        flash - since the round-2 rewrite of the attention kernels it is the faster path at every measured shape (ms/step flash
        vs gemm: 32 x 50: 3.08 / 4.14, 4 x 512: 4.29 / 4.88, 2 x 2048: 15.0 / 15.7, 4 x 4096: 77.2 / 100.5;
        profiles/r02am_set_transformer_bench.txt, r02final2_set_transformer_bench.txt) and it needs no [P, P] stash in HBM; key_dim != 128: gemm."""
        self._acquire_device(device)
        self.particle_feature_dimensions = int(particle_feature_dimensions)
        self.number_positional_encoding_frequencies = int(number_positional_encoding_frequencies)
        self.particle_encoder_arch_spec = [int(u) for u in particle_encoder_arch_spec]
        self.bottleneck_dimension = int(bottleneck_dimension)
        self.key_dim, self.number_heads_per_mha = int(key_dim), int(number_heads_per_mha)
        self.number_attention_blocks = int(number_attention_blocks)
        self.ff_arch_per_block = [int(u) for u in ff_arch_per_block]
        assert self.ff_arch_per_block[-1] == self.bottleneck_dimension, "the feed-forward block must return to the model width"
        self.final_processing_arch = [int(u) for u in final_processing_arch]
        self.output_dimensionality = int(output_dimensionality)
        self.logvar_initialization = float(logvar_initialization)
        self.layer_norm_epsilon = float(layer_norm_epsilon)
        self.noise_seed = int(noise_seed)
        if attention not in ("auto", "flash", "gemm"):
            raise ValueError(f"attention={attention!r}")
        if attention == "flash" and self.key_dim != 128:
            raise ValueError("attention='flash' needs key_dim == 128 (the notebook's value)")
        self.attention = attention
        self.attention_score_stash_bytes = int(attention_score_stash_bytes)
        # token count from which the q / k / v projections and the context gradient run as streaming skinny-K launches
        # (same-box A/B, profiles/r03ad_*: 400 tokens +1.5 %, 1024 -1 %, 1600 -1.7 %, 2048 -5.7 %, 16 384 -1 % of the step)
        self.skinny_k_min_tokens = int(os.environ.get("DIB_SKINNY_K_MIN_TOKENS", "1024"))
        self.attention_impl = "flash" if (attention == "flash" or (attention == "auto" and self.key_dim == 128)) else "gemm"
        assert self.bottleneck_dimension <= 256 and self.bottleneck_dimension % 4 == 0
        # ---- flat parameter layout (Keras creation order) ----
        self.shapes = self.param_shapes()
        self.offsets: Dict[str, int] = {}
        o = 0
        for name, shp in self.shapes.items():
            self.offsets[name] = o
            o = _align4(o + int(np.prod(shp)))
        self.n_alloc = o
        self.n_params = int(sum(int(np.prod(s)) for s in self.shapes.values()))
        fdt = self.state_dtype
        z = lambda n, dt=fdt: torch.zeros(n, dtype=dt, device=self.device)
        self.params, self.grads, self.adam_m, self.adam_v = z(o), z(o), z(o), z(o)
        self.beta_dev = torch.ones(1, dtype=fdt, device=self.device)
        self.lr_dev = torch.full((1,), 1e-4, dtype=fdt, device=self.device)
        self.t_dev = z(1, torch.int64)
        self.set_params(self.init_params(init_seed))
        self._plans: Dict[Tuple[int, int], dict] = {}
        self.max_step_plans = 4
        self.max_graphs = 4                # captured step graphs kept (each pins its plan, workspace and score stash)
        self.use_graphs = (os.environ.get("DIB_ENABLE_GRAPHS", "0") == "1") if use_graphs is None else bool(use_graphs)
        self._graphs: Dict[Tuple[int, int], dict] = {}
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.device)   # noise step of graph replays (uint32 bits)
        self._step = 0
        self.last = {}

    state_dtype = torch.float32   # parameters, gradients and Adam moments live on the device in the reference's own precision

    def _acquire_device(self, device: Optional[str]) -> None:
        """The GPU and the HIP library: there is no CPU path.  (The device steps - forward, loss_and_backward, _loss_only,
        adam_step - and this method are what tests/_oracle_set_transformer.py overrides in a subclass to run the HOST logic of
        train_step / fit on the float64 checker.)"""
        if not torch.cuda.is_available():
            raise RuntimeError("SetTransformerDIB needs an AMD GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.lib = _lib.load_library()
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")

    # ---- parameters ---------------------------------------------------------------------------------------------
    def param_shapes(self) -> Dict[str, tuple]:
        """Keras creation order.  Kernels [in, out]; MultiHeadAttention kernels [dim, heads, key_dim] / [heads, key_dim, dim]."""
        s: Dict[str, tuple] = {}
        d_in = self.particle_feature_dimensions * self.number_positional_encoding_frequencies
        for l, u in enumerate(self.particle_encoder_arch_spec + [2 * self.bottleneck_dimension]):
            s[f"enc{l}_w"], s[f"enc{l}_b"] = (d_in, u), (u,)
            d_in = u
        D, H, K = self.bottleneck_dimension, self.number_heads_per_mha, self.key_dim
        for b in range(self.number_attention_blocks):
            for nm in ("q", "k", "v"):
                s[f"blk{b}_{nm}_w"], s[f"blk{b}_{nm}_b"] = (D, H, K), (H, K)
            s[f"blk{b}_o_w"], s[f"blk{b}_o_b"] = (H, K, D), (D,)
            s[f"blk{b}_ln1_g"], s[f"blk{b}_ln1_b"] = (D,), (D,)
            d = D
            for l, u in enumerate(self.ff_arch_per_block):
                s[f"blk{b}_ff{l}_w"], s[f"blk{b}_ff{l}_b"] = (d, u), (u,)
                d = u
            s[f"blk{b}_ln2_g"], s[f"blk{b}_ln2_b"] = (D,), (D,)
        d = D
        for l, u in enumerate(self.final_processing_arch):
            s[f"fin{l}_w"], s[f"fin{l}_b"] = (d, u), (u,)
            d = u
        s["out_w"], s["out_b"] = (d, self.output_dimensionality), (self.output_dimensionality,)
        return s

    def init_params(self, seed: int = 0) -> Dict[str, np.ndarray]:
        """Keras defaults: glorot-uniform kernels (receptive-field fan convention for the attention einsum kernels), zero
        biases, LayerNormalization gamma = 1, beta = 0."""
        rng = np.random.default_rng(seed)
        p = {}
        for name, shp in self.shapes.items():
            if name.endswith("_g"):
                a = np.ones(shp)
            elif name.endswith("_b"):
                a = np.zeros(shp)
            else:
                if len(shp) == 2:
                    fan_in, fan_out = shp
                elif "_o_w" in name:
                    fan_in, fan_out = shp[0] * shp[1], shp[2]
                else:
                    fan_in, fan_out = shp[0], shp[1] * shp[2]
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                a = rng.uniform(-lim, lim, size=shp)
            p[name] = a.astype(np.float32)
        return p

    def set_params(self, p: Dict[str, np.ndarray]) -> None:
        flat = np.zeros(self.n_alloc, dtype=np.float32)
        for name, shp in self.shapes.items():
            a = np.asarray(p[name], dtype=np.float32).reshape(-1)
            assert a.size == int(np.prod(shp)), name
            flat[self.offsets[name]: self.offsets[name] + a.size] = a
        self.params.copy_(torch.from_numpy(flat).to(self.params.dtype))

    def _unflatten(self, t: torch.Tensor) -> Dict[str, np.ndarray]:
        flat = t.detach().cpu().numpy()
        return {name: flat[self.offsets[name]: self.offsets[name] + int(np.prod(shp))].reshape(shp).copy()
                for name, shp in self.shapes.items()}

    def get_params(self) -> Dict[str, np.ndarray]:
        return self._unflatten(self.params)

    def get_grads(self) -> Dict[str, np.ndarray]:
        return self._unflatten(self.grads)

    @property
    def trainable_variables(self) -> List[torch.Tensor]:
        """particle_encoder.trainable_variables + set_transformer.trainable_variables: views into the flat buffer."""
        return [self.params[self.offsets[n]: self.offsets[n] + int(np.prod(s))].view(*s) for n, s in self.shapes.items()]

    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.t_dev.zero_()

    # ---- plan: workspace map + descriptor tables for a (batch, particles) shape ------------------------------------------
    def _stream(self):
        return c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _plan(self, B: int, P: int) -> dict:
        key = (B, P)
        if key in self._plans:
            self._plans[key] = self._plans.pop(key)   # most recently used last
            return self._plans[key]
        D, H, K = self.bottleneck_dimension, self.number_heads_per_mha, self.key_dim
        HK, T = H * K, B * P
        ldS = _align4(P)
        impl = self.attention_impl   # fixed by the constructor: flash for key_dim == 128 unless attention="gemm"
        F0 = self.particle_feature_dimensions
        pe_w = F0 * self.number_positional_encoding_frequencies
        enc_units = self.particle_encoder_arch_spec + [2 * D]
        ff = self.ff_arch_per_block
        off: Dict[str, int] = {}
        o = 0

        def take(name, n):
            nonlocal o
            off[name] = o
            o = _align4(o + int(n)) + 0
            return off[name]

        take("feats", T * F0)
        take("pe", T * pe_w)
        for l, u in enumerate(enc_units):
            take(f"enc_h{l}", T * u)                # last one = enc_out (mu | raw logvar)
        take("x0", T * D)                           # u = sampled embeddings
        for b in range(self.number_attention_blocks):
            for nm in ("q", "k", "v", "ctx"):
                take(f"b{b}_{nm}", T * HK)
            if impl == "gemm":
                take(f"b{b}_S", B * H * P * ldS)    # attention probabilities (stashed for the backward)
            else:
                take(f"b{b}_lse", B * H * P)        # per-query log-sum-exp (the flash backward recomputes the rest)
            take(f"b{b}_mha", T * D)
            take(f"b{b}_xhat1", T * D); take(f"b{b}_rstd1", T); take(f"b{b}_h", T * D)
            d = D
            for l, u in enumerate(ff):
                take(f"b{b}_ff{l}", T * u)
            take(f"b{b}_xhat2", T * D); take(f"b{b}_rstd2", T); take(f"b{b}_x", T * D)
        take("pool", B * D)
        for l, u in enumerate(self.final_processing_arch):
            take(f"fin{l}", B * u)
        take("pred", B * self.output_dimensionality)
        take("g_pred", B * self.output_dimensionality)
        take("out3", 4)
        take("kl_sum", 4)
        # backward scratch (reused by every block)
        for l, u in enumerate(self.final_processing_arch):
            take(f"g_fin{l}", B * u)
        take("g_pool", B * D)
        # g_x / g_s: gradient w.r.t. a block's output / input, ping-ponging from block to block (no copy); g_a: both addends
        # of LN2; g_z: feed-forward pre-activation; g_h: the feed-forward branch's gradient w.r.t. h
        take("g_x", T * D); take("g_s", T * D); take("g_a", T * D); take("g_z", T * D); take("g_h", T * D)
        for l, u in enumerate(ff[:-1]):
            take(f"g_ff{l}", T * u)
        for nm in ("q", "k", "v", "ctx"):
            take(f"g_{nm}", T * HK)
        if impl == "gemm":
            take("g_S", B * H * P * ldS)
        else:
            take("attn_delta", int(self.lib.dib_attention_bwd_workspace_bytes(B, P, H)) // 4)   # delta + dQ key-block partials
        for nm in ("q", "k", "v"):
            take(f"g_x{nm}", T * D)
        # split-K of the two skinny [T, heads*key_dim] x [heads*key_dim, D] products when there are few row tiles (_SplitKGemm)
        ksplit, mt = 1, (T + 63) // 64
        if mt < 128:
            for cand in (8, 4, 2):
                if HK % (cand * 32) == 0 and HK // cand >= 64:
                    ksplit = cand
                    break
        if ksplit > 1:
            take("ksplit_ws", 3 * ksplit * T * D)
        for l, u in enumerate(enc_units):
            take(f"g_enc_h{l}", T * u)
        ln_ws = int(self.lib.dib_add_layernorm_bwd_workspace_bytes(T, D)) // 4
        take("ln_ws", ln_ws)
        take("kl_ws", int(self.lib.dib_token_kl_workspace_bytes(T, D)) // 4 + 4)
        # the token-wise half of every block as one launch per direction (csrc/dib_st_chain.h) for up to 4096 tokens
        chain_descs = []
        if getattr(self, "use_chain", True) and len(ff) <= 3:
            for b in range(self.number_attention_blocks):
                pre = f"blk{b}_"
                dsc = _BlockDesc()
                dsc.o_w, dsc.o_b = self.offsets[pre + "o_w"], self.offsets[pre + "o_b"]
                dsc.ln1_g, dsc.ln1_b = self.offsets[pre + "ln1_g"], self.offsets[pre + "ln1_b"]
                dsc.ln2_g, dsc.ln2_b = self.offsets[pre + "ln2_g"], self.offsets[pre + "ln2_b"]
                for l, u in enumerate(ff):
                    dsc.ff_w[l], dsc.ff_b[l], dsc.ff_width[l] = self.offsets[pre + f"ff{l}_w"], self.offsets[pre + f"ff{l}_b"], u
                dsc.n_ff, dsc.D, dsc.HK, dsc.eps, dsc.act = len(ff), D, HK, self.layer_norm_epsilon, ACT_RELU
                chain_descs.append(dsc)
            if not (chain_descs and all(self.lib.dib_st_chain_supported(ctypes.byref(dsc), T) for dsc in chain_descs)):
                chain_descs = []
        if chain_descs:
            take("chain_ws", int(self.lib.dib_st_chain_workspace_bytes(T, D)) // 4)
        # Deferred weight gradients (round 6): on the chain path every block keeps the operands of its weight gradients in
        # buffers of its own - dL/d(feed-forward pre-activations), dL/dq|k|v, and the gradient of LN1's addends (slot 0 of
        # b{b}_dx; slots 1.. are the split-K slabs of the q/k/v input gradient, so that ONE fixed-order sum over the slots is
        # the gradient handed to the next block and slot 0 stays what the output projection's weight gradient contracts
        # with) - and ALL blocks' weight gradients run at the end of the backward as one grouped launch per shape class
        # (q/k/v: 3 x blocks groups of [D, HK]; output projection: [HK, D]; feed-forward) instead of 3 launches per block:
        # at the notebook's size 18 launches of 7-18 us on a few dozen workgroups each become 3 that fill the chip.
        defer = bool(chain_descs) and ksplit > 1 and bool(getattr(self, "defer_wgrads", True))
        if defer:
            for b in range(self.number_attention_blocks):
                take(f"b{b}_g_z", T * D)
                for l, u in enumerate(ff[:-1]):
                    take(f"b{b}_g_ff{l}", T * u)
                for nm in "qkv":
                    take(f"b{b}_g_{nm}", T * HK)
                take(f"b{b}_dx", (1 + max(3 * ksplit, H)) * T * D)   # slot 0 + split-K slabs, or + one slab per head (attn_bwd_proj)
        gn = (lambda b, nm: f"b{b}_{nm}") if defer else (lambda b, nm: nm)   # per-block / shared gradient buffer names
        take("loss_ws", int(self.lib.dib_loss_rows_workspace_bytes(B)) // 4 + 4)
        ws = torch.zeros(o, dtype=torch.float32, device=self.device)
        # flash attention, stash mode: one score-tile buffer per block, outside the fp32-indexed workspace (its own allocation:
        # 3.2 GB per block at 4 x 4096); None = recompute mode
        # It is allocated LAZILY by the first forward that a backward will follow (_ensure_stash): evaluation-only shapes
        # (validation batches, the sampled second pass of fit) never own one.
        stash = None
        stash_block_bytes = int(self.lib.dib_attention_stash_bytes(B, P, H)) if impl == "flash" else 0

        # weight-gradient target: contraction over T tokens is split into slabs when T is large (fixed-order reduce)
        # (from 512 tokens up: with one slab the q/k/v and output-projection wgrads of the reference size, 1600 tokens, ran on
        # 12-36 workgroups looping over all rows - 110-137 us each, the top entries of the first profile)
        # (64-row slabs up to 2048 tokens: at 1600 tokens the 6 slabs of the T // 256 rule left the feed-forward wgrads on 6
        # workgroups walking 9 dependent k-tiles each - 24 us per launch, 28 such launches per step)
        # (deferred weight gradients: many groups per launch fill the chip with FEW splits, and every slab costs the optimizer's
        # launch a pass over the whole gradient buffer - 25 slabs x 5.2 MB were 130 MB, 33 us of a 1.29 ms step at the notebook's
        # size; `deferred_max_slabs`)
        nsplit = max(1, min(int(getattr(self, "deferred_max_slabs", 8)) if defer else 32, T // 64))
        rps = ((T + nsplit - 1) // nsplit + 31) // 32 * 32
        nsplit = (T + rps - 1) // rps
        slabs = torch.zeros(nsplit * self.n_alloc, dtype=torch.float32, device=self.device) if nsplit > 1 else None
        gt = slabs if nsplit > 1 else self.grads
        po = self.offsets

        def dense_fwd(x, kin, w, b, y, kout, act, M):
            return _Gemm(0, [_d(off[x], kin, po[w], kout, off[y], kout, M, kout, kin, bias_off=po[b])], ws, self.params, ws,
                         bias=self.params, act=act)

        def dense_dgrad(dy, kout, w, dx, kin, M, aux=None, act=0):
            return _Gemm(1, [_d(off[dy], kout, po[w], kout, off[dx], kin, M, kin, kout,
                                aux_off=off[aux] if aux else 0, ldaux=kin)], ws, self.params, ws,
                         aux=ws if aux else None, act=act if aux else 0)

        def dense_wgrad(x, kin, dy, kout, w, b, M, split=True):
            ns, r = (nsplit, rps) if split else (1, max(M, 1))
            return _Gemm(2, [_d(off[x], kin, off[dy], kout, po[w], kout, kin, kout, M, bias_off=po[b])], ws, ws, gt,
                         bias_out=gt, nsplit=ns, rows_per_split=r, split_stride=self.n_alloc)

        g: Dict[str, _Gemm] = {}
        # particle encoder (shared by all particles): [T, 60] -> 128 -> 128 -> 64
        kin, src = pe_w, "pe"
        for l, u in enumerate(enc_units):
            act = ACT_LEAKY01 if l < len(enc_units) - 1 else ACT_NONE
            g[f"enc{l}_fwd"] = dense_fwd(src, kin, f"enc{l}_w", f"enc{l}_b", f"enc_h{l}", u, act, T)
            g[f"enc{l}_wgrad"] = dense_wgrad(src, kin, f"g_enc_h{l}", u, f"enc{l}_w", f"enc{l}_b", T)
            if l > 0:
                g[f"enc{l}_dgrad"] = dense_dgrad(f"g_enc_h{l}", u, f"enc{l}_w", f"g_enc_h{l - 1}", kin, T, aux=f"enc_h{l - 1}",
                                                 act=ACT_LEAKY01)
            kin, src = u, f"enc_h{l}"
        bh = [(b_, h_) for b_ in range(B) for h_ in range(H)]
        dw_qkv, dw_o, dw_ff = [], [], []   # deferred weight gradients: descriptors of all blocks by shape class
        for b in range(self.number_attention_blocks):
            xin = "x0" if b == 0 else f"b{b - 1}_x"
            pre = f"blk{b}_"
            # q, k, v projections: 3 groups
            qkv_descs = [_d(off[xin], D, po[pre + nm + "_w"], HK, off[f"b{b}_{nm}"], HK, T, HK, D, bias_off=po[pre + nm + "_b"])
                         for nm in "qkv"]
            # from skinny_k_min_tokens tokens up the projections out of the D-wide residual stream are store-bound streaming
            # launches (dib_gemm_skinny_k); below, the tiled grouped GEMM
            skinny = T >= self.skinny_k_min_tokens
            mk = _SkinnyKGemm if skinny and _SkinnyKGemm.fits(0, qkv_descs) else _Gemm
            g[f"b{b}_qkv_fwd"] = mk(0, qkv_descs, ws, self.params, ws, bias=self.params)
            gemm_attn = impl == "gemm"
            # scores S_bh = Q_bh K_bh^T (scale folded into the softmax)
            if gemm_attn:
                g[f"b{b}_qk"] = _Gemm(1, [_d(off[f"b{b}_q"] + bi * P * HK + hi * K, HK, off[f"b{b}_k"] + bi * P * HK + hi * K, HK,
                                           off[f"b{b}_S"] + (bi * H + hi) * P * ldS, ldS, P, P, K) for bi, hi in bh], ws, ws, ws)
            # ctx_bh = P_bh V_bh
            if gemm_attn:
                g[f"b{b}_pv"] = _Gemm(0, [_d(off[f"b{b}_S"] + (bi * H + hi) * P * ldS, ldS, off[f"b{b}_v"] + bi * P * HK + hi * K, HK,
                                           off[f"b{b}_ctx"] + bi * P * HK + hi * K, HK, P, K, P) for bi, hi in bh], ws, ws, ws)
            if ksplit > 1:
                ck = HK // ksplit
                g[f"b{b}_o_fwd"] = _SplitKGemm(
                    _Gemm(0, [_d(off[f"b{b}_ctx"] + s_ * ck, HK, po[pre + "o_w"] + s_ * ck * D, D, off["ksplit_ws"] + s_ * T * D, D,
                                 T, D, ck, bias_off=po[pre + "o_b"] if s_ == 0 else -1) for s_ in range(ksplit)],
                          ws, self.params, ws, bias=self.params),
                    ws, off["ksplit_ws"], T * D, ksplit, ws, off[f"b{b}_mha"], mode="defer")   # LN1 sums the slabs
            else:
                g[f"b{b}_o_fwd"] = dense_fwd(f"b{b}_ctx", HK, pre + "o_w", pre + "o_b", f"b{b}_mha", D, ACT_NONE, T)
            d, src = D, f"b{b}_h"
            for l, u in enumerate(ff):
                g[f"b{b}_ff{l}_fwd"] = dense_fwd(src, d, pre + f"ff{l}_w", pre + f"ff{l}_b", f"b{b}_ff{l}", u, ACT_RELU, T)
                d, src = u, f"b{b}_ff{l}"
            # ---- backward ----
            # feed-forward: g_z = dL/d(pre-activation of the last ff layer)
            nff = len(ff)
            gy, ky = "g_z", ff[-1]
            for l in range(nff - 1, -1, -1):
                kin_l = D if l == 0 else ff[l - 1]
                src_l = f"b{b}_h" if l == 0 else f"b{b}_ff{l - 1}"
                g[f"b{b}_ff{l}_wgrad"] = dense_wgrad(src_l, kin_l, gy, ky, pre + f"ff{l}_w", pre + f"ff{l}_b", T)
                if l > 0:
                    g[f"b{b}_ff{l}_dgrad"] = dense_dgrad(gy, ky, pre + f"ff{l}_w", f"g_ff{l - 1}", kin_l, T, aux=src_l, act=ACT_RELU)
                    gy, ky = f"g_ff{l - 1}", kin_l
                else:
                    g[f"b{b}_ff0_dgrad"] = dense_dgrad(gy, ky, pre + "ff0_w", "g_h", D, T)
            # attention output projection
            gout = self._block_grad_names(b)[1]   # gradient w.r.t. the block's input x (= gradient of LN1's two addends)
            if defer:
                off[f"b{b}_gln1"] = off[f"b{b}_dx"]   # slot 0 of the block's dx region (alias)
            g[f"b{b}_o_wgrad"] = dense_wgrad(f"b{b}_ctx", HK, f"b{b}_gln1" if defer else gout, D, pre + "o_w", pre + "o_b", T)
            o_dgrad_descs = [_d(off[gout], D, po[pre + "o_w"], D, off["g_ctx"], HK, T, HK, D)]
            mk = _SkinnyKGemm if skinny and _SkinnyKGemm.fits(1, o_dgrad_descs) else _Gemm
            g[f"b{b}_o_dgrad"] = mk(1, o_dgrad_descs, ws, self.params, ws)
            if gemm_attn:
                g[f"b{b}_dv"] = _Gemm(2, [_d(off[f"b{b}_S"] + (bi * H + hi) * P * ldS, ldS, off["g_ctx"] + bi * P * HK + hi * K, HK,
                                           off[gn(b, "g_v")] + bi * P * HK + hi * K, HK, P, K, P) for bi, hi in bh], ws, ws, ws,
                                    nsplit=1, rows_per_split=max(P, 1))
            if gemm_attn:
                g[f"b{b}_dp"] = _Gemm(1, [_d(off["g_ctx"] + bi * P * HK + hi * K, HK, off[f"b{b}_v"] + bi * P * HK + hi * K, HK,
                                           off["g_S"] + (bi * H + hi) * P * ldS, ldS, P, P, K) for bi, hi in bh], ws, ws, ws)
            if gemm_attn:
                g[f"b{b}_dq"] = _Gemm(0, [_d(off["g_S"] + (bi * H + hi) * P * ldS, ldS, off[f"b{b}_k"] + bi * P * HK + hi * K, HK,
                                           off[gn(b, "g_q")] + bi * P * HK + hi * K, HK, P, K, P) for bi, hi in bh], ws, ws, ws)
            if gemm_attn:
                g[f"b{b}_dk"] = _Gemm(2, [_d(off["g_S"] + (bi * H + hi) * P * ldS, ldS, off[f"b{b}_q"] + bi * P * HK + hi * K, HK,
                                           off[gn(b, "g_k")] + bi * P * HK + hi * K, HK, P, K, P) for bi, hi in bh], ws, ws, ws,
                                    nsplit=1, rows_per_split=max(P, 1))
            qkv_wgrad_descs = [_d(off[xin], D, off[gn(b, f"g_{nm}")], HK, po[pre + nm + "_w"], HK, D, HK, T, bias_off=po[pre + nm + "_b"])
                               for nm in "qkv"]
            g[f"b{b}_qkv_wgrad"] = _Gemm(2, qkv_wgrad_descs, ws, ws, gt, bias_out=gt,
                                         nsplit=nsplit, rows_per_split=rps, split_stride=self.n_alloc)
            if defer:
                dw_qkv += qkv_wgrad_descs
                dw_o.append(_d(off[f"b{b}_ctx"], HK, off[f"b{b}_gln1"], D, po[pre + "o_w"], D, HK, D, T, bias_off=po[pre + "o_b"]))
            if chain_descs:
                # the feed-forward layers' weight gradients in one grouped launch (dy = the chain backward's g_ff); the output
                # projection's (dy = the gradient of LN1's addends = the block-input gradient buffer BEFORE the projections' dgrads
                # are added) and q / k / v's keep their own launches
                descs = []
                for l in range(nff):
                    kin_l, src_l = (D, f"b{b}_h") if l == 0 else (ff[l - 1], f"b{b}_ff{l - 1}")
                    dy_l = gn(b, "g_z" if l == nff - 1 else f"g_ff{l}")
                    descs.append(_d(off[src_l], kin_l, off[dy_l], ff[l], po[pre + f"ff{l}_w"], ff[l], kin_l, ff[l], T,
                                    bias_off=po[pre + f"ff{l}_b"]))
                # (one launch for ALL of them was tried: a grouped launch's grid is max-shape tiles x groups, and [1536, 32] next to
                # [32, 1536] made it 21 600 mostly empty workgroups - 48 us; the feed-forward layers share a shape class)
                g[f"b{b}_ff_wgrad"] = _Gemm(2, descs, ws, ws, gt, bias_out=gt, nsplit=nsplit, rows_per_split=rps,
                                            split_stride=self.n_alloc)
                if defer:
                    dw_ff += descs
            if defer:
                # the slabs land behind the LN1-addend gradient in the block's own region; the sum over all slots is taken by
                # the consumer: the previous block's chain launch sums them as it loads its tile (dib_st_chain_bwd g_out_slabs),
                # block 0's go through one reduce launch into the buffer the bottleneck's backward reads
                ck = HK // ksplit
                g[f"b{b}_qkv_dgrad"] = _SplitKGemm(
                    _Gemm(1, [_d(off[f"b{b}_g_{nm}"] + s_ * ck, HK, po[pre + nm + "_w"] + s_ * ck, HK,
                                 off[f"b{b}_dx"] + (1 + i_ * ksplit + s_) * T * D, D, T, D, ck)
                              for i_, nm in enumerate("qkv") for s_ in range(ksplit)], ws, self.params, ws),
                    ws, off[f"b{b}_dx"], T * D, 1 + 3 * ksplit, ws, off[gout], mode="store" if b == 0 else "defer")
            elif ksplit > 1:   # 3 projections x ksplit chunks -> 3 * ksplit slabs, summed straight into g_xq (= g_xq + g_xk + g_xv)
                ck = HK // ksplit
                g[f"b{b}_qkv_dgrad"] = _SplitKGemm(
                    _Gemm(1, [_d(off[f"g_{nm}"] + s_ * ck, HK, po[pre + nm + "_w"] + s_ * ck, HK,
                                 off["ksplit_ws"] + (i_ * ksplit + s_) * T * D, D, T, D, ck)
                              for i_, nm in enumerate("qkv") for s_ in range(ksplit)], ws, self.params, ws),
                    ws, off["ksplit_ws"], T * D, 3 * ksplit, ws, off[self._block_grad_names(b)[1]], mode="add")
            else:
                g[f"b{b}_qkv_dgrad"] = _Gemm(1, [_d(off[f"g_{nm}"], HK, po[pre + nm + "_w"], HK, off[f"g_x{nm}"], D, T, D, HK)
                                                 for nm in "qkv"], ws, self.params, ws)
        # head: pooled [B, D] -> Dense(256, LeakyReLU(0.1)) -> Dense(out)
        d, src = D, "pool"
        for l, u in enumerate(self.final_processing_arch):
            g[f"fin{l}_fwd"] = dense_fwd(src, d, f"fin{l}_w", f"fin{l}_b", f"fin{l}", u, ACT_LEAKY01, B)
            d, src = u, f"fin{l}"
        g["out_fwd"] = dense_fwd(src, d, "out_w", "out_b", "pred", self.output_dimensionality, ACT_NONE, B)
        g["out_wgrad"] = dense_wgrad(src, d, "g_pred", self.output_dimensionality, "out_w", "out_b", B, split=False)
        nfin = len(self.final_processing_arch)
        if nfin:
            g["out_dgrad"] = dense_dgrad("g_pred", self.output_dimensionality, "out_w", f"g_fin{nfin - 1}", d, B,
                                         aux=f"fin{nfin - 1}", act=ACT_LEAKY01)
        else:
            g["out_dgrad"] = dense_dgrad("g_pred", self.output_dimensionality, "out_w", "g_pool", d, B)
        for l in range(nfin - 1, -1, -1):
            kin_l = D if l == 0 else self.final_processing_arch[l - 1]
            src_l = "pool" if l == 0 else f"fin{l - 1}"
            u = self.final_processing_arch[l]
            g[f"fin{l}_wgrad"] = dense_wgrad(src_l, kin_l, f"g_fin{l}", u, f"fin{l}_w", f"fin{l}_b", B, split=False)
            if l > 0:
                g[f"fin{l}_dgrad"] = dense_dgrad(f"g_fin{l}", u, f"fin{l}_w", f"g_fin{l - 1}", kin_l, B, aux=src_l, act=ACT_LEAKY01)
            else:
                g["fin0_dgrad"] = dense_dgrad(f"g_fin{l}", u, "fin0_w", "g_pool", D, B)
        # the particle encoder (PositionalEncoding -> Dense(LeakyReLU(0.1))* -> Dense) on the row-tile MLP kernels for up to 2048
        # tokens: one launch forward (encoding included), one for the dgrad chain, instead of 4 + 2
        enc_mlp = None
        if getattr(self, "encoder_row_tiles", True) and 2 <= len(enc_units) <= 4:
            from .dense import _MlpDesc
            dsc = _MlpDesc()
            for l, u in enumerate(enc_units):
                dsc.w_off[l], dsc.b_off[l], dsc.width[l] = po[f"enc{l}_w"], po[f"enc{l}_b"], u
            dsc.n_hidden, dsc.in_dim, dsc.n_freq, dsc.act = len(enc_units) - 1, F0, self.number_positional_encoding_frequencies, ACT_LEAKY01
            if self.lib.dib_mlp_small_supported(ctypes.byref(dsc), T):
                nh = len(enc_units) - 1
                hp = lambda pre: (c_void_p * 3)(*[_ptr(ws, off[f"{pre}{l}"]).value if l < nh else None for l in range(3)])
                enc_mlp = dict(desc=dsc, h=hp("enc_h"), g=hp("g_enc_h"))
        # the head (pooled neighbourhood -> Dense(LeakyReLU(0.1))* -> Dense(1) -> BCE) as ONE launch for its whole share of a training
        # step (dib_mlp_small_head_step): forward, loss, gradient of the logit, dgrad chain, the output layer's gradient
        head_mlp = None
        if getattr(self, "head_row_tiles", True) and self.output_dimensionality == 1 and 1 <= nfin <= 3:
            from .dense import _MlpDesc
            dsc = _MlpDesc()
            for l, u in enumerate(self.final_processing_arch):
                dsc.w_off[l], dsc.b_off[l], dsc.width[l] = po[f"fin{l}_w"], po[f"fin{l}_b"], u
            dsc.w_off[nfin], dsc.b_off[nfin], dsc.width[nfin] = po["out_w"], po["out_b"], 1
            dsc.n_hidden, dsc.in_dim, dsc.n_freq, dsc.act = nfin, D, 1, ACT_LEAKY01
            if self.lib.dib_mlp_small_head_supported(ctypes.byref(dsc), B):
                hp = lambda pre: (c_void_p * 3)(*[_ptr(ws, off[f"{pre}{l}"]).value if l < nfin else None for l in range(3)])
                head_ws = torch.zeros(int(self.lib.dib_mlp_small_head_workspace_bytes(ctypes.byref(dsc), B)) // 4 + 4,
                                      dtype=torch.float32, device=self.device)
                head_mlp = dict(desc=dsc, h=hp("fin"), g=hp("g_fin"), ws=head_ws)
        deferred = []
        if defer:
            # the particle encoder's weight gradients contract over the same T tokens and fit the feed-forward class's tiles:
            # three more groups of that launch instead of three launches; the head's (contraction over the B neighbourhoods)
            # become one grouped launch
            dw_ff += [dict(zip(DESC.names, g[f"enc{l}_wgrad"].host[0])) for l in range(len(enc_units))]
            g["head_wgrads"] = _Gemm(2, [dict(zip(DESC.names, g[k].host[0])) for k in
                                         ([] if head_mlp is not None else ["out_wgrad"]) + [f"fin{l}_wgrad" for l in range(nfin)]],
                                     ws, ws, gt, bias_out=gt, nsplit=1, rows_per_split=max(B, 1), split_stride=self.n_alloc)
            # one grouped launch per shape class.  A grouped launch's grid is (splits, tiles of the LARGEST group shape, groups):
            # the split count of each class is chosen so that its workgroups make about `deferred_wgrad_target_wgs` - many
            # groups need few, long splits (1536: the best of 384 / 512 / 768 / 1024 / 1536 at the notebook's size, 1.335 ...
            # 1.319 ms per step, profiles/r06b_set_transformer_deferred_wgrads_ab.txt; 2048 and 3072 no better, r06c); slabs
            # beyond a launch's count are never written and stay zero (the slab buffer is zero-initialised and every launch
            # always writes the same slabs)
            target = int(getattr(self, "deferred_wgrad_target_wgs", 1536))
            for name, descs, (tm, tn) in (("dw_qkv", dw_qkv, (64, 128)), ("dw_o", dw_o, (128, 64)), ("dw_ff", dw_ff, (128, 128))):
                mm, nn = max(d["M"] for d in descs), max(d["N"] for d in descs)
                tiles = len(descs) * ((mm + tm - 1) // tm) * ((nn + tn - 1) // tn)
                ns = max(1, min(nsplit, int(round(target / tiles))))
                r = ((T + ns - 1) // ns + 63) // 64 * 64
                ns = (T + r - 1) // r
                g[name] = _Gemm(2, descs, ws, ws, gt, bias_out=gt, nsplit=ns, rows_per_split=r, split_stride=self.n_alloc)
                deferred.append(name)
        for gg in g.values():
            gg.upload(self.device)
        plan = dict(impl=impl, B=B, P=P, T=T, ldS=ldS, off=off, ws=ws, g=g, nsplit=nsplit, slabs=slabs, gt=gt, pe_w=pe_w,
                    enc_units=enc_units, stash=stash, stash_block_bytes=stash_block_bytes, stash_denied=None, ksplit=ksplit,
                    chain=chain_descs, deferred_wgrads=deferred, enc_mlp=enc_mlp, head_mlp=head_mlp,
                    # <= 64 particles: the q / k / v projections inside the attention forward (dib_attention_fwd_proj)
                    attn_proj=bool(impl == "flash" and getattr(self, "attention_proj", True)
                                   and self.lib.dib_attention_fwd_proj_supported(P, K, D)),
                    # ... and their input gradient inside the attention backward (one slab per head behind the LN1-addend
                    # gradient; needs the per-block dx regions of the deferred mode and the 8-wave kernel)
                    attn_bwd_proj=bool(defer and impl == "flash" and getattr(self, "attention_bwd_proj", True)
                                       and self.lib.dib_attention_fwd_proj_supported(P, K, D)
                                       and _lib.get_tuning("attn_small_bwd_waves") >= 8),
                    qkv_off=[((ctypes.c_int64 * 3)(*[po[f"blk{b}_{nm}_w"] for nm in "qkv"]),
                              (ctypes.c_int64 * 3)(*[po[f"blk{b}_{nm}_b"] for nm in "qkv"]))
                             for b in range(self.number_attention_blocks)])
        # a plan holds the whole step workspace + the gradient slabs (166 MB at 4 x 4096): keep the few most recent shapes
        # (training batch, validation batch, a ragged tail), evict least recently used beyond that
        step_keys = [k for k in self._plans if k[0] != "enc" and k not in self._graphs]   # a captured graph pins its plan
        if len(step_keys) >= self.max_step_plans:
            self._plans.pop(step_keys[0])
        self._plans[key] = plan
        return plan

    def _ensure_stash(self, pl) -> bool:
        """Flash-attention score stash of plan `pl` (one buffer per attention block; 19.3 GB for 4 x 4096 particles x 6 blocks),
        allocated on first use.  It is granted only if (a) all blocks' tiles of this shape plus the stashes of every other live
        plan fit `attention_score_stash_bytes`, (b) the device has that much memory free (free HBM + what the caching
        allocator holds unused, less a 1 GiB reserve) and (c) the allocation itself succeeds; otherwise the plan stays in
        recompute mode (pl["stash_denied"] says why) - a fallback in MEMORY POLICY only, the same kernels run."""
        if pl["stash"] is not None:
            return True
        if pl["stash_block_bytes"] <= 0 or pl["stash_denied"] is not None:
            return False
        need = pl["stash_block_bytes"] * self.number_attention_blocks
        live = sum(q["stash_block_bytes"] * self.number_attention_blocks for q in self._plans.values()
                   if isinstance(q, dict) and q.get("stash") is not None)
        if live + need > self.attention_score_stash_bytes:
            pl["stash_denied"] = f"budget: {live} live + {need} needed > attention_score_stash_bytes = {self.attention_score_stash_bytes}"
            return False
        free, _total = torch.cuda.mem_get_info(self.device)
        cached = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        if need > free + cached - (1 << 30):
            pl["stash_denied"] = f"memory: {need} needed, {free} free + {cached} cached on the device"
            return False
        bufs = []
        try:
            for _ in range(self.number_attention_blocks):
                bufs.append(torch.empty(pl["stash_block_bytes"] // 4, dtype=torch.float32, device=self.device))
        except torch.cuda.OutOfMemoryError:
            del bufs
            torch.cuda.empty_cache()
            pl["stash_denied"] = "memory: allocation failed"
            return False
        pl["stash"] = bufs
        return True

    def _block_grad_names(self, b: int):
        """(buffer holding dL/d(output of block b), buffer receiving dL/d(input of block b)): "g_x" and "g_s" alternate from
        block to block in backward order, so a block's result is the next block's input without a copy."""
        k = self.number_attention_blocks - 1 - b
        return ("g_x", "g_s") if k % 2 == 0 else ("g_s", "g_x")

    def _view(self, plan, name, *shape):
        o = plan["off"][name]
        return plan["ws"][o: o + int(np.prod(shape))].view(*shape)

    # ---- forward (notebook train_step, forward part) -------------------------------------------------------------------
    def forward(self, batch_inp, step: Optional[int] = None, deterministic: bool = False, row0: int = 0,
                embs_reparam=None, _step_from_device: bool = False, for_backward: bool = True,
                _skip_head: bool = False) -> torch.Tensor:
        """embs = particle_encoder(batch_inp); logvar - 3; reparameterised sample; kl; loci_prediction = set_transformer(u).
        batch_inp [B, P, particle_feature_dimensions].  Returns the logits [B, out]; self.last holds kl (device scalar).
        embs_reparam [B, P, bottleneck] (optional): use these sampled embeddings instead of the library's counter-based
        noise (the notebook evaluates `set_transformer(tf.random.normal(...))` on its own samples; also how the golden
        fixture, which carries its own noise, is replayed).
        for_backward=False (evaluation passes): the flash attention does not write its score stash (3.2 GB per block at
        4 x 4096 particles); a loss_and_backward after such a forward recomputes the scores instead.
        Backward after any of the three forwards is consistent with it: the bottleneck's noise term is recovered as
        eps * sigma = x0 - mu from the sample that was actually used (dib_token_reparam_kl_bwd) - library noise, an injected
        sample (treated as mu + sigma * eps with its implied eps held fixed: the reparameterised gradient of that sample), or
        the deterministic forward (x0 = mu: the term vanishes).  (Round 2 regenerated the library's eps in the backward, which
        was silently wrong for the last two - advisor finding.)"""
        x = batch_inp if isinstance(batch_inp, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(batch_inp, dtype=np.float32))
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, P, F0 = x.shape
        assert F0 == self.particle_feature_dimensions
        pl = self._plan(B, P)
        use_stash = bool(for_backward) and pl["impl"] == "flash" and self._ensure_stash(pl)
        lib, st, ws, off, g = self.lib, self._stream(), pl["ws"], pl["off"], pl["g"]
        T, D = pl["T"], self.bottleneck_dimension
        step = self._step if step is None else int(step)
        self._view(pl, "feats", T, F0).copy_(x.view(T, F0))
        ne = len(pl["enc_units"])
        em = pl["enc_mlp"]
        if em is not None:
            check(lib.dib_mlp_small_fwd(ctypes.byref(em["desc"]), _ptr(self.params), _ptr(ws, off["feats"]), F0, None, T,
                                        _ptr(ws, off["pe"]), em["h"], _ptr(ws, off[f"enc_h{ne - 1}"]), st), "dib_mlp_small_fwd")
        else:
            check(lib.dib_positional_encoding(_ptr(ws, off["feats"]), F0, T, F0, self.number_positional_encoding_frequencies,
                                              _ptr(ws, off["pe"]), st), "dib_positional_encoding")
            for l in range(ne):
                g[f"enc{l}_fwd"].run(lib, st)
        check(lib.dib_token_reparam_kl_fwd(_ptr(ws, off[f"enc_h{ne - 1}"]), T, D, self.logvar_initialization, self.noise_seed,
                                           step & 0xFFFFFFFF, _ptr(self.step_dev) if _step_from_device else c_void_p(0),
                                           int(row0), 1 if deterministic else 0, _ptr(ws, off["x0"]),
                                           _ptr(ws, off["kl_sum"]), _ptr(ws, off["kl_ws"]), st), "dib_token_reparam_kl_fwd")
        if embs_reparam is not None:
            er = embs_reparam if isinstance(embs_reparam, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(embs_reparam, dtype=np.float32))
            self._view(pl, "x0", T, D).copy_(er.to(device=self.device, dtype=torch.float32).reshape(T, D))
        scale = 1.0 / math.sqrt(self.key_dim)
        H = self.number_heads_per_mha
        for b in range(self.number_attention_blocks):
            xin = "x0" if b == 0 else f"b{b - 1}_x"
            pre = f"blk{b}_"
            if not pl["attn_proj"]:
                g[f"b{b}_qkv_fwd"].run(lib, st)
            if pl["impl"] == "gemm":
                g[f"b{b}_qk"].run(lib, st)
                check(lib.dib_softmax_rows_fwd(_ptr(ws, off[f"b{b}_S"]), B * H * P, P, pl["ldS"], scale, st), "dib_softmax_rows_fwd")
                g[f"b{b}_pv"].run(lib, st)
            else:
                HK = H * self.key_dim
                if pl["attn_proj"]:   # q, k, v = the block input's projections, computed (and written) by the attention launch
                    wo, bo = pl["qkv_off"][b]
                    check(lib.dib_attention_fwd_proj(_ptr(ws, off[xin]), D, _ptr(self.params), wo, bo, B, P, H, self.key_dim, D, HK, scale,
                                                     _ptr(ws, off[f"b{b}_q"]), _ptr(ws, off[f"b{b}_k"]), _ptr(ws, off[f"b{b}_v"]),
                                                     _ptr(ws, off[f"b{b}_ctx"]), _ptr(ws, off[f"b{b}_lse"]), st), "dib_attention_fwd_proj")
                else:
                    check(lib.dib_attention_fwd(_ptr(ws, off[f"b{b}_q"]), _ptr(ws, off[f"b{b}_k"]), _ptr(ws, off[f"b{b}_v"]), B, P, H,
                                                self.key_dim, HK, scale, _ptr(ws, off[f"b{b}_ctx"]), _ptr(ws, off[f"b{b}_lse"]),
                                                _ptr(pl["stash"][b]) if use_stash else c_void_p(0), st), "dib_attention_fwd")
            if pl["chain"]:   # output projection -> Add + LN -> feed-forward -> Add + LN: one launch (16-token tiles)
                ffp = (c_void_p * 3)(*[_ptr(ws, off[f"b{b}_ff{l}"]) for l in range(len(self.ff_arch_per_block))])
                check(lib.dib_st_chain_fwd(ctypes.byref(pl["chain"][b]), T, _ptr(self.params), _ptr(ws, off[f"b{b}_ctx"]),
                                           _ptr(ws, off[xin]), _ptr(ws, off[f"b{b}_h"]), _ptr(ws, off[f"b{b}_xhat1"]),
                                           _ptr(ws, off[f"b{b}_rstd1"]), ffp, _ptr(ws, off[f"b{b}_x"]), _ptr(ws, off[f"b{b}_xhat2"]),
                                           _ptr(ws, off[f"b{b}_rstd2"]), st), "dib_st_chain_fwd")
                continue
            g[f"b{b}_o_fwd"].run(lib, st)
            ks = pl["ksplit"]   # split-K output projection: its slabs are the second addend
            check(lib.dib_add_layernorm_fwd(_ptr(ws, off[xin]), _ptr(ws, off["ksplit_ws"] if ks > 1 else off[f"b{b}_mha"]),
                                            max(ks, 1), T * D, T, D,
                                            _ptr(self.params, self.offsets[pre + "ln1_g"]), _ptr(self.params, self.offsets[pre + "ln1_b"]),
                                            self.layer_norm_epsilon, _ptr(ws, off[f"b{b}_h"]), _ptr(ws, off[f"b{b}_xhat1"]),
                                            _ptr(ws, off[f"b{b}_rstd1"]), st), "dib_add_layernorm_fwd")
            for l in range(len(self.ff_arch_per_block)):
                g[f"b{b}_ff{l}_fwd"].run(lib, st)
            last_ff = f"b{b}_ff{len(self.ff_arch_per_block) - 1}"
            check(lib.dib_add_layernorm_fwd(_ptr(ws, off[f"b{b}_h"]), _ptr(ws, off[last_ff]), 1, 0, T, D,
                                            _ptr(self.params, self.offsets[pre + "ln2_g"]), _ptr(self.params, self.offsets[pre + "ln2_b"]),
                                            self.layer_norm_epsilon, _ptr(ws, off[f"b{b}_x"]), _ptr(ws, off[f"b{b}_xhat2"]),
                                            _ptr(ws, off[f"b{b}_rstd2"]), st), "dib_add_layernorm_fwd")
        xl = "x0" if self.number_attention_blocks == 0 else f"b{self.number_attention_blocks - 1}_x"
        check(lib.dib_mean_pool_fwd(_ptr(ws, off[xl]), B, P, D, _ptr(ws, off["pool"]), st), "dib_mean_pool_fwd")
        # (_skip_head: a training step whose loss_and_backward runs the head's forward itself, in its one-launch head step -
        # the returned logits are then those of the PREVIOUS call until loss_and_backward has run)
        if not (_skip_head and pl["head_mlp"] is not None):
            for l in range(len(self.final_processing_arch)):
                g[f"fin{l}_fwd"].run(lib, st)
            g["out_fwd"].run(lib, st)
        self.attention_impl = pl["impl"]   # what this batch shape ran on (reporting)
        self.last = dict(plan=pl, step=step, row0=int(row0), B=B, P=P, stash=use_stash,
                         kl=self._view(pl, "kl_sum", 1) / B)   # "sum over dimension and particles, avg over batch"
        return self._view(pl, "pred", B, self.output_dimensionality)

    # ---- loss + backward ----------------------------------------------------------------------------------------------
    def loss_and_backward(self, is_loci, inv_global_batch: Optional[float] = None, reduce: bool = True) -> None:
        """bce_losses = mean BCE(is_loci, logits); loss = bce_losses + beta_var * kl; tape.gradient(loss, variables).
        Gradients land in self.grads; self.last gets bce (device scalar).  reduce=False: the batch-slab partials are left for
        adam_step(fused_reduce=True), which sums them in the optimizer's own launch (dib_reduce_adam_step)."""
        pl = self.last["plan"]
        B, P, T = self.last["B"], self.last["P"], pl["T"]
        lib, st, ws, off, g = self.lib, self._stream(), pl["ws"], pl["off"], pl["g"]
        D, H = self.bottleneck_dimension, self.number_heads_per_mha
        y = is_loci if isinstance(is_loci, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(is_loci, dtype=np.float32))
        y = y.to(device=self.device, dtype=torch.float32).reshape(B, -1).contiguous()
        inv = 1.0 / B if inv_global_batch is None else float(inv_global_batch)
        gt = pl["gt"]
        if pl["nsplit"] == 1:
            self.grads.zero_()  # blocks are overwritten; alignment gaps stay zero
        dw = bool(pl["deferred_wgrads"])
        hm = pl["head_mlp"]
        if hm is not None:
            # the head's whole share of the step in one launch: hidden layers, logit, BCE, dL/dlogit, dgrad chain down to the
            # pooled embedding, the output layer's gradient and {loss sum, #correct, loss sum * inv} (csrc/dib_small.h)
            check(lib.dib_mlp_small_head_step(ctypes.byref(hm["desc"]), _ptr(self.params), _ptr(ws, off["pool"]), B, _ptr(y), y.stride(0),
                                              LOSS_BCE_LOGITS, inv, hm["h"], hm["g"], _ptr(ws, off["pred"]), _ptr(ws, off["g_pred"]),
                                              _ptr(ws, off["g_pool"]), _ptr(gt), _ptr(ws, off["out3"]), _ptr(hm["ws"]), st),
                  "dib_mlp_small_head_step")
            if dw:
                g["head_wgrads"].run(lib, st)
            else:
                for l in range(len(self.final_processing_arch) - 1, -1, -1):
                    g[f"fin{l}_wgrad"].run(lib, st)
        else:
            check(lib.dib_loss_rows(LOSS_BCE_LOGITS, _ptr(ws, off["pred"]), self.output_dimensionality, _ptr(y), y.stride(0), B, inv,
                                    _ptr(ws, off["g_pred"]), _ptr(ws, off["out3"]), _ptr(ws, off["loss_ws"]), st), "dib_loss_rows")
            # head (deferred weight gradients: the dgrad chain first, then ONE grouped launch for the head's weight gradients)
            if not dw:
                g["out_wgrad"].run(lib, st)
            g["out_dgrad"].run(lib, st)
            for l in range(len(self.final_processing_arch) - 1, -1, -1):
                if not dw:
                    g[f"fin{l}_wgrad"].run(lib, st)
                g[f"fin{l}_dgrad"].run(lib, st)
            if dw:
                g["head_wgrads"].run(lib, st)
        check(lib.dib_mean_pool_bwd(_ptr(ws, off["g_pool"]), B, P, D, _ptr(ws, off["g_x"]), st), "dib_mean_pool_bwd")
        scale = 1.0 / math.sqrt(self.key_dim)
        nff = len(self.ff_arch_per_block)
        for b in range(self.number_attention_blocks - 1, -1, -1):
            pre = f"blk{b}_"
            gin, gout = self._block_grad_names(b)
            if pl["chain"]:
                # LN2 backward -> feed-forward dgrads -> LN1 backward -> output-projection dgrad in one launch; then the attention
                # backward, ONE grouped launch for the block's weight gradients, and the projections' dgrads (added to gout)
                defer = bool(pl["deferred_wgrads"])   # the block's weight-gradient operands stay in buffers of its own
                gb = (lambda nm: f"b{b}_{nm}") if defer else (lambda nm: nm)
                ffp = (c_void_p * 3)(*[_ptr(ws, off[f"b{b}_ff{l}"]) for l in range(nff)])
                gfp = (c_void_p * 3)(*[_ptr(ws, off[gb("g_z" if l == nff - 1 else f"g_ff{l}")]) for l in range(nff)])
                # deferred: the gradient of this block's output is [LN1-addend gradient | q/k/v input-gradient slabs] of the
                # NEXT block, summed by the kernel as it loads its tile; the last block's comes from the pooling backward
                from_slabs = defer and b + 1 < self.number_attention_blocks
                check(lib.dib_st_chain_bwd(ctypes.byref(pl["chain"][b]), T, _ptr(self.params),
                                           _ptr(ws, off[f"b{b + 1}_dx"] if from_slabs else off[gin]),
                                           (1 + (H if pl["attn_bwd_proj"] else 3 * pl["ksplit"])) if from_slabs else 1,
                                           T * D if from_slabs else 0,
                                           _ptr(ws, off[f"b{b}_xhat2"]), _ptr(ws, off[f"b{b}_rstd2"]), ffp,
                                           _ptr(ws, off[f"b{b}_xhat1"]), _ptr(ws, off[f"b{b}_rstd1"]), gfp,
                                           _ptr(ws, off[f"b{b}_gln1" if defer else gout]),
                                           _ptr(ws, off["g_ctx"]), _ptr(gt), _ptr(ws, off["chain_ws"]), st), "dib_st_chain_bwd")
                if not defer:
                    g[f"b{b}_ff_wgrad"].run(lib, st)
                    g[f"b{b}_o_wgrad"].run(lib, st)
                if pl["attn_bwd_proj"]:
                    # dq, dk, dv AND the head's share of the projections' input gradient (slab 1 + head of the block's dx region)
                    HK = H * self.key_dim
                    check(lib.dib_attention_bwd_proj(_ptr(ws, off[f"b{b}_q"]), _ptr(ws, off[f"b{b}_k"]), _ptr(ws, off[f"b{b}_v"]),
                                                     _ptr(ws, off["g_ctx"]), _ptr(ws, off[f"b{b}_lse"]), B, P, H, self.key_dim, D, HK, scale,
                                                     _ptr(ws, off[gb("g_q")]), _ptr(ws, off[gb("g_k")]), _ptr(ws, off[gb("g_v")]),
                                                     _ptr(self.params), pl["qkv_off"][b][0], _ptr(ws, off[f"b{b}_dx"]), T * D, st),
                          "dib_attention_bwd_proj")
                    if b == 0:   # block 0's total goes to the buffer the bottleneck's backward reads
                        check(lib.dib_reduce_splits(_ptr(ws, off["b0_dx"]), T * D, 1 + H, T * D, _ptr(ws, off[gout]), st),
                              "dib_reduce_splits")
                    continue
                self._attention_backward(pl, b, B, P, H, scale, *(gb(f"g_{nm}") for nm in "qkv"))
                if not defer:
                    g[f"b{b}_qkv_wgrad"].run(lib, st)
                g[f"b{b}_qkv_dgrad"].run(lib, st)
                if pl["ksplit"] == 1:
                    for nm in "qkv":
                        check(lib.dib_add_inplace(_ptr(ws, off[gout]), _ptr(ws, off[f"g_x{nm}"]), T * D, st), "dib_add_inplace")
                continue
            # x_out = LN2(h + ff): gin -> g_a (gradient of both addends) and, in the same pass, g_z = g_a * relu'(ff output)
            # (the feed-forward branch's pre-activation gradient), d(gamma2, beta2)
            check(lib.dib_add_layernorm_bwd_fused(_ptr(ws, off[gin]), c_void_p(0), _ptr(ws, off[f"b{b}_xhat2"]),
                                                  _ptr(ws, off[f"b{b}_rstd2"]), _ptr(self.params, self.offsets[pre + "ln2_g"]), T, D,
                                                  _ptr(ws, off["g_a"]), _ptr(ws, off[f"b{b}_ff{nff - 1}"]), ACT_RELU,
                                                  _ptr(ws, off["g_z"]), _ptr(gt, self.offsets[pre + "ln2_g"]), _ptr(ws, off["ln_ws"]),
                                                  st), "dib_add_layernorm_bwd_fused")
            for l in range(nff - 1, -1, -1):
                g[f"b{b}_ff{l}_wgrad"].run(lib, st)
                g[f"b{b}_ff{l}_dgrad"].run(lib, st)
            # h = LN1(x + mha): (g_h + g_a, the residual) -> gout, d(gamma1, beta1)
            check(lib.dib_add_layernorm_bwd_fused(_ptr(ws, off["g_h"]), _ptr(ws, off["g_a"]), _ptr(ws, off[f"b{b}_xhat1"]),
                                                  _ptr(ws, off[f"b{b}_rstd1"]), _ptr(self.params, self.offsets[pre + "ln1_g"]), T, D,
                                                  _ptr(ws, off[gout]), c_void_p(0), 0, c_void_p(0),
                                                  _ptr(gt, self.offsets[pre + "ln1_g"]), _ptr(ws, off["ln_ws"]), st),
                  "dib_add_layernorm_bwd_fused")
            # multi-head attention
            g[f"b{b}_o_wgrad"].run(lib, st)
            g[f"b{b}_o_dgrad"].run(lib, st)
            self._attention_backward(pl, b, B, P, H, scale)
            g[f"b{b}_qkv_wgrad"].run(lib, st)
            g[f"b{b}_qkv_dgrad"].run(lib, st)
            # gradient w.r.t. the block's input = residual (already in gout) + the three projection inputs
            if pl["ksplit"] == 1:   # (split-K: the slab reduce ADDS the three projections' gradients to gout itself)
                check(lib.dib_add_inplace(_ptr(ws, off[gout]), _ptr(ws, off["g_xq"]), T * D, st), "dib_add_inplace")
                check(lib.dib_add_inplace(_ptr(ws, off[gout]), _ptr(ws, off["g_xk"]), T * D, st), "dib_add_inplace")
                check(lib.dib_add_inplace(_ptr(ws, off[gout]), _ptr(ws, off["g_xv"]), T * D, st), "dib_add_inplace")
        # bottleneck: d(mu | raw logvar), beta * KL included
        ne = len(pl["enc_units"])
        # eps * sigma = x0 - mu: the gradient of the forward that ran (library noise, embs_reparam or deterministic)
        g_u = "g_x" if self.number_attention_blocks % 2 == 0 else "g_s"   # where the last processed block left dL/d(x0)
        check(lib.dib_token_reparam_kl_bwd(_ptr(ws, off[f"enc_h{ne - 1}"]), _ptr(ws, off[g_u]), _ptr(ws, off["x0"]), T, D,
                                           self.logvar_initialization, _ptr(self.beta_dev), inv,
                                           _ptr(ws, off[f"g_enc_h{ne - 1}"]), st), "dib_token_reparam_kl_bwd")
        em = pl["enc_mlp"]
        if em is not None:   # the encoder's dgrad chain in one launch (its weight gradients: grouped, on the stashes)
            check(lib.dib_mlp_small_bwd(ctypes.byref(em["desc"]), _ptr(self.params), _ptr(ws, off[f"g_enc_h{ne - 1}"]), em["h"], em["g"],
                                        T, st), "dib_mlp_small_bwd")
        for l in range(ne - 1, -1, -1):
            if not dw:   # (deferred: three more groups of the feed-forward class's launch below)
                g[f"enc{l}_wgrad"].run(lib, st)
            if l > 0 and em is None:
                g[f"enc{l}_dgrad"].run(lib, st)
        for name in pl["deferred_wgrads"]:   # every block's weight gradients, one grouped launch per shape class
            g[name].run(lib, st)
        self._unreduced = pl if (pl["nsplit"] > 1 and not reduce) else None
        if pl["nsplit"] > 1 and reduce:
            check(lib.dib_reduce_splits(_ptr(pl["slabs"]), self.n_alloc, pl["nsplit"], self.n_alloc, _ptr(self.grads), st),
                  "dib_reduce_splits")
        out3 = self._view(pl, "out3", 3)
        self.last["bce"] = out3[2:3] if hm is not None else out3[0:1] * inv   # (the head step writes loss sum * inv itself)
        self.last["correct"] = out3[1:2]

    def _attention_backward(self, pl, b: int, B: int, P: int, H: int, scale: float, gq: str = "g_q", gk: str = "g_k",
                            gv: str = "g_v") -> None:
        """g_ctx -> g_q, g_k, g_v of block b (flash kernels, or the grouped-GEMM products with the stashed probabilities);
        gq / gk / gv name the workspace buffers that receive them (flash path)."""
        lib, st, ws, off, g = self.lib, self._stream(), pl["ws"], pl["off"], pl["g"]
        if pl["impl"] == "gemm":
            g[f"b{b}_dv"].run(lib, st)
            g[f"b{b}_dp"].run(lib, st)
            check(lib.dib_softmax_rows_bwd(_ptr(ws, off[f"b{b}_S"]), _ptr(ws, off["g_S"]), B * H * P, P, pl["ldS"], scale, st),
                  "dib_softmax_rows_bwd")
            g[f"b{b}_dq"].run(lib, st)
            g[f"b{b}_dk"].run(lib, st)
        else:
            HK = H * self.key_dim
            check(lib.dib_attention_bwd(_ptr(ws, off[f"b{b}_q"]), _ptr(ws, off[f"b{b}_k"]), _ptr(ws, off[f"b{b}_v"]),
                                        _ptr(ws, off[f"b{b}_ctx"]), _ptr(ws, off["g_ctx"]), _ptr(ws, off[f"b{b}_lse"]),
                                        _ptr(pl["stash"][b]) if self.last.get("stash") else c_void_p(0),   # as the forward ran
                                        B, P, H, self.key_dim, HK, scale, _ptr(ws, off[gq]), _ptr(ws, off[gk]), _ptr(ws, off[gv]),
                                        _ptr(ws, off["attn_delta"]), st), "dib_attention_bwd")

    def adam_step(self, beta_1=0.9, beta_2=0.999, epsilon=1e-7, fused_reduce: bool = False) -> None:
        if fused_reduce and self.n_alloc % 4 == 0:   # slab reduce + Keras-Adam + step-count bump in one launch
            pl = getattr(self, "_unreduced", None)
            if getattr(self, "_sync", None) is None:
                self._sync = torch.zeros(_lib.SYNC_WORDS, dtype=torch.int32, device=self.device)
            check(self.lib.dib_reduce_adam_step(_ptr(pl["slabs"]) if pl is not None else None, pl["nsplit"] if pl is not None else 0,
                                                self.n_alloc, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                                self.n_alloc, _ptr(self.lr_dev), _ptr(self.t_dev), beta_1, beta_2, epsilon, 1.0,
                                                _ptr(self._sync), self._stream()), "dib_reduce_adam_step")
            self._unreduced = None
            return
        pl = getattr(self, "_unreduced", None)
        if pl is not None:   # (a deferred reduce that the fused path could not take)
            check(self.lib.dib_reduce_splits(_ptr(pl["slabs"]), self.n_alloc, pl["nsplit"], self.n_alloc, _ptr(self.grads),
                                             self._stream()), "dib_reduce_splits")
            self._unreduced = None
        check(self.lib.dib_adam_step(_ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v), self.n_alloc,
                                     _ptr(self.lr_dev), _ptr(self.t_dev), beta_1, beta_2, epsilon, 1.0, self._stream()),
              "dib_adam_step")

    def train_step(self, batch_inp, is_loci, training: bool = True):
        """The notebook's `train_step(batch_inp, is_loci, training=True)`: returns bce_losses (device scalar tensor [1]).

        Data parallel (BASELINE config 5 shards the neighbourhoods over the GPUs of a node): when torch.distributed is
        initialised every rank calls train_step with the SAME global batch; rank r takes neighbourhoods
        [r*B/N, (r+1)*B/N), the noise is keyed by the global token index (results independent of N), the flat gradient
        buffer is all-reduced (RCCL over xGMI with the nccl backend), every rank applies the same Adam update."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _FORCE_DP_BRANCH):
            if training and self.use_graphs:
                return self._train_step_graph(batch_inp, is_loci)
            self.forward(batch_inp, for_backward=training, _skip_head=training)
            self.loss_and_backward(is_loci, reduce=False) if training else self._loss_only(is_loci)
            if training:
                self.adam_step(fused_reduce=True)
            self._step += 1
            return self.last["bce"]
        rank, world = dist.get_rank(), dist.get_world_size()
        B, P = int(batch_inp.shape[0]), int(batch_inp.shape[1])
        lo, hi = (B * rank) // world, (B * (rank + 1)) // world
        stats = torch.zeros(2, dtype=self.params.dtype, device=self.device)   # [bce sum / B, kl sum / B] of the local rows
        if hi > lo:
            self.forward(batch_inp[lo:hi], row0=lo * P, for_backward=training, _skip_head=training)
            if training:
                self.loss_and_backward(is_loci[lo:hi], inv_global_batch=1.0 / B)
            else:
                self._loss_only(is_loci[lo:hi], inv_global_batch=1.0 / B)
            stats[0:1] = self.last["bce"]
            stats[1:2] = self.last["kl"] * ((hi - lo) / B)
        elif training:
            self.grads.zero_()
        if training:
            dist.all_reduce(self.grads)        # every rank issues the same collectives, rows or no rows
        dist.all_reduce(stats)
        if training:
            self.adam_step()
        self._step += 1
        self.last["bce"], self.last["kl"] = stats[0:1], stats[1:2]
        return self.last["bce"]

    def _set_step_dev(self, value: int) -> None:
        v = int(value) & 0xFFFFFFFF
        self.step_dev.fill_(v - (1 << 32) if v >= (1 << 31) else v)   # uint32 bit pattern in an int32 tensor

    def _train_step_graph(self, batch_inp, is_loci):
        """train_step as one hipGraph replay: inputs are copied into fixed staging buffers, everything else the step reads
        that changes between replays (beta, learning rate, Adam t, the noise step) already lives in device memory.  The
        eager warm-up runs exactly the captured sequence (module loads / hipFuncSetAttribute outside the capture) and every
        piece of state it touches is restored.  Same launches in the same order as the eager step: bit-identical results."""
        x = batch_inp if isinstance(batch_inp, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(batch_inp, dtype=np.float32))
        y = is_loci if isinstance(is_loci, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(is_loci, dtype=np.float32))
        B, P = int(x.shape[0]), int(x.shape[1])
        g = self._graphs.get((B, P))
        if g is None:
            xs = torch.zeros((B, P, self.particle_feature_dimensions), dtype=torch.float32, device=self.device)
            ys = torch.zeros((B, self.output_dimensionality), dtype=torch.float32, device=self.device)
            xs.copy_(x.to(self.device).reshape(xs.shape))
            ys.copy_(y.to(self.device).reshape(ys.shape))
            while len(self._graphs) >= self.max_graphs:   # oldest captured shape first: its plan (and stash) becomes evictable
                torch.cuda.synchronize(self.device)      # no replay may be in flight when a graph object dies
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[(B, P)] = {}          # pins the plan against LRU eviction from here on
            try:
                self._plan(B, P)
                state = (self.params, self.adam_m, self.adam_v, self.t_dev, self.step_dev, self.grads)
                saved = [t.clone() for t in state]

                def body():
                    self.forward(xs, step=0, _step_from_device=True, _skip_head=True)
                    self.loss_and_backward(ys, reduce=False)
                    self.adam_step(fused_reduce=True)

                self._set_step_dev(self._step)
                body()
                for t, v in zip(state, saved):
                    t.copy_(v)
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    body()
            except Exception:
                self._graphs.pop((B, P), None)   # no half-built entry: the next call starts over (or the caller goes eager)
                raise
            g = self._graphs[(B, P)] = dict(graph=graph, xs=xs, ys=ys, last=dict(self.last))
        else:
            g["xs"].copy_(x.to(self.device).reshape(g["xs"].shape))
            g["ys"].copy_(y.to(self.device).reshape(g["ys"].shape))
        self._set_step_dev(self._step)
        g["graph"].replay()
        self._step += 1
        self.last = dict(g["last"], step=self._step - 1)   # kl / bce / correct: tensors of the graph's pool, rewritten by every replay
        return self.last["bce"]

    def _loss_only(self, is_loci, inv_global_batch: Optional[float] = None):
        pl = self.last["plan"]
        B = self.last["B"]
        inv = 1.0 / B if inv_global_batch is None else float(inv_global_batch)
        y = is_loci if isinstance(is_loci, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(is_loci, dtype=np.float32))
        y = y.to(device=self.device, dtype=torch.float32).reshape(B, -1).contiguous()
        ws, off = pl["ws"], pl["off"]
        check(self.lib.dib_loss_rows(LOSS_BCE_LOGITS, _ptr(ws, off["pred"]), self.output_dimensionality, _ptr(y), y.stride(0), B,
                                     inv, _ptr(ws, off["g_pred"]), _ptr(ws, off["out3"]), _ptr(ws, off["loss_ws"]),
                                     self._stream()), "dib_loss_rows")
        out3 = self._view(pl, "out3", 3)
        self.last["bce"] = out3[0:1] * inv
        self.last["correct"] = out3[1:2]

    # ---- the notebook's evaluation helpers ---------------------------------------------------------------------------------
    def particle_encoder(self, feats) -> torch.Tensor:
        """particle_encoder(features): [..., particle_feature_dimensions] -> [..., 2 * bottleneck] (mu | raw logvar)."""
        x = feats if isinstance(feats, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32))
        x = x.to(device=self.device, dtype=torch.float32)
        lead = x.shape[:-1]
        x = x.reshape(-1, self.particle_feature_dimensions).contiguous()
        T, F0 = x.shape[0], self.particle_feature_dimensions
        pl = self._encoder_plan(T)
        lib, st, ws, off = self.lib, self._stream(), pl["ws"], pl["off"]
        ws[off["feats"]: off["feats"] + T * F0].view(T, F0).copy_(x)
        check(lib.dib_positional_encoding(_ptr(ws, off["feats"]), F0, T, F0, self.number_positional_encoding_frequencies,
                                          _ptr(ws, off["pe"]), st), "dib_positional_encoding")
        for gg in pl["g"]:
            gg.run(lib, st)
        E2 = 2 * self.bottleneck_dimension
        o = off[f"enc_h{len(pl['g']) - 1}"]
        return ws[o: o + T * E2].view(T, E2).clone().view(*lead, E2)

    def _encoder_plan(self, T: int) -> dict:
        """Encoder-only workspace + GEMM descriptors for `particle_encoder` on T particles (no attention buffers)."""
        key = ("enc", T)
        if key in self._plans:
            return self._plans[key]
        F0 = self.particle_feature_dimensions
        pe_w = F0 * self.number_positional_encoding_frequencies
        units = self.particle_encoder_arch_spec + [2 * self.bottleneck_dimension]
        off, o = {}, 0
        for name, n in [("feats", T * F0), ("pe", T * pe_w)] + [(f"enc_h{l}", T * u) for l, u in enumerate(units)]:
            off[name] = o
            o = _align4(o + n)
        ws = torch.zeros(o, dtype=torch.float32, device=self.device)
        gs, kin, src = [], pe_w, "pe"
        for l, u in enumerate(units):
            act = ACT_LEAKY01 if l < len(units) - 1 else ACT_NONE
            gs.append(_Gemm(0, [_d(off[src], kin, self.offsets[f"enc{l}_w"], u, off[f"enc_h{l}"], u, T, u, kin,
                                   bias_off=self.offsets[f"enc{l}_b"])], ws, self.params, ws, bias=self.params, act=act))
            kin, src = u, f"enc_h{l}"
        for gg in gs:
            gg.upload(self.device)
        # keep at most a few encoder plans (evaluation batches come in a handful of sizes)
        enc_keys = [k for k in self._plans if k[0] == "enc"]
        if len(enc_keys) >= 4:
            self._plans.pop(enc_keys[0])
        self._plans[key] = dict(ws=ws, off=off, g=gs)
        return self._plans[key]

    def probe_info_bounds(self, probe_features, data_features, seed: int = 0, step: int = 0, return_samples: bool = False):
        """One pass of the notebook's probe-grid estimator: per-probe (infonce_per, loo_per) in nats for probe particles
        `probe_features` [M, particle_feature_dimensions] against the data particles `data_features`
        [N, particle_feature_dimensions] (both encoded by `particle_encoder`, logvar - 3), on the device in float64."""
        ep = self.particle_encoder(probe_features).reshape(-1, 2 * self.bottleneck_dimension).contiguous()
        ed = self.particle_encoder(data_features).reshape(-1, 2 * self.bottleneck_dimension).contiguous()
        M, N, E = ep.shape[0], ed.shape[0], self.bottleneck_dimension
        ws = torch.empty(int(self.lib.dib_mi_probe_workspace_bytes(M, N, E)) // 8, dtype=torch.float64, device=self.device)
        rows = torch.empty((2, M), dtype=torch.float64, device=self.device)
        u = torch.empty((M, E), dtype=torch.float64, device=self.device) if return_samples else None
        check(self.lib.dib_mi_probe_bounds(_ptr8(ep), M, _ptr8(ed), N, E, self.logvar_initialization, int(seed),
                                           int(step) & 0xFFFFFFFF, 0, _ptr8(rows[0]), _ptr8(rows[1]), _ptr8(u), _ptr8(ws),
                                           self._stream()), "dib_mi_probe_bounds")
        return (rows[0], rows[1], u, ep, ed) if return_samples else (rows[0], rows[1])

    def information_map(self, particle_positions_probe, type_id: int, particle_features_val, num_eval_batches: int = 16,
                        eval_batch_size_probe_grid: int = 512, number_probes_to_eval_at_a_time: int = 100, seed: int = 0):
        """The notebook's per-particle information map for one particle type: for every probe position on the grid, the
        average over `num_eval_batches` random data batches (eval_batch_size_probe_grid neighbourhoods each, all their
        particles) of the lower / upper MI bounds.  Returns info_bounds_grid [n_probes, 2] (nats)."""
        pos = np.asarray(particle_positions_probe, dtype=np.float32)
        types = (type_id + 1) * np.ones(pos.shape[0], dtype=np.float32)
        features = convert_to_per_particle_feature_set(pos, types, number_particles_to_use=-1)
        xv = np.asarray(particle_features_val, dtype=np.float32)
        rng = np.random.default_rng(seed)
        lo_acc = torch.zeros(pos.shape[0], dtype=torch.float64, device=self.device)
        up_acc = torch.zeros_like(lo_acc)
        for probe_ind_start in range(0, pos.shape[0], number_probes_to_eval_at_a_time):
            sl = slice(probe_ind_start, min(pos.shape[0], probe_ind_start + number_probes_to_eval_at_a_time))
            for b in range(num_eval_batches):
                batch_inds = rng.choice(xv.shape[0], size=eval_batch_size_probe_grid, replace=True)
                batch_particles = xv[batch_inds].reshape(-1, self.particle_feature_dimensions)
                lo, up = self.probe_info_bounds(features[sl], batch_particles, seed=seed, step=probe_ind_start * 131 + b)
                lo_acc[sl] += lo
                up_acc[sl] += up
        return torch.stack([lo_acc, up_acc], -1).cpu().numpy() / num_eval_batches

    # ---- the notebook's training loop ------------------------------------------------------------------------------------
    @staticmethod
    def learning_rate_schedule(step: int, learning_rate: float, number_training_steps: int) -> float:
        """min(step / number_linear_ramp_lr_steps, 1) * learning_rate, number_linear_ramp_lr_steps = number_training_steps // 10."""
        return min(step / max(number_training_steps // 10, 1), 1) * learning_rate

    @staticmethod
    def beta_schedule(step: int, beta_start: float, beta_end: float, number_training_steps: int) -> float:
        """np.exp(np.log(beta_start) + float(step) / number_training_steps * (np.log(beta_end) - np.log(beta_start)))."""
        return float(np.exp(np.log(beta_start) + float(step) / number_training_steps * (np.log(beta_end) - np.log(beta_start))))

    def fit(self, particle_features_train, loci_train, number_training_steps=25_000, learning_rate=1e-4, beta_start=2e-6,
            beta_end=2e-1, batch_size=32, particle_features_val=None, loci_val=None, eval_every=None, batch_seed=0,
            verbose=False):
        """The notebook's loop: per step ramp the learning rate, anneal beta, draw `batch_size` neighbourhoods with
        replacement, train_step; every `eval_every` steps evaluate BCE and accuracy (sign of the logit) on the validation
        neighbourhoods.  Returns dict(bce_series_val, acc_series_val, bce_series_train)."""
        xtr = torch.from_numpy(np.ascontiguousarray(particle_features_train, dtype=np.float32)).to(self.device)
        ytr = torch.from_numpy(np.ascontiguousarray(loci_train, dtype=np.float32).reshape(-1, 1)).to(self.device)
        rng = np.random.default_rng(batch_seed)
        eval_every = eval_every or max(number_training_steps // 200, 1)
        hist = dict(bce_series_val=[], acc_series_val=[], bce_series_train=[], eval_steps=[])
        for step in range(number_training_steps):
            self.lr_dev.fill_(self.learning_rate_schedule(step, learning_rate, number_training_steps))
            self.beta_dev.fill_(self.beta_schedule(step, beta_start, beta_end, number_training_steps))
            batch_inds = torch.from_numpy(rng.choice(xtr.shape[0], size=batch_size, replace=True)).to(self.device)
            bce = self.train_step(xtr[batch_inds], ytr[batch_inds])
            if step % eval_every == 0:
                hist["bce_series_train"].append(float(bce.item()))
                if particle_features_val is not None:
                    bces, right, n = [], 0.0, 0
                    xv = np.asarray(particle_features_val, dtype=np.float32)
                    yv = np.asarray(loci_val, dtype=np.float32).reshape(-1, 1)
                    for s0 in range(0, xv.shape[0], batch_size):
                        xb, yb = xv[s0: s0 + batch_size], yv[s0: s0 + batch_size]
                        bces.append(float(self.train_step(xb, yb, training=False).item()))
                        pred = self.forward(xb, for_backward=False).cpu().numpy()   # a second sampled pass, as in the notebook
                        right += float((np.sign(pred) == (yb * 2 - 1)).sum())
                        n += len(xb)
                    hist["bce_series_val"].append(float(np.mean(bces)))
                    hist["acc_series_val"].append(right / n)
                    hist["eval_steps"].append(step)
                    if verbose:
                        print(f"Step: {step}, acc : {right / n:.4f}")
        return hist
