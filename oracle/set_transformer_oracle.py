"""TEST INFRASTRUCTURE ONLY - CPU restatement of the per-particle Distributed-IB set transformer of the reference's
amorphous-plasticity notebook (BASELINE config 5): PyTorch-CPU float64 with autograd for the backward.
It is the checker of the HIP path (dib_amd/set_transformer.py, include/dib_st.h; tests/test_gpu_set_transformer.py) and
nothing in the product imports it.

Reference: complex_systems/InfoDecomp_Amorphous_plasticity_per_particle_measurements_and_set_transformer.ipynb, code
cell 8 ("Create the particle encoder and the set transformer" ... `train_step`), cell 6
(`convert_to_per_particle_feature_set`), cell 4 (`PositionalEncoding`).  Cited below as nb:<marker>.

  particle encoder (shared by all particles)   nb:"particle_encoder = tf.keras.Sequential(layers)"
      PositionalEncoding(2**arange(1,5)) on 12 features -> Dense(128, LeakyReLU(0.1)) x2 -> Dense(2*32)
  bottleneck                                     nb:"def train_step"
      mu | logvar = split; logvar += -3; u = mu + exp(logvar/2) * eps
      KL = mean_batch( sum_{particle, dim} 0.5 (mu^2 + e^logvar - logvar - 1) )
  set transformer (Lee et al. 2019 as built there) nb:"for block_num in range(number_attention_blocks)"
      6 x [ MHA(12 heads, key_dim 128)(x,x,x) -> Add -> LayerNorm -> FF(Dense(128,relu), Dense(32,relu)) -> Add -> LayerNorm ]
      mean over particles -> Dense(256, LeakyReLU(0.1)) -> Dense(1);  loss = BCE-from-logits + beta * KL

Keras semantics restated (parity unpinned against TensorFlow itself, as for the main oracle): Dense = act(x W[in,out] + b);
MultiHeadAttention = per-head query/key/value projections [dim, heads, key_dim] with biases, scores q.k / sqrt(key_dim),
softmax over keys, output projection [heads, key_dim, dim] with bias; LayerNormalization over the last axis with
epsilon 1e-3, gamma, beta.  The wiring (residual order, normalisation placement, pooling axis, KL axes, the -3 offset) is
pinned by executing the notebook's own model-building code on the NumPy stand-in (tests/golden/make_golden_set_transformer.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch

SAFETY_EPS = 1e-10  # nb cell 2


@dataclass
class SetTransformerSpec:
    particle_feature_dimensions: int = 12            # nb:"particle_feature_dimensions = 2 + 2 + 2 + 2 + 2 + 2"
    number_positional_encoding_frequencies: int = 5
    particle_encoder_arch_spec: List[int] = field(default_factory=lambda: [128, 128])
    bottleneck_dimension: int = 32
    key_dim: int = 128
    number_heads_per_mha: int = 12
    number_attention_blocks: int = 6
    ff_arch_per_block: List[int] = field(default_factory=lambda: [128, 32])     # [128]*1 + [bottleneck_dimension]
    final_processing_arch: List[int] = field(default_factory=lambda: [256])
    output_dimensionality: int = 1
    leaky_slope: float = 0.1                          # tf.keras.layers.LeakyReLU(0.1)
    logvar_initialization: float = -3.0
    layer_norm_epsilon: float = 1e-3                  # Keras LayerNormalization default

    @property
    def frequencies(self):
        return [2.0 ** k for k in range(1, self.number_positional_encoding_frequencies)]

    @property
    def encoder_input_dim(self):
        return self.particle_feature_dimensions * self.number_positional_encoding_frequencies


def convert_to_per_particle_feature_set(particle_positions, types, number_particles_to_use=60):
    """nb cell 6: 12 features per particle (x, x^2, r, log r, log x^2, unit vector, one-hot type), nearest particles first."""
    pos = np.asarray(particle_positions, dtype=np.float32)
    types = np.asarray(types).astype(np.int32)
    one_hot = np.eye(2, dtype=np.float32)[types - 1]
    radii = np.sqrt(np.sum(np.square(pos), -1, keepdims=True) + np.float32(SAFETY_EPS)).astype(np.float32)
    unit = pos / radii
    feats = np.concatenate([pos, pos ** 2, radii, np.log(radii + np.float32(1e-3)), np.log(pos ** 2 + np.float32(1e-3)),
                            unit, one_hot], -1).astype(np.float32)
    if number_particles_to_use > 0:
        order = np.argsort(np.squeeze(radii, -1), kind="stable")
        feats = feats[order][:number_particles_to_use]
    return feats


def param_shapes(spec: SetTransformerSpec) -> Dict[str, tuple]:
    """Keras creation order = flat order.  Kernels [in, out]; MHA kernels [dim, heads, key_dim] / [heads, key_dim, dim]."""
    s: Dict[str, tuple] = {}
    d_in = spec.encoder_input_dim
    for l, u in enumerate(spec.particle_encoder_arch_spec + [2 * spec.bottleneck_dimension]):
        s[f"enc{l}_w"], s[f"enc{l}_b"] = (d_in, u), (u,)
        d_in = u
    D, H, K = spec.bottleneck_dimension, spec.number_heads_per_mha, spec.key_dim
    for b in range(spec.number_attention_blocks):
        for nm in ("q", "k", "v"):
            s[f"blk{b}_{nm}_w"], s[f"blk{b}_{nm}_b"] = (D, H, K), (H, K)
        s[f"blk{b}_o_w"], s[f"blk{b}_o_b"] = (H, K, D), (D,)
        s[f"blk{b}_ln1_g"], s[f"blk{b}_ln1_b"] = (D,), (D,)
        d = D
        for l, u in enumerate(spec.ff_arch_per_block):
            s[f"blk{b}_ff{l}_w"], s[f"blk{b}_ff{l}_b"] = (d, u), (u,)
            d = u
        s[f"blk{b}_ln2_g"], s[f"blk{b}_ln2_b"] = (D,), (D,)
    d = D
    for l, u in enumerate(spec.final_processing_arch):
        s[f"fin{l}_w"], s[f"fin{l}_b"] = (d, u), (u,)
        d = u
    s["out_w"], s["out_b"] = (d, spec.output_dimensionality), (spec.output_dimensionality,)
    return s


def init_params(spec: SetTransformerSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Keras defaults: glorot-uniform kernels (fan_in / fan_out with receptive-field convention for the MHA einsum
    kernels), zero biases, LayerNorm gamma = 1, beta = 0."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in param_shapes(spec).items():
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
        p[name] = torch.tensor(a, dtype=torch.float64)
    return p


class _MaskedLeaky(torch.autograd.Function):
    """leaky-relu / relu (slope 0) whose backward uses a GIVEN subgradient choice instead of its own `z > 0` (same idea as
    oracle/dib_torch_cpu.py:_MaskedReLU): at 4096 particles a float32 device and this float64 restatement legitimately
    disagree about the sign of a few pre-activations that sit within round-off of 0; the full-size parity test compares
    gradients under the device's own choices and separately bounds how many choices differ and how close to 0 they sit."""

    @staticmethod
    def forward(ctx, z, mask, slope):
        ctx.save_for_backward(mask)
        ctx.slope = slope
        return torch.where(z > 0, z, slope * z)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * torch.where(mask, 1.0, ctx.slope).to(g.dtype), None, None


def _leaky(x, slope, masks=None, name=None, boundary=None):
    """act(x); with masks[name] given, the backward follows that mask and `boundary[name]` records
    (number of units where it differs from x > 0, largest |x| among them)."""
    if masks is None or name not in masks:
        return torch.where(x > 0, x, slope * x)
    m = torch.as_tensor(masks[name]).reshape(x.shape)
    if boundary is not None:
        diff = m != (x > 0)
        boundary[name] = (int(diff.sum()), float(x.detach().abs()[diff].max()) if bool(diff.any()) else 0.0)
    return _MaskedLeaky.apply(x, m, slope)


def _layer_norm(x, g, b, eps):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * g + b


def positional_encoding(x, frequencies):
    """nb cell 4 (same layer as models.py:22-23)."""
    return torch.cat([x] + [torch.sin(f * x) for f in frequencies], -1)


def particle_encoder(spec, p, feats, masks=None, boundary=None):
    """[..., 12] -> [..., 2*bottleneck] (mu | raw logvar)."""
    h = positional_encoding(feats, spec.frequencies)
    n = len(spec.particle_encoder_arch_spec)
    for l in range(n):
        h = _leaky(h @ p[f"enc{l}_w"] + p[f"enc{l}_b"], spec.leaky_slope, masks, f"enc{l}", boundary)
    return h @ p[f"enc{n}_w"] + p[f"enc{n}_b"]


def multi_head_attention(spec, p, b, x):
    """Keras MultiHeadAttention(heads, key_dim)(x, x, x): self-attention over the particle axis."""
    q = torch.einsum("bpd,dhk->bphk", x, p[f"blk{b}_q_w"]) + p[f"blk{b}_q_b"]
    k = torch.einsum("bpd,dhk->bphk", x, p[f"blk{b}_k_w"]) + p[f"blk{b}_k_b"]
    v = torch.einsum("bpd,dhk->bphk", x, p[f"blk{b}_v_w"]) + p[f"blk{b}_v_b"]
    scale = 1.0 / math.sqrt(spec.key_dim)
    # one (neighbourhood, head) at a time: the [P, P] score matrix of 4096 particles is 134 MB in float64, all 12 heads of
    # it with autograd's copies would be several GB; the arithmetic is the einsum("bphk,bqhk->bhpq") of the one-shot form
    heads = []
    for hh in range(q.shape[2]):
        scores = torch.einsum("bpk,bqk->bpq", q[:, :, hh, :] * scale, k[:, :, hh, :])
        attn = torch.softmax(scores, dim=-1)
        heads.append(torch.einsum("bpq,bqk->bpk", attn, v[:, :, hh, :]))
    ctx = torch.stack(heads, dim=2)
    return torch.einsum("bphk,hkd->bpd", ctx, p[f"blk{b}_o_w"]) + p[f"blk{b}_o_b"]


def set_transformer(spec, p, u, masks=None, boundary=None):
    """[B, P, bottleneck] -> [B, out] (logits)."""
    x = u
    for b in range(spec.number_attention_blocks):
        h = _layer_norm(x + multi_head_attention(spec, p, b, x), p[f"blk{b}_ln1_g"], p[f"blk{b}_ln1_b"],
                        spec.layer_norm_epsilon)
        ff = h
        for l in range(len(spec.ff_arch_per_block)):
            ff = _leaky(ff @ p[f"blk{b}_ff{l}_w"] + p[f"blk{b}_ff{l}_b"], 0.0, masks, f"b{b}_ff{l}", boundary)   # relu
        x = _layer_norm(h + ff, p[f"blk{b}_ln2_g"], p[f"blk{b}_ln2_b"], spec.layer_norm_epsilon)
    x = x.mean(dim=-2)                                        # nb:"x = tf.reduce_mean(x, axis=-2)"
    for l in range(len(spec.final_processing_arch)):
        x = _leaky(x @ p[f"fin{l}_w"] + p[f"fin{l}_b"], spec.leaky_slope, masks, f"fin{l}", boundary)
    return x @ p["out_w"] + p["out_b"]


def forward(spec, p, feats, eps, is_loci=None, beta=0.0, masks=None, boundary=None):
    """nb:"def train_step" forward part.  feats [B,P,12], eps [B,P,bottleneck] standard normal.  Returns dict.
    masks (optional): {"enc<l>", "b<b>_ff<l>", "fin<l>"} -> boolean act' choices for the backward (see _MaskedLeaky)."""
    feats = torch.as_tensor(feats, dtype=torch.float64)
    eps = torch.as_tensor(eps, dtype=torch.float64)
    enc = particle_encoder(spec, p, feats, masks, boundary)
    mu, logvar = enc[..., : spec.bottleneck_dimension], enc[..., spec.bottleneck_dimension:]
    logvar = logvar + spec.logvar_initialization
    u = mu + torch.exp(logvar / 2.0) * eps
    kl = (0.5 * (mu ** 2 + torch.exp(logvar) - logvar - 1.0)).sum(dim=(-1, -2)).mean()
    pred = set_transformer(spec, p, u, masks, boundary)
    out = dict(mu=mu, logvar=logvar, u=u, kl=kl, pred=pred)
    if is_loci is not None:
        y = torch.as_tensor(is_loci, dtype=torch.float64).reshape(pred.shape)
        # Keras BinaryCrossentropy(from_logits=True): mean over batch of max(z,0) - z*y + log(1 + exp(-|z|))
        bce = (torch.clamp(pred, min=0) - pred * y + torch.log1p(torch.exp(-pred.abs()))).mean()
        out["bce"] = bce
        out["loss"] = bce + beta * kl
    return out


def loss_and_grads(spec, p, feats, eps, is_loci, beta, masks=None, boundary=None):
    """d(bce + beta * KL)/d(params) by autograd (nb: tape.gradient(loss, all_trainable_variables))."""
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    out = forward(spec, q, feats, eps, is_loci, beta, masks, boundary)
    grads = torch.autograd.grad(out["loss"], list(q.values()))
    return {k: float(v.detach()) for k, v in out.items() if v.dim() == 0}, dict(zip(q.keys(), grads))


def learning_rate_schedule(step, learning_rate, number_training_steps):
    """nb:"Ramp the learning rate": linear warm-up over the first tenth of training."""
    ramp = number_training_steps // 10
    return min(step / ramp, 1) * learning_rate


def beta_schedule(step, beta_start, beta_end, number_training_steps):
    """nb:"Anneal beta": per-STEP log ramp in numpy float64 (not the per-epoch float32 callback of models.py)."""
    return float(np.exp(np.log(beta_start) + float(step) / number_training_steps * (np.log(beta_end) - np.log(beta_start))))


def flops_per_neighbourhood(spec: SetTransformerSpec, particles: int) -> int:
    """Algorithmic forward GEMM FLOPs for one neighbourhood of `particles` particles (planning number)."""
    fl = 0
    d = spec.encoder_input_dim
    for u in spec.particle_encoder_arch_spec + [2 * spec.bottleneck_dimension]:
        fl += 2 * d * u * particles
        d = u
    D, H, K = spec.bottleneck_dimension, spec.number_heads_per_mha, spec.key_dim
    per_block = 3 * 2 * D * H * K * particles + 2 * 2 * H * K * particles * particles + 2 * H * K * D * particles
    d = D
    for u in spec.ff_arch_per_block:
        per_block += 2 * d * u * particles
        d = u
    fl += spec.number_attention_blocks * per_block
    d = D
    for u in spec.final_processing_arch + [spec.output_dimensionality]:
        fl += 2 * d * u
        d = u
    return fl


def probe_info_bounds(mus_probes, logvars_probes, sampled_u_probes, mus_data, logvars_data):
    """Per-probe InfoNCE lower / leave-one-out upper bounds (nats) of the notebook's per-particle information map
    (nb: "Now use probe points along with a bunch of real points to get the info for points on a grid", inner loop):
      p_ii = N(u_i; mu_i, sigma_i) for the probe's own Gaussian, p_ij = N(u_i; mu_j, sigma_j) over the N data Gaussians,
      infonce_i = log(p_ii / mean([p_ii, p_i1 .. p_iN]))   (N + 1 terms),   loo_i = log(p_ii / mean_j p_ij)   (N terms).
    Literal float64 restatement (exp then log, like the notebook).  logvars already include the -3 offset."""
    mp, lp = np.asarray(mus_probes, np.float64), np.asarray(logvars_probes, np.float64)
    md, ld = np.asarray(mus_data, np.float64), np.asarray(logvars_data, np.float64)
    u = np.asarray(sampled_u_probes, np.float64)
    E = mp.shape[-1]
    norm = (2.0 * np.pi) ** (E / 2.0)
    p_ii = np.exp(-np.sum(((u - mp) / np.exp(lp / 2.0)) ** 2, -1) / 2.0 - np.sum(lp, -1) / 2.0) / norm
    d = (u[:, None, :] - md[None, :, :]) / np.exp(ld / 2.0)[None, :, :]
    p_ij = np.exp(-np.sum(d ** 2, -1) / 2.0 - np.sum(ld, -1)[None, :] / 2.0) / norm
    infonce = np.log(p_ii / np.mean(np.concatenate([p_ii[:, None], p_ij], -1), axis=1))
    loo = np.log(p_ii / np.mean(p_ij, axis=1))
    return infonce, loo
