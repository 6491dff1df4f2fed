"""Test helpers: oracle <-> flat-buffer mapping, spec zoo, the dispatch switches of the library."""
import contextlib

import numpy as np

import dib_oracle as orc


def spec_kwargs(spec: orc.DIBSpec):
    return dict(feature_dimensionalities=list(spec.feature_dimensionalities),
                feature_encoder_architecture=list(spec.feature_encoder_architecture),
                integration_network_architecture=list(spec.integration_network_architecture),
                output_dimensionality=spec.output_dimensionality,
                use_positional_encoding=spec.use_positional_encoding,
                number_positional_encoding_frequencies=spec.number_positional_encoding_frequencies,
                activation_fn=spec.activation_fn, feature_embedding_dimension=spec.feature_embedding_dimension,
                output_activation_fn=spec.output_activation_fn)


def flat_to_params(blocks, flat, spec: orc.DIBSpec, dtype=np.float64) -> orc.DIBParams:
    F = spec.number_features
    nle = len(spec.feature_encoder_architecture) + 1
    nli = len(spec.integration_network_architecture) + 1
    enc_W = [[None] * nle for _ in range(F)]
    enc_b = [[None] * nle for _ in range(F)]
    int_W, int_b = [None] * nli, [None] * nli
    for b in blocks:
        n = b["rows"] * b["cols"]
        v = np.asarray(flat[b["offset"]: b["offset"] + n], dtype=dtype)
        if b["net"] == 0:
            if b["what"] == 0:
                enc_W[b["feature"]][b["layer"]] = v.reshape(b["rows"], b["cols"]).copy()
            else:
                enc_b[b["feature"]][b["layer"]] = v.copy()
        else:
            if b["what"] == 0:
                int_W[b["layer"]] = v.reshape(b["rows"], b["cols"]).copy()
            else:
                int_b[b["layer"]] = v.copy()
    return orc.DIBParams(enc_W, enc_b, int_W, int_b)


def params_to_flat(blocks, params: orc.DIBParams, n_alloc: int, dtype=np.float32) -> np.ndarray:
    flat = np.zeros(n_alloc, dtype=dtype)
    for b in blocks:
        if b["net"] == 0:
            t = params.enc_W[b["feature"]][b["layer"]] if b["what"] == 0 else params.enc_b[b["feature"]][b["layer"]]
        else:
            t = params.int_W[b["layer"]] if b["what"] == 0 else params.int_b[b["layer"]]
        flat[b["offset"]: b["offset"] + t.size] = np.asarray(t, dtype=dtype).reshape(-1)
    return flat


def random_params(spec: orc.DIBSpec, seed=0, bias_scale=0.1) -> orc.DIBParams:
    p = orc.glorot_uniform_init(spec, seed)
    rng = np.random.default_rng(seed + 1000)
    for t in p.tensors():
        if t.ndim == 1:
            t[:] = bias_scale * rng.standard_normal(t.shape)
    return p


# a zoo of architectures: BASELINE configs at reduced size + ragged / odd shapes
SPECS = {
    "boolean4_32x32": orc.DIBSpec([1, 1, 1, 1], [32, 32], [64, 64], 1, feature_embedding_dimension=32),
    "pendulum_ragged": orc.DIBSpec([2, 1, 2, 1], [128, 128], [256, 256], 6, feature_embedding_dimension=32),
    "odd_shapes_tanh": orc.DIBSpec([3, 5, 1], [17, 9], [13], 3, activation_fn="tanh", feature_embedding_dimension=6,
                                   number_positional_encoding_frequencies=3),
    "no_posenc_leaky": orc.DIBSpec([4, 2], [24], [20, 12], 2, use_positional_encoding=False,
                                   activation_fn="leaky_relu", feature_embedding_dimension=8),
    "ib_single_feature": orc.DIBSpec([10], [64, 64], [32], 1, feature_embedding_dimension=16),
    "no_hidden": orc.DIBSpec([1, 1], [], [], 1, feature_embedding_dimension=4),
    "tabular8_default": orc.DIBSpec([1] * 8, [128, 128], [256, 256], 1, feature_embedding_dimension=32),
    "sigmoid_out_elu": orc.DIBSpec([2, 2], [16], [16], 1, activation_fn="elu", output_activation_fn="sigmoid",
                                   feature_embedding_dimension=4),
    # fused-kernel edge cases: leaky / linear activations, no positional encoding, encoder-input widths at the
    # fused-backward limit (15: fused fwd+bwd) and just above it (16: fused fwd + general GEMM bwd), fwd-only configs
    "fused_leaky": orc.DIBSpec([1, 2, 1], [32, 32], [24], 1, activation_fn="leaky_relu", feature_embedding_dimension=32),
    "fused_linear_act": orc.DIBSpec([1, 1], [32, 32], [16], 1, activation_fn=None, feature_embedding_dimension=32),
    "fused_no_posenc": orc.DIBSpec([3, 2, 5], [128, 128], [32], 1, use_positional_encoding=False,
                                   feature_embedding_dimension=32),
    "fused_in15": orc.DIBSpec([3, 1], [32, 32], [16], 1, feature_embedding_dimension=32),
    "fused_fwd_in16_gemm_bwd": orc.DIBSpec([4, 2], [32, 32], [16], 1, number_positional_encoding_frequencies=4,
                                           feature_embedding_dimension=32),
    "fused_fwd_only_e16": orc.DIBSpec([1, 1, 2], [64, 64], [32], 1, feature_embedding_dimension=16),
    "fused_fwd_only_e8": orc.DIBSpec([1, 1, 1, 1], [32, 32], [64], 1, feature_embedding_dimension=8),   # smoke()'s architecture
    # the non-relu template variants of the remaining fused instantiations (the Boolean notebook's networks are leaky_relu:
    # complex_systems/InfoDecomp_Boolean_circuits.ipynb:266-268; train.py:37 makes the activation a flag)
    "fused_128_leaky": orc.DIBSpec([1, 2, 1], [128, 128], [32], 1, activation_fn="leaky_relu", feature_embedding_dimension=32),
    "fused_fwd_only_e8_leaky": orc.DIBSpec([1, 1], [32, 32], [16], 1, activation_fn="leaky_relu", feature_embedding_dimension=8),
    "fused_fwd_only_e16_linear": orc.DIBSpec([2, 1], [64, 64], [16], 1, activation_fn=None, feature_embedding_dimension=16),
    # the fused 1-unit output head's wider variants (last integration layer of 257-512 / more than 512 units)
    "head_wide_384": orc.DIBSpec([1, 1], [32, 32], [384], 1, feature_embedding_dimension=32),
    "head_wide_640": orc.DIBSpec([1, 1], [32, 32], [64, 640], 1, feature_embedding_dimension=32),
}
# zoo entries one of the fused large-batch instantiations covers (csrc/dib_api.hip kFused: encoder = two hidden layers of
# (128,128,32) / (32,32,32) / forward-only (32,32,8), (64,64,16); inputs <= 16 wide; relu / leaky_relu / linear)
FUSED_ELIGIBLE = ("boolean4_32x32", "pendulum_ragged", "tabular8_default", "fused_leaky", "fused_linear_act", "fused_no_posenc",
                  "fused_in15", "fused_fwd_in16_gemm_bwd", "fused_fwd_only_e16", "fused_fwd_only_e8", "fused_128_leaky",
                  "fused_fwd_only_e8_leaky", "fused_fwd_only_e16_linear", "head_wide_384", "head_wide_640")

# The library picks its kernels by batch size and architecture (csrc/dib_api.hip: small_regime, fused_id).  Every test that
# compares with the float64 oracle runs on BOTH sides of every switch:
#   "default"      what a caller gets: row-tile kernels (csrc/dib_small.h) while ceil(B / 16) x F <= 512 and B <= 2048,
#                  otherwise the fused encoder-bank kernels where the architecture has an instantiation, otherwise grouped GEMMs
#   "large_batch"  dib_set_tuning("small_batch", 0): the large-batch kernels at every batch size
#   "grouped_gemm" ... and dib_set_tuning("fused_encoder", 0) for layouts created inside: the general grouped-GEMM path
#   "cluster_tiles" the row-tile integration kernel with every row tile on a cluster of 4 workgroups (csrc/dib_small.h "cluster
#                  mode") whatever the network's size: dib_set_tuning("int_cluster", 4), ("int_cluster_min_weights", 0) -
#                  "default" clusters (8 workgroups) only networks whose first integration layer has >= 32 768 weights, i.e. few zoo entries
DISPATCH_PATHS = ("default", "large_batch", "grouped_gemm", "cluster_tiles")


@contextlib.contextmanager
def dispatch_path(path):
    """Engines must be CREATED inside the context ("fused_encoder" is read by dib_layout_create)."""
    from dib_amd import _lib
    assert path in DISPATCH_PATHS, path
    old = {k: _lib.get_tuning(k) for k in ("small_batch", "fused_encoder", "int_cluster", "int_cluster_min_weights")}
    try:
        if path == "cluster_tiles":
            _lib.set_tuning("int_cluster", 4)
            _lib.set_tuning("int_cluster_min_weights", 0)
        elif path != "default":
            _lib.set_tuning("small_batch", 0)
        if path == "grouped_gemm":
            _lib.set_tuning("fused_encoder", 0)
        yield
    finally:
        for k, v in old.items():
            _lib.set_tuning(k, v)

