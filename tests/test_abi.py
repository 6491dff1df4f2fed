"""CPU: the C-ABI library loads, exports every symbol include/dib_hip.h declares, and its host-only
entry points (layout, Philox reference) agree with the oracle.  No compute calls (no GPU here)."""
import ctypes
import os
import re
from ctypes import byref, c_int, c_int64, c_void_p

import numpy as np
import pytest

import dib_oracle as orc
from _helpers import SPECS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from dib_amd import _lib
    return _lib, _lib.load_library()


def test_header_symbols_are_exported_and_bound():
    mod, lib = _lib()
    hdr = open(os.path.join(ROOT, "include", "dib_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|int64_t|float|const char\*)\s+(dib_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in dib_hip.h but not exported"
    assert declared == set(mod.SIGNATURES), declared ^ set(mod.SIGNATURES)
    hdr_st = open(os.path.join(ROOT, "include", "dib_st.h")).read()
    declared_st = set(re.findall(r"^(?:int|void|int64_t|float|const char\*)\s+(dib_[a-z0-9_]+)\s*\(", hdr_st, re.M))
    assert len(declared_st) >= 12
    for name in declared_st:
        assert hasattr(lib, name), f"{name} declared in dib_st.h but not exported"
    assert declared_st == set(mod.SIGNATURES_ST), declared_st ^ set(mod.SIGNATURES_ST)
    assert b"gfx950" in lib.dib_version()
    assert lib.dib_error_string(-2) == b"shape mismatch"


def _create(lib, spec):
    from dib_amd._lib import ACTIVATIONS
    ci = lambda xs: (c_int * max(1, len(xs)))(*xs)
    h = c_void_p()
    rc = lib.dib_layout_create(spec.number_features, ci(list(spec.feature_dimensionalities)),
                               len(spec.feature_encoder_architecture), ci(list(spec.feature_encoder_architecture)),
                               spec.feature_embedding_dimension, len(spec.integration_network_architecture),
                               ci(list(spec.integration_network_architecture)), spec.output_dimensionality,
                               int(spec.use_positional_encoding), spec.number_positional_encoding_frequencies,
                               ACTIVATIONS[spec.activation_fn], ACTIVATIONS[spec.output_activation_fn], byref(h))
    return rc, h


@pytest.mark.parametrize("name", list(SPECS) + ["north_star"])
def test_layout_blocks_match_oracle_shapes(name):
    _, lib = _lib()
    spec = SPECS.get(name) or orc.DIBSpec([1] * 64, [128, 128], [256, 256], 1)
    rc, h = _create(lib, spec)
    assert rc == 0
    total = lib.dib_layout_param_count(h)
    assert spec.num_params() <= total <= spec.num_params() + 4 * (2 * (len(spec.feature_encoder_architecture) + 1) * (spec.number_features + 1) + 16)
    off, rows, cols = c_int64(), c_int(), c_int()
    seen = []
    for f in range(spec.number_features):
        for l, (i, o) in enumerate(spec.encoder_layer_dims(f)):
            assert lib.dib_layout_param_block(h, 0, l, f, 0, byref(off), byref(rows), byref(cols)) == 0
            assert (rows.value, cols.value) == (i, o) and off.value % 4 == 0
            seen.append((off.value, i * o))
            assert lib.dib_layout_param_block(h, 0, l, f, 1, byref(off), byref(rows), byref(cols)) == 0
            assert cols.value == o
            seen.append((off.value, o))
    for l, (i, o) in enumerate(spec.integration_layer_dims()):
        assert lib.dib_layout_param_block(h, 1, l, 0, 0, byref(off), byref(rows), byref(cols)) == 0
        assert (rows.value, cols.value) == (i, o)
        seen.append((off.value, i * o))
        assert lib.dib_layout_param_block(h, 1, l, 0, 1, byref(off), byref(rows), byref(cols)) == 0
        seen.append((off.value, o))
    seen.sort()
    for (o1, n1), (o2, _) in zip(seen, seen[1:]):
        assert o1 + n1 <= o2, "parameter blocks overlap"
    assert seen[-1][0] + seen[-1][1] <= total
    assert sum(n for _, n in seen) == spec.num_params()
    assert lib.dib_workspace_bytes(h, 128) > 0 and lib.dib_layout_table_bytes(h) > 0
    assert lib.dib_workspace_offset(h, 128, 0) >= 0
    assert lib.dib_layout_wgrad_splits(h, 65536) > 1 and lib.dib_layout_wgrad_splits(h, 128) == 1
    assert lib.dib_layout_param_block(h, 0, 99, 0, 0, byref(off), byref(rows), byref(cols)) == -1
    lib.dib_layout_destroy(h)


def test_layout_rejects_bad_arguments():
    _, lib = _lib()
    h = c_void_p()
    one = (c_int * 1)(1)
    assert lib.dib_layout_create(0, one, 0, one, 4, 0, one, 1, 1, 5, 1, 0, byref(h)) == -1
    assert lib.dib_layout_create(1, one, 0, one, 4, 0, one, 1, 1, 5, 99, 0, byref(h)) == -4
    assert lib.dib_encoder_bank_fwd(None, None, 0, None, 0, 0, None, 0, 0, 0, None, None) == -1


def test_host_philox_reference_matches_oracle():
    """dib_philox_normal_ref (C, float32) vs oracle (numpy, float64) from the same Philox bits."""
    _, lib = _lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed = int(rng.integers(0, 2 ** 63))
        step, row, f, e = [int(v) for v in rng.integers(0, 2 ** 31, 4)]
        f, e = f % 64, e % 32
        got = lib.dib_philox_normal_ref(seed, step, row, f, e)
        want = orc.philox_normal(seed, step, np.array([row]), f, 32)[0, e]
        assert abs(got - want) < 2e-5 * (1 + abs(want)), (seed, step, row, f, e, got, want)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dib_amd
    m = dib_amd.DistributedIBNet([1, 1], [8], [8], 1)
    m.compile(optimizer="adam", loss=dib_amd.losses.BinaryCrossentropy(from_logits=True))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.fit(np.zeros((4, 2)), np.zeros(4), epochs=1, batch_size=2, verbose=False)
