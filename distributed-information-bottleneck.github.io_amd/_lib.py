"""ctypes binding of libdib_hip.so (C ABI declared in include/dib_hip.h) + in-tree build helper.

The library is built IN-TREE with hipcc for gfx950 (`build_library`) so the .so travels with the
repo snapshot to the GPU box.  There is no CPU fallback: if the library cannot be loaded the
product path raises.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libdib_hip.so")
# Kernel A/B experiments (tools/ab_bench.sh) point DIB_LIB_PATH at a variant build (exp/lib_TAG.so); the product artefact is
# never overwritten.  A set-but-missing path is an error, not a silent return to the product library.
LIB_OVERRIDE = os.environ.get("DIB_LIB_PATH") or None
INCLUDE = os.path.join(os.path.dirname(_HERE), "include", "dib_hip.h")
INCLUDE_ST = os.path.join(os.path.dirname(_HERE), "include", "dib_st.h")
SOURCES = ["dib_api.hip", "dib_gemm.h", "dib_elementwise.h", "dib_common.h", "dib_fused.h", "dib_tail.h", "dib_small.h", "dib_st_chain.h", "dib_st.h", "dib_attn.h", "dib_attn_small.h", "dib_infonce_mfma.h",
           INCLUDE_ST]

# error codes (include/dib_hip.h)
DIB_OK = 0
ABI_VERSION = 6   # include/dib_hip.h DIB_ABI_VERSION this binding's SIGNATURES were written against
ACTIVATIONS = {None: 0, "linear": 0, "None": 0, "relu": 1, "leaky_relu": 2, "tanh": 3, "sigmoid": 4, "elu": 5,
               "softplus": 6}
ACT_LEAKY_RELU_01 = 7  # tf.keras.layers.LeakyReLU(0.1) (include/dib_st.h)
LOSS_KINDS = {"bce_logits": 0, "bce": 1, "sparse_cce_logits": 2, "mse": 3}
WS_U, WS_PRED, WS_ENC_OUT, WS_G_U, WS_STEP_OUT, WS_G_PRED = range(6)
WS_ENC_H0, WS_INT_H0 = 16, 32
# include/dib_hip.h flag bits
SYNC_WORDS = 1056   # DIB_SYNC_WORDS
FWD_DETERMINISTIC, FWD_INFERENCE, FWD_DEFER_SUMS = 1, 2, 4
HEAD_DEFER_SUMS, HEAD_NO_GRAD, HEAD_DEFER_WGRAD = 1, 2, 4
BWD_INTEGRATION_DONE = 1
TAIL_FINALIZE, TAIL_KL, TAIL_LOSS, TAIL_ADAM, TAIL_BUMP, TAIL_METRICS, TAIL_SGD, TAIL_HEAD_WGRAD, TAIL_LOSS_HEAD = \
    1, 2, 4, 8, 16, 32, 64, 128, 256
SIMILARITIES = {"l2sq": 0, "l2": 1, "l1": 2, "linf": 3, "cosine": 4}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libdib_hip.so")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for s in SOURCES + [INCLUDE]:
        p = s if os.path.isabs(s) else os.path.join(CSRC, s)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           os.path.join(CSRC, "dib_api.hip"), "-o", LIB_PATH + f".tmp{os.getpid()}"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB_PATH + f".tmp{os.getpid()}", LIB_PATH)  # per-process temp name: concurrent ranks cannot clobber each other
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); must list every symbol declared in include/dib_hip.h
SIGNATURES = {
    "dib_version": (c_char_p, []),
    "dib_abi_version": (c_int, []),
    "dib_error_string": (c_char_p, [c_int]),
    "dib_layout_create": (c_int, [c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int, POINTER(c_int), c_int,
                                  c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "dib_layout_destroy": (None, [c_void_p]),
    "dib_layout_param_count": (c_int64, [c_void_p]),
    "dib_layout_param_block": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int),
                                       POINTER(c_int)]),
    "dib_layout_table_bytes": (c_int64, [c_void_p]),
    "dib_layout_upload_tables": (c_int, [c_void_p, c_void_p, c_void_p]),
    "dib_layout_set_step_counter": (c_int, [c_void_p, c_void_p]),
    "dib_workspace_bytes": (c_int64, [c_void_p, c_int]),
    "dib_workspace_init": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "dib_workspace_offset": (c_int64, [c_void_p, c_int, c_int]),
    "dib_layout_wgrad_splits": (c_int, [c_void_p, c_int]),
    "dib_encoder_bank_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_uint64,
                                     c_uint32, c_int, c_void_p, c_void_p]),
    "dib_integration_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dib_loss_fwd_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_float, c_int, c_void_p,
                                 c_void_p]),
    "dib_integration_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_output_head_fused_supported": (c_int, [c_void_p, c_int]),
    "dib_integration_fwd_hidden": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dib_output_head_fused": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_float, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    "dib_integration_head_step": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_float, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    "dib_integration_bwd_hidden": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_backward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p]),
    "dib_encoder_bank_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "dib_encoder_bank_bwd_stage": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p]),
    "dib_grads_finalize": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dib_grads_finalize_part": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dib_layout_part_range": (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "dib_metrics_accumulate": (c_int, [c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "dib_step_tail": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                              c_float, c_float, c_float, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "dib_set_tuning": (c_int, [c_char_p, c_int]),
    "dib_get_tuning": (c_int, [c_char_p, POINTER(c_int)]),
    "dib_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_float,
                              c_float, c_float, c_void_p]),
    "dib_sgd_step": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_float, c_void_p]),
    "dib_encode_deterministic": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_bhattacharyya": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dib_infonce_workspace_bytes": (c_int64, [c_int]),
    "dib_infonce_fwd_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "dib_positional_encoding": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dib_positional_encoding_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dib_mlp_small_supported": (c_int, [c_void_p, c_int]),
    "dib_mlp_small_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_mlp_small_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dib_mlp_small_head_supported": (c_int, [c_void_p, c_int]),
    "dib_mlp_small_head_workspace_bytes": (c_int64, [c_void_p, c_int]),
    "dib_mlp_small_head_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_integration_fwd_and_mlp_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                                c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_backward_and_mlp_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dib_reduce_adam_step": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "dib_mi_workspace_bytes": (c_int64, [c_int, c_int]),
    "dib_mi_sandwich_rows": (c_int, [c_void_p, c_int, c_int, c_uint64, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "dib_philox_normal_fill": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_uint64, c_uint32,
                                       c_void_p]),
    "dib_philox_normal_ref": (c_float, [c_uint64, c_uint32, c_uint32, c_uint32, c_uint32]),
    "dib_launch_count": (c_int64, []),
    "dib_profile_enable": (c_int, [c_int]),
    "dib_profile_summary": (c_int, [POINTER(ctypes.c_double), POINTER(c_int)]),
    "dib_gemm": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                         c_void_p, c_int, c_int, c_void_p, c_void_p]),
}

# include/dib_st.h: building blocks of the per-particle set transformer
SIGNATURES_ST = {
    "dib_gemm_grouped": (c_int, [c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "dib_reduce_splits": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "dib_reduce_splits_add": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "dib_gemm_skinny_k": (c_int, [c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_softmax_rows_fwd": (c_int, [c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "dib_softmax_rows_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "dib_attention_stash_bytes": (c_int64, [c_int, c_int, c_int]),
    "dib_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_float, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "dib_attention_fwd_proj_supported": (c_int, [c_int, c_int, c_int]),
    "dib_attention_fwd_proj": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_attention_bwd_proj": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dib_attention_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "dib_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_add_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_float, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    "dib_add_layernorm_bwd_workspace_bytes": (c_int64, [c_int64, c_int]),
    "dib_add_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "dib_add_layernorm_bwd_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                            c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_st_chain_supported": (c_int, [c_void_p, c_int64]),
    "dib_st_chain_workspace_bytes": (c_int64, [c_int64, c_int]),
    "dib_st_chain_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "dib_st_chain_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_mean_pool_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dib_mean_pool_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dib_add_inplace": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "dib_act_grad_mul": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "dib_token_kl_workspace_bytes": (c_int64, [c_int64, c_int]),
    "dib_token_reparam_kl_fwd": (c_int, [c_void_p, c_int64, c_int, c_float, c_uint64, c_uint32, c_void_p, c_int64, c_int, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "dib_token_reparam_kl_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_float, c_void_p,
                                         c_void_p]),
    "dib_mi_probe_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "dib_mi_probe_bounds": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_uint64, c_uint32, c_uint32, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "dib_loss_rows_workspace_bytes": (c_int64, [c_int]),
    "dib_loss_rows": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p,
                              c_void_p]),
}


def load_library(build_if_missing: bool = True):
    """dlopen libdib_hip.so and attach signatures.  Raises (no fallback) if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if LIB_OVERRIDE is not None:
        if not os.path.exists(LIB_OVERRIDE):
            raise RuntimeError(f"DIB_LIB_PATH={LIB_OVERRIDE} does not exist")
        return _attach(ctypes.CDLL(os.path.abspath(LIB_OVERRIDE)))
    if build_if_missing and _stale():
        try:
            build_library()
        except Exception:
            if not os.path.exists(LIB_PATH):
                raise
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    return _attach(ctypes.CDLL(LIB_PATH))


def _attach(lib):
    global _lib
    try:
        lib.dib_abi_version.restype = c_int
        have = int(lib.dib_abi_version())
    except AttributeError:
        have = None
    # A library of another ABI revision is refused outright: a stale variant build called with shifted arguments would corrupt
    # memory silently.  (Rounds 4-5 had a lenient mode, DIB_LIB_ABI_CHECK=0, that bound only the entry points unchanged since
    # ABI 3 for same-box A/Bs against an older round's library; since ABI 5 every step goes through entry points that
    # did not exist then, so the mode could no longer run a step and was removed - tools/runs/r04g.sh is a historical record.)
    if have != ABI_VERSION:
        raise RuntimeError(f"libdib_hip ABI version {have} != {ABI_VERSION} expected by this binding ({getattr(lib, '_name', '?')}): "
                           "rebuild it (python -c 'import __graft_entry__ as g; g.build()' / tools/build_variant.sh)")
    for name, (res, args) in list(SIGNATURES.items()) + list(SIGNATURES_ST.items()):
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def set_tuning(key: str, value: int) -> None:
    """include/dib_hip.h dib_set_tuning: the library's only hidden inputs (it reads no environment variable)."""
    check(load_library().dib_set_tuning(key.encode(), int(value)), f"dib_set_tuning({key})")


def get_tuning(key: str) -> int:
    v = c_int()
    check(load_library().dib_get_tuning(key.encode(), ctypes.byref(v)), f"dib_get_tuning({key})")
    return int(v.value)


class DibError(RuntimeError):
    pass


def check(code: int, what: str = "") -> None:
    if code != DIB_OK:
        msg = load_library().dib_error_string(int(code)).decode()
        raise DibError(f"libdib_hip {what} failed: {msg} (code {code})")
