"""Context-tree-weighting entropy-rate estimator: Python surface of csrc/dib_ctw.cpp (include/dib_ctw.h).

Mirrors the reference's Cython module `ctw` (reference chaos/ctw.pyx:2, chaos/README.MD usage):

    from dib_amd.ctw import estimate_entropy
    estimate_entropy(seq, alphabet_size)          # bits per symbol, same value as the reference build

plus `estimate_entropy_batch` for the notebook's "75 sequences per partition" pattern (host threads).
Host C++ only - no GPU involved; the library is built in-tree with g++ on first use.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "dib_ctw.cpp")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "dib_ctw.h")
LIB_PATH = os.path.join(_HERE, "libdib_ctw.so")

SIGNATURES = {
    "dib_ctw_version": (c_char_p, []),
    "dib_ctw_estimate_entropy": (c_int, [c_void_p, c_int64, c_int, POINTER(c_double)]),
    "dib_ctw_estimate_entropy_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dib_ctw_node_count": (c_int, [c_void_p, c_int64, c_int, POINTER(c_int64)]),
}
_ERRORS = {-1: "invalid argument (empty pointer, alphabet outside [1,127] or symbol outside [0, alphabet))",
           -2: "out of memory while building the context tree"}
_lib = None


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in (SRC, HEADER))


def build_library(force: bool = False, verbose: bool = False) -> str:
    """g++ -O2 (no fast-math: results must stay bit-identical to the reference arithmetic)."""
    if not force and not _stale():
        return LIB_PATH
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found: cannot build libdib_ctw.so")
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", SRC, "-o", LIB_PATH + f".tmp{os.getpid()}"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed:\n" + res.stdout + res.stderr)
    os.replace(LIB_PATH + f".tmp{os.getpid()}", LIB_PATH)  # per-process temp name: concurrent ranks cannot clobber each other
    return LIB_PATH


def load_library():
    global _lib
    if _lib is None:
        if _stale():
            try:
                build_library()
            except Exception:
                if not os.path.exists(LIB_PATH):
                    raise
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _symbols(seq, alphabet_size: int) -> np.ndarray:
    a = np.asarray(seq)
    if a.ndim != 1:
        a = a.reshape(-1)
    if a.size and (a.min() < 0 or a.max() >= alphabet_size):
        raise ValueError(f"symbols must lie in [0, {alphabet_size})")
    return np.ascontiguousarray(a, dtype=np.int8)


def _check(rc: int) -> None:
    if rc != 0:
        raise ValueError("dib_ctw: " + _ERRORS.get(rc, f"error {rc}"))


def estimate_entropy(seq, alphabet_size: int) -> float:
    """Entropy rate (bits/symbol) of a symbol sequence - drop-in for reference chaos/ctw.pyx:2."""
    alphabet_size = int(alphabet_size)
    if not 1 <= alphabet_size <= 127:
        raise ValueError("alphabet_size must be in [1, 127]")
    s = _symbols(seq, alphabet_size)
    out = c_double()
    _check(load_library().dib_ctw_estimate_entropy(s.ctypes.data, s.size, alphabet_size, ctypes.byref(out)))
    return out.value


def estimate_entropy_batch(seqs, alphabet_size: int, threads: int = 0) -> np.ndarray:
    """Entropy rates of many sequences (list of 1-D arrays, or a 2-D array of equal-length rows), computed on
    `threads` host threads (0 = all)."""
    alphabet_size = int(alphabet_size)
    if not 1 <= alphabet_size <= 127:
        raise ValueError("alphabet_size must be in [1, 127]")
    parts = [_symbols(s, alphabet_size) for s in seqs]
    offsets = np.zeros(len(parts) + 1, dtype=np.int64)
    if parts:
        offsets[1:] = np.cumsum([p.size for p in parts])
    flat = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int8)
    out = np.empty(len(parts), dtype=np.float64)
    _check(load_library().dib_ctw_estimate_entropy_batch(flat.ctypes.data, offsets.ctypes.data, len(parts), alphabet_size,
                                                        int(threads), out.ctypes.data))
    return out


def node_count(seq, alphabet_size: int) -> int:
    """Size of the suffix tree built for `seq` (memory: 8*alphabet_size + 8 bytes per node)."""
    s = _symbols(seq, int(alphabet_size))
    out = c_int64()
    _check(load_library().dib_ctw_node_count(s.ctypes.data, s.size, int(alphabet_size), ctypes.byref(out)))
    return out.value
