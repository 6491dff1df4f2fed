"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's context-tree-weighting entropy-rate estimator.

Follows reference chaos/cppctw.cpp line by line (each function cites the lines it restates) with plain Python
objects, for sequences of up to a few thousand symbols.  Only tests/ may import this module; the product
(distributed-information-bottleneck.github.io_amd/csrc/dib_ctw.cpp) never does.

Pinning: tests/golden/ctw_golden.npz holds rates produced by the reference's own C++ source compiled where it lies
(oracle/Makefile -> oracle/_ref/libctw_ref.so; generator tests/golden/make_golden_ctw.py).  This restatement is checked
against those vectors bit for bit (tests/test_ctw.py), so parity here is pinned to the real reference.

lgamma / pow / log2 are taken from the C library through ctypes: Python's math.lgamma is a different implementation
and would differ from the reference's libm calls in the last bits.
"""
import ctypes
import ctypes.util

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _name, _n in (("lgamma", 1), ("pow", 2), ("log2", 1), ("log", 1)):
    _f = getattr(_libm, _name)
    _f.restype = ctypes.c_double
    _f.argtypes = [ctypes.c_double] * _n

MAX_DEPTH = 512  # chaos/cppctw.cpp:13


class _Node:
    """chaos/cppctw.cpp:15-41 (SuffixTreeNode fields and constructors)."""
    __slots__ = ("children", "counts", "tail_ind", "tail_symbol", "weighted", "local")

    def __init__(self, alphabet, tail_ind=-1, tail_symbol=-1):
        self.children = [None] * alphabet
        self.counts = [0] * alphabet
        self.tail_ind = tail_ind
        self.tail_symbol = tail_symbol
        self.weighted = 0.0
        self.local = 0.0


def build_tree(seq, alphabet):
    """chaos/cppctw.cpp:104-152 (SuffixTree::process_sequence)."""
    root = _Node(alphabet)
    for t, sym in enumerate(seq):
        node = root
        node.counts[sym] += 1                                   # :117
        for c in range(t - 1, -1, -1):                          # :119
            if node.tail_ind > 0:                               # :121-128 expand a tail leaf by one context symbol
                back = seq[node.tail_ind - 1]
                kid = _Node(alphabet, node.tail_ind - 1, node.tail_symbol)
                kid.counts[node.tail_symbol] += 1
                node.children[back] = kid
                node.tail_ind = -1
                node.tail_symbol = -1
            ctx = seq[c]                                        # :130
            if node.children[ctx] is None:                      # :131-146
                if t - c > MAX_DEPTH:
                    break
                kid = _Node(alphabet, c, sym) if c > 0 else _Node(alphabet)
                kid.counts[sym] += 1
                node.children[ctx] = kid
                break
            node = node.children[ctx]                           # :147-148
            node.counts[sym] += 1
    return root


def _update_code_lengths(node, alphabet, beta):
    """chaos/cppctw.cpp:55-82; explicit stack instead of recursion, same evaluation order of every sum."""
    order, stack = [], [node]
    while stack:
        v = stack.pop()
        order.append(v)
        stack.extend(k for k in v.children if k is not None)
    ln2 = _libm.log(2.0)
    for v in reversed(order):                                   # children before parents
        total = 0.0
        for cnt in v.counts:
            total += cnt
        le = _libm.lgamma(total + alphabet * beta) - _libm.lgamma(alphabet * beta)
        for cnt in v.counts:
            le -= _libm.lgamma(cnt + beta) - _libm.lgamma(beta)
        le /= ln2
        v.local = le
        l_c, childfull = 0.0, False
        for k in v.children:                                    # ascending symbol order, as the reference loop
            if k is not None:
                childfull = True
                l_c += k.weighted
        if childfull and total > 1:
            v.weighted = 1 + min(l_c, le) - _libm.log2(1 + _libm.pow(2.0, -abs(le - l_c)))
        else:
            v.weighted = le


def estimate_entropy(seq, alphabet_size):
    """chaos/cppctw.cpp:160-171 + :98-102: beta = 1/|A|; the rate is rounded to float32 (the method returns float)."""
    seq = [int(s) for s in seq]
    alphabet = int(alphabet_size)
    beta = 1.0 / alphabet
    root = build_tree(seq, alphabet)
    _update_code_lengths(root, alphabet, beta)
    if len(seq) == 0:
        return float("nan")
    return float(np.float32(root.weighted / len(seq)))


def count_nodes(seq, alphabet_size):
    root, n, stack = build_tree([int(s) for s in seq], int(alphabet_size)), 0, []
    stack.append(root)
    while stack:
        v = stack.pop()
        n += 1
        stack.extend(k for k in v.children if k is not None)
    return n
