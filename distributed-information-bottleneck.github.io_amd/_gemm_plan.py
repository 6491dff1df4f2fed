"""Grouped-GEMM launch records shared by the host classes that drive `dib_gemm_grouped` (include/dib_st.h) directly:
SetTransformerDIB (set_transformer.py) and DenseStack (dense.py).  A `_Gemm` is one launch: a device-resident descriptor
table (one `dib_gemm_desc` per group: element offsets into the base tensors, leading dimensions, M/N/K) plus the base
tensors; the tables are built and uploaded once per batch shape, so a step is a sequence of C-ABI calls with no per-call
descriptor traffic."""
from __future__ import annotations

from ctypes import c_void_p
from typing import Optional

import numpy as np
import torch

from ._lib import check

DESC = np.dtype([("a_off", "<i8"), ("b_off", "<i8"), ("c_off", "<i8"), ("bias_off", "<i8"), ("aux_off", "<i8"),
                 ("a_boff", "<i8"), ("b_boff", "<i8"), ("c_boff", "<i8"), ("aux_boff", "<i8"),
                 ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("lda", "<i4"), ("ldb", "<i4"), ("ldc", "<i4"), ("ldaux", "<i4"),
                 ("flags", "<i4")])
assert DESC.itemsize == 104  # include/dib_st.h: dib_gemm_desc


def _ptr(t: Optional[torch.Tensor], off: int = 0):
    return c_void_p(t.data_ptr() + 4 * off) if t is not None else c_void_p(0)


def _ptr8(t: Optional[torch.Tensor]):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)



class _Gemm:
    """One grouped-GEMM launch: a device descriptor table + base tensors."""

    def __init__(self, mode, descs, A, B, C, bias=None, aux=None, bias_out=None, act=0, nsplit=1, rows_per_split=0,
                 split_stride=0):
        self.mode, self.n = mode, len(descs)
        self.max_m = int(max(d["M"] for d in descs))
        self.max_n = int(max(d["N"] for d in descs))
        arr = np.zeros(len(descs), dtype=DESC)
        for i, d in enumerate(descs):
            for k, v in d.items():
                arr[i][k] = v
        self.host = arr
        self.dev = None
        self.A, self.B, self.C, self.bias, self.aux, self.bias_out = A, B, C, bias, aux, bias_out
        self.act, self.nsplit, self.rps, self.stride = act, nsplit, rows_per_split, split_stride

    def upload(self, device):
        self.dev = torch.from_numpy(self.host.view(np.uint8).copy()).to(device)

    def run(self, lib, stream):
        check(lib.dib_gemm_grouped(self.mode, self.n, _ptr(self.dev), self.max_m, self.max_n, _ptr(self.A), _ptr(self.B),
                                   _ptr(self.C), _ptr(self.bias), _ptr(self.aux), _ptr(self.bias_out), self.act, self.nsplit,
                                   self.rps, self.stride, stream), "dib_gemm_grouped")



class _SkinnyKGemm(_Gemm):
    """The same launch record for `dib_gemm_skinny_k` (include/dib_st.h): mode 0 / 1 products whose contraction is at most
    32 wide and whose output is large - a streaming kernel bound by its output stores.  All groups share M, N, K."""

    @staticmethod
    def fits(mode, descs, act=0, aux=None) -> bool:
        d0 = descs[0]
        return (mode in (0, 1) and act == 0 and aux is None and 0 < d0["K"] <= 32 and d0["K"] % 4 == 0 and d0["N"] % 32 == 0
                and all((d["M"], d["N"], d["K"]) == (d0["M"], d0["N"], d0["K"]) for d in descs))

    def run(self, lib, stream):
        d0 = self.host[0]
        check(lib.dib_gemm_skinny_k(self.mode, self.n, _ptr(self.dev), int(d0["M"]), int(d0["N"]), int(d0["K"]), _ptr(self.A),
                                    _ptr(self.B), _ptr(self.C), _ptr(self.bias), stream), "dib_gemm_skinny_k")


def _d(a_off, lda, b_off, ldb, c_off, ldc, M, N, K, bias_off=-1, aux_off=0, ldaux=0):
    return dict(a_off=a_off, b_off=b_off, c_off=c_off, bias_off=bias_off, aux_off=aux_off, M=M, N=N, K=K, lda=lda, ldb=ldb,
                ldc=ldc, ldaux=ldaux)


