"""Geometry of a LIST of torch tensors -- data pointer, sizes, strides, dtype, device -- as numpy arrays, in one pass.

A database flush hands ``fingerprint_batch`` 2 x 2 048 embedding matrices and 2 048 contact maps; the C ABI (include/dctfp.h)
wants host arrays of device pointers and row counts.  Reading ``data_ptr`` / ``size`` / ``stride`` / ``dtype`` / ``device`` of
every tensor through Python cost 8 ms of a 22 ms flush (profiles/r05/flush_timeline_before.txt).  ``_tensor_table.so``
(csrc/tensor_table.cpp, built by build_ext.py against torch's headers) walks the list in C++; without it the same table is
filled one attribute at a time.  Host plumbing only: nothing here touches the data."""

from __future__ import annotations

import importlib.util
import os
from itertools import repeat

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_helper = None
_helper_tried = False

#: columns of the meta table
DIM, SIZE0, SIZE1, STRIDE0, STRIDE1, CODE = range(6)


def helper():
    """The compiled walker, or None when it has not been built (or cannot be loaded against this torch)."""
    global _helper, _helper_tried
    if not _helper_tried:
        _helper_tried = True
        path = os.path.join(_HERE, '_tensor_table.so')
        if os.path.exists(path) and os.environ.get('DCTFP_NO_TENSOR_TABLE') != '1':
            try:
                spec = importlib.util.spec_from_file_location('dctdomain_amd._tensor_table', path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _helper = mod
            except Exception:        # noqa: BLE001  (built against another torch: fall back)
                _helper = None
    return _helper


_SCALAR = {torch.uint8: 0, torch.int8: 1, torch.int16: 2, torch.int32: 3, torch.int64: 4, torch.float16: 5, torch.float32: 6,
           torch.float64: 7, torch.bool: 11, torch.bfloat16: 15}       # c10::ScalarType
_DEVTYPE = {'cpu': 0, 'cuda': 1}                                       # c10::DeviceType (a ROCm build calls its GPUs 'cuda')


def code_of(t: torch.Tensor) -> int:
    """The dtype / device code the table holds for a tensor like ``t``."""
    h = helper()
    if h is not None:
        return int(h.scalar_code(t))
    dev = t.device
    return _SCALAR.get(t.dtype, 255) | (_DEVTYPE.get(dev.type, 255) << 8) | (((dev.index + 1) if dev.index is not None else 0) << 16)


def tensor_table(seq):
    """(ptrs uint64[n], meta int64[n, 6]) of a list / tuple: meta[i] = {dim, size(0), size(1), stride(0), stride(1), code};
    dim = -1 marks an entry that is no tensor."""
    n = len(seq)
    ptrs = np.empty(n, dtype=np.uint64)
    meta = np.empty((n, 6), dtype=np.int64)
    if n == 0:
        return ptrs, meta
    h = helper()
    if h is not None:
        try:
            h.fill(seq, ptrs.ctypes.data, meta.ctypes.data, n)
            return ptrs, meta
        except RuntimeError:         # (a tensor ATen would not describe -- sparse, ...: attribute by attribute, and whoever reads the table refuses it)
            pass
    if not all(map(torch.is_tensor, seq)):
        for i, t in enumerate(seq):          # (mixed lists are rare: entry by entry)
            if torch.is_tensor(t):
                d = t.dim()
                ptrs[i] = t.data_ptr()
                meta[i] = (d, t.size(0) if d >= 1 else 0, t.size(1) if d >= 2 else 0, t.stride(0) if d >= 1 else 0,
                           t.stride(1) if d >= 2 else 0, code_of(t))
            else:
                ptrs[i] = 0
                meta[i] = (-1, 0, 0, 0, 0, 0)
        return ptrs, meta
    dims = np.fromiter(map(torch.Tensor.dim, seq), dtype=np.int64, count=n)
    ptrs[:] = np.fromiter(map(torch.Tensor.data_ptr, seq), dtype=np.uint64, count=n)
    meta[:, DIM] = dims
    if (dims == 2).all():
        meta[:, SIZE0] = np.fromiter(map(torch.Tensor.size, seq, repeat(0)), dtype=np.int64, count=n)
        meta[:, SIZE1] = np.fromiter(map(torch.Tensor.size, seq, repeat(1)), dtype=np.int64, count=n)
        meta[:, STRIDE0] = np.fromiter(map(torch.Tensor.stride, seq, repeat(0)), dtype=np.int64, count=n)
        meta[:, STRIDE1] = np.fromiter(map(torch.Tensor.stride, seq, repeat(1)), dtype=np.int64, count=n)
    else:
        for i, t in enumerate(seq):
            d = int(dims[i])
            meta[i, SIZE0:CODE] = (t.size(0) if d >= 1 else 0, t.size(1) if d >= 2 else 0, t.stride(0) if d >= 1 else 0,
                                   t.stride(1) if d >= 2 else 0)
    meta[:, CODE] = np.fromiter(map(code_of, seq), dtype=np.int64, count=n)
    return ptrs, meta
