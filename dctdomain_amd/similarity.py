"""GPU L1 distances between int8 fingerprints (``dctfp_l1_matrix`` / ``dctfp_block_min``): the
arithmetic under the reference's two consumers, ``src/dct-sim.py`` and ``src/query_db.py``."""

from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from . import _lib


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('dctdomain_amd needs an MI355X GPU; there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def to_device_int8(fps) -> torch.Tensor:
    t = fps if isinstance(fps, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(fps, dtype=np.int8)))
    if t.dtype != torch.int8:
        t = t.to(torch.int8)
    if t.device.type != 'cuda':
        t = t.to(_dev())
    return t.contiguous()


def l1_matrix(a, b) -> torch.Tensor:
    """int32 (na, nb) matrix of L1 distances, on the GPU."""
    ta, tb = to_device_int8(a), to_device_int8(b)
    if ta.dim() != 2 or tb.dim() != 2 or ta.shape[1] != tb.shape[1]:
        raise ValueError('fingerprint sets must be 2-D with equal width')
    out = torch.empty((ta.shape[0], tb.shape[0]), dtype=torch.int32, device=ta.device)
    if out.numel():
        ctx = _lib.get_context(ta.device.index)
        stream = torch.cuda.current_stream(ta.device)
        lda = ta.stride(0) if ta.shape[0] > 1 else ta.shape[1]      # a size-1 axis may carry any stride
        ldb = tb.stride(0) if tb.shape[0] > 1 else tb.shape[1]
        ldo = out.stride(0) if out.shape[0] > 1 else out.shape[1]
        _lib.check(ctx._lib.dctfp_l1_matrix(ctx.handle, ta.data_ptr(), ta.shape[0], lda, tb.data_ptr(),
                                            tb.shape[0], ldb, ta.shape[1], out.data_ptr(), ldo,
                                            C.c_void_p(stream.cuda_stream)))
    return out


def block_min(dist: torch.Tensor, idx_a, idx_b):
    """(min, last) int32 arrays of shape (npa, npb) over the protein blocks of ``dist``."""
    ia = torch.as_tensor(np.asarray(idx_a, dtype=np.int64), device=dist.device)
    ib = torch.as_tensor(np.asarray(idx_b, dtype=np.int64), device=dist.device)
    npa, npb = len(ia) - 1, len(ib) - 1
    if dist.numel() == 0:       # proteins without a single fingerprint on either side: every block is empty
        empty = np.full((npa, npb), 0x7fffffff, dtype=np.int32)     # what block_min_kernel writes for an empty block
        return empty, empty.copy()
    mn = torch.empty((npa, npb), dtype=torch.int32, device=dist.device)
    last = torch.empty((npa, npb), dtype=torch.int32, device=dist.device)
    if mn.numel():
        ctx = _lib.get_context(dist.device.index)
        stream = torch.cuda.current_stream(dist.device)
        _lib.check(ctx._lib.dctfp_block_min(ctx.handle, dist.data_ptr(),
                                            dist.stride(0) if dist.shape[0] > 1 else dist.shape[1], ia.data_ptr(), npa,
                                            ib.data_ptr(), npb, mn.data_ptr(), last.data_ptr(),
                                            C.c_void_p(stream.cuda_stream)))
    return _pair_to_host(mn, last, np.int32)


def order_pairs(v: np.ndarray, i: np.ndarray):
    """(values, indices) with every row ordered by (value, index): one sort of packed 64-bit keys instead of ``np.lexsort``
    along an axis (134 ms for 6 700 x 100 pairs, and as long again to apply; this: 5 ms).  int32-range values, indices < 2^32."""
    key = ((v.astype(np.int64) + (1 << 31)).astype(np.uint64) << np.uint64(32)) | i.astype(np.uint64)
    key.sort(axis=1)
    return (key >> np.uint64(32)).astype(np.int64) - (1 << 31), (key & np.uint64(0xffffffff)).astype(np.int64)


def row_select(dist: torch.Tensor, k: int):
    """(values, indices) numpy arrays (n_rows, k): the k smallest entries of each row, ascending, ties
    to the lower column -- selected (``dctfp_row_select``) and ordered (``dctfp_row_order``, k <= 1024) on the GPU."""
    n_rows, n_cols = dist.shape
    k = min(int(k), n_cols)
    val = torch.empty((n_rows, k), dtype=torch.int32, device=dist.device)
    idx = torch.empty((n_rows, k), dtype=torch.int32, device=dist.device)
    on_device = k <= 1024
    if n_rows:
        ctx = _lib.get_context(dist.device.index)
        stream = C.c_void_p(torch.cuda.current_stream(dist.device).cuda_stream)
        _lib.check(ctx._lib.dctfp_row_select(ctx.handle, dist.data_ptr(), n_rows, n_cols,
                                             dist.stride(0) if n_rows > 1 else n_cols, k, val.data_ptr(), idx.data_ptr(), stream))
        if on_device:
            _lib.check(ctx._lib.dctfp_row_order(ctx.handle, val.data_ptr(), idx.data_ptr(), n_rows, k, stream))
    v, i = _pair_to_host(val, idx)
    return (v, i) if on_device else order_pairs(v, i)


_PINNED = threading.local()


def _pair_to_host(val: torch.Tensor, idx: torch.Tensor, dtype=np.int64):
    """Two int32 device tensors of one shape -> numpy arrays through one page-locked staging buffer of this thread (a pageable
    ``.cpu()`` of a tile's 2 x 2.7 MB took 18 ms each on the GPU boxes; this: both in under a millisecond)."""
    n = val.numel()
    pin = getattr(_PINNED, 'buf', None)
    if pin is None or pin.numel() < 2 * n:
        pin = torch.empty(max(2 * n + n // 2, 1 << 18), dtype=torch.int32, pin_memory=True)
        _PINNED.buf = pin
    pin[:n].view(val.shape).copy_(val, non_blocking=True)
    pin[n:2 * n].view(idx.shape).copy_(idx, non_blocking=True)
    torch.cuda.current_stream(val.device).synchronize()
    host = pin[:2 * n].numpy().astype(dtype)                    # (the copy out of the staging buffer, and the widening, in one)
    return host[:n].reshape(val.shape), host[n:].reshape(idx.shape)
