"""Ragged-batch front end of the fingerprint kernels: many sequences, several layers,
several domains per sequence, one ``dctfp_quantize`` call.

The reference processes one ``Fingerprint`` at a time inside a multiprocessing pool
(mgtools/DCTdomain src/make_db.py:36-51).  Here the embeddings of a whole batch stay on the
GPU as torch tensors and the domain strings are turned into one piece table."""

from __future__ import annotations

import ctypes as C
from itertools import chain
from typing import List, Sequence

import numpy as np
import torch

from itertools import repeat
from operator import attrgetter

from . import _lib
from .domains import split_domain

_DTYPE_OF, _DEVICE_OF = attrgetter('dtype'), attrgetter('device')


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.DCTFP_F32
    if t.dtype == torch.float64:
        return _lib.DCTFP_F64
    if t.dtype == torch.float16:
        return _lib.DCTFP_F16
    if t.dtype == torch.bfloat16:
        return _lib.DCTFP_BF16
    raise TypeError(f'embedding dtype {t.dtype}: use float32, float64, float16 or bfloat16')


class PieceTable:
    """Host-side piece table of a batch (numpy image of ``dctfp_piece[]``)."""

    def __init__(self, seq_rows: Sequence[int], domains: Sequence[Sequence[str]]):
        """``seq_rows[s]`` rows of sequence ``s``; ``domains[s]`` its domain strings.
        Domains whose cleaned piece list is empty are skipped, as the reference does
        (src/fingerprint.py:190-191).

        The strings are parsed by ``dctfp_build_pieces`` (include/dctfp.h) in one call -- a database flush has hundreds of
        thousands of them, and the per-string Python loop this replaces cost more than the kernels of the flush
        (profiles/r03/host_time.txt: 176 ms for 170 044 domains against 12 ms of GPU).  Strings that are not of the plain
        form ``digits-digits[,...]`` go through ``domains.split_domain``, which keeps Python's own parsing rules."""
        self.seq_rows = np.ascontiguousarray(np.asarray(seq_rows, dtype=np.int64))
        n_seq = len(self.seq_rows)
        if len(domains) != n_seq:
            raise ValueError(f'{len(domains)} domain lists for {n_seq} sequences')
        if n_seq == 1 and len(domains[0]) <= 8:   # a protein at a time: the loop over a handful of strings beats the call overhead
            self._init_python(list(domains[0]), (len(domains[0]),))
            return
        counts = np.fromiter(map(len, domains), dtype=np.int32, count=n_seq)
        n_str = int(counts.sum())
        flat = list(chain.from_iterable(domains))   # the table's own list: the caller's may change before keys are asked for
        try:
            text = '\n'.join(flat).encode('ascii')
        except (UnicodeEncodeError, TypeError):
            text = None
        if text is None or n_str <= 8:
            self._init_python(flat, counts)
            return
        cap = len(text) // 4 + 2               # a piece is at least "b-e" and a separator
        pieces = np.empty(cap, dtype=_lib.PIECE_DTYPE)
        str_row = np.empty(n_str, dtype=np.int32)
        str_len = np.empty(n_str, dtype=np.int64)
        changed = np.empty(n_str, dtype=np.uint8)
        key_text = C.create_string_buffer(len(text) + n_str + 1)
        n_pieces, key_len, n_dom, n_other = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        lib = _lib.load()
        _lib.check(lib.dctfp_build_pieces(text, len(text), counts.ctypes.data, self.seq_rows.ctypes.data, n_seq,
                                          pieces.ctypes.data, cap, C.byref(n_pieces), str_row.ctypes.data, str_len.ctypes.data,
                                          changed.ctypes.data, key_text, len(key_text), C.byref(key_len), C.byref(n_dom),
                                          C.byref(n_other)), lib)
        if n_other.value:                      # a string Python's int() / split must judge (or one with a line break inside)
            self._init_python(flat, counts)
            return
        self.n_domains = n_dom.value
        self.pieces = pieces[:n_pieces.value]
        kept = str_row >= 0
        self.lengths = str_len if self.n_domains == n_str else str_len[kept]
        # keys / owner / source name the results; they are built when somebody asks (a timed loop over dctfp_quantize does not)
        self._lazy = (flat, counts, None if self.n_domains == n_str else kept, changed,
                      key_text.raw[:key_len.value] if key_len.value else b'')

    def _resolve(self):
        flat, counts, kept, changed, key_blob = self._lazy
        self._lazy = None
        n_seq, n_str = len(counts), int(counts.sum())
        owner = np.repeat(np.arange(n_seq, dtype=np.int64), counts)
        first = np.zeros(n_seq + 1, dtype=np.int64)
        np.cumsum(counts, out=first[1:])
        source = np.arange(n_str, dtype=np.int64) - np.repeat(first[:-1], counts)
        if key_blob:
            for i, k in zip(np.flatnonzero(changed == 1), key_blob.decode('ascii').split('\n')):
                flat[i] = k
        if kept is not None:
            idx = np.flatnonzero(kept)
            flat, owner, source = [flat[i] for i in idx], owner[idx], source[idx]
        self._keys, self._owner, self._source = flat, owner, source

    @property
    def keys(self) -> List[str]:
        """Cleaned key per output row: the name under which the reference files the fingerprint."""
        if self._lazy is not None:
            self._resolve()
        return self._keys

    @property
    def owner(self):
        """Sequence index per output row (numpy int64)."""
        if self._lazy is not None:
            self._resolve()
        if '_counts' in self.__dict__:
            self._from_counts()
        return self._owner

    @property
    def source(self):
        """Index of the domain string inside ``domains[s]`` per output row (numpy int64)."""
        if self._lazy is not None:
            self._resolve()
        if '_counts' in self.__dict__:
            self._from_counts()
        return self._source

    def _init_python(self, flat, counts):
        """The same table from ``domains.split_domain`` string by string (Python's own int() / split semantics)."""
        recs, keys, owner, source, lengths = [], [], [], [], []
        d = i = 0
        for s, c in enumerate(counts):
            n_rows = int(self.seq_rows[s])
            for di in range(int(c)):
                pieces, key = split_domain(flat[i], n_rows)
                i += 1
                if not pieces:
                    continue
                for start, n in pieces:
                    recs.append((start, n, d, s, 0))
                keys.append(key)
                owner.append(s)
                source.append(di)
                lengths.append(sum(n for _, n in pieces))
                d += 1
        self.n_domains = d
        self.pieces = np.array(recs, dtype=_lib.PIECE_DTYPE) if recs else np.zeros(0, dtype=_lib.PIECE_DTYPE)
        self._lazy = None
        self._keys = keys
        self._owner = np.asarray(owner, dtype=np.int64)
        self._source = np.asarray(source, dtype=np.int64)
        self.lengths = np.asarray(lengths, dtype=np.int64)

    @classmethod
    def from_pieces(cls, seq_rows, pieces, n_domains: int, keys: List[str], str_count):
        """A table whose pieces were built elsewhere (``dctfp_reccut_pieces``: the domain cutter's results of a flush, strings
        and pieces in one pass): ``keys[d]`` names output row d, ``str_count[s]`` rows belong to sequence s, in order."""
        self = cls.__new__(cls)
        self.seq_rows = np.ascontiguousarray(np.asarray(seq_rows, dtype=np.int64))
        self.pieces = pieces
        self.n_domains = int(n_domains)
        if len(keys) != self.n_domains:
            raise ValueError(f'{len(keys)} keys for {self.n_domains} domains')
        self._lazy = None
        self._keys = keys
        self._counts = np.asarray(str_count, dtype=np.int64)
        return self

    def _from_counts(self):
        counts = self.__dict__.pop('_counts')
        n_seq = len(counts)
        first = np.zeros(n_seq + 1, dtype=np.int64)
        np.cumsum(counts, out=first[1:])
        self._owner = np.repeat(np.arange(n_seq, dtype=np.int64), counts)
        self._source = np.arange(self.n_domains, dtype=np.int64) - np.repeat(first[:-1], counts)

    @property
    def lengths(self):
        """Rows of every domain (numpy int64)."""
        v = self.__dict__.get('_lengths')
        if v is None:
            v = np.bincount(self.pieces['domain'], weights=self.pieces['n_rows'], minlength=self.n_domains).astype(np.int64)
            self._lengths = v
        return v

    @lengths.setter
    def lengths(self, v):
        self._lengths = v

    @classmethod
    def whole_sequences(cls, seq_rows: Sequence[int]):
        """One domain ``1-L`` per sequence, built without string parsing (bench path)."""
        self = cls.__new__(cls)
        sr = np.ascontiguousarray(np.asarray(seq_rows, dtype=np.int64))
        n = len(sr)
        p = np.zeros(n, dtype=_lib.PIECE_DTYPE)
        p['row_start'] = 0
        p['n_rows'] = sr
        p['domain'] = np.arange(n)
        p['seq'] = np.arange(n)
        self.pieces = p
        self.seq_rows = sr
        self.n_domains = n
        self._lazy = None
        self._keys = [f'1-{v}' for v in sr.tolist()]
        self._owner = np.arange(n, dtype=np.int64)
        self._source = np.zeros(n, dtype=np.int64)
        self.lengths = sr.copy()
        return self


class LayerBatch:
    """One embedding layer of a batch: per-sequence device matrices sharing D, dtype and ld.

    ``tensors`` is either one 2-D tensor holding all sequences back to back (then
    ``row_offsets`` gives each sequence's first row) or a list of 2-D tensors."""

    def __init__(self, tensors, n_keep: int, m_keep: int, row_offsets=None):
        if isinstance(tensors, torch.Tensor):
            big = tensors
            if big.dim() != 2 or big.stride(1) != 1:
                raise ValueError('layer tensor must be 2-D with contiguous channels')
            if row_offsets is None:
                row_offsets = [0]
            esz = big.element_size()
            base = big.data_ptr()
            self.ptrs = np.uint64(base) + np.asarray(row_offsets, dtype=np.uint64) * np.uint64(big.stride(0) * esz)
            self.ld = big.stride(0)
            self.n_cols = big.shape[1]
            self.dtype = _dtype_code(big)
            self.device = big.device
            self._keep = [big]
        else:
            ts = list(tensors)
            if not ts:
                raise ValueError('empty layer')
            t0 = ts[0]
            if not all(map(torch.is_tensor, ts)) or t0.dim() != 2:
                raise ValueError('all sequences of a layer must be 2-D matrices')
            dt, dev, width = t0.dtype, t0.device, t0.shape[1]
            # one C-level pass per attribute (a flush hands over thousands of tensors; a Python expression per tensor was 5 ms of it)
            n = len(ts)
            dims = np.fromiter(map(torch.Tensor.dim, ts), dtype=np.int64, count=n)
            ok = bool((dims == 2).all()) and len(set(map(_DTYPE_OF, ts))) == 1 and len(set(map(_DEVICE_OF, ts))) == 1
            ld = width
            if ok:
                rows = np.fromiter(map(torch.Tensor.size, ts, repeat(0)), dtype=np.int64, count=n)
                cols = np.fromiter(map(torch.Tensor.size, ts, repeat(1)), dtype=np.int64, count=n)
                st0 = np.fromiter(map(torch.Tensor.stride, ts, repeat(0)), dtype=np.int64, count=n)
                st1 = np.fromiter(map(torch.Tensor.stride, ts, repeat(1)), dtype=np.int64, count=n)
                longer = rows > 1               # (a matrix of one row says nothing about the row stride)
                ld = int(st0[longer][0]) if longer.any() else width
                ok = bool((cols == width).all() and (st1 == 1).all() and (st0[longer] == ld).all())
            if not ok:
                raise ValueError('all sequences of a layer must share D, dtype, device and row stride')
            self.ptrs = np.fromiter(map(torch.Tensor.data_ptr, ts), dtype=np.uint64, count=len(ts))
            self.ld = ld
            self.n_cols = width
            self.dtype = _dtype_code(t0)
            self.device = dev
            self._keep = ts
        if self.device.type != 'cuda':
            raise ValueError('embeddings must live on the GPU (torch device "cuda")')
        self.ptrs = np.ascontiguousarray(self.ptrs, dtype=np.uint64)   # host array of device pointers
        self.n_keep = int(n_keep)
        self.m_keep = int(m_keep)

    @classmethod
    def from_table(cls, tensors, ptrs, meta, n_keep: int, m_keep: int):
        """The list constructor's checks on a geometry table of the same tensors (``_geom.tensor_table``: one C++ pass over the
        list instead of nine Python passes -- 2 of the 6 ms a flush of 2 048 proteins spent on its embedding tables)."""
        from . import _geom
        n = len(tensors)
        if n == 0:
            raise ValueError('empty layer')
        if not (meta[:, _geom.DIM] == 2).all():
            raise ValueError('all sequences of a layer must be 2-D matrices')
        t0 = tensors[0]
        width = int(meta[0, _geom.SIZE1])
        rows, st0 = meta[:, _geom.SIZE0], meta[:, _geom.STRIDE0]
        longer = rows > 1               # (a matrix of one row says nothing about the row stride)
        ld = int(st0[longer][0]) if longer.any() else width
        if not ((meta[:, _geom.CODE] == meta[0, _geom.CODE]).all() and (meta[:, _geom.SIZE1] == width).all()
                and (meta[:, _geom.STRIDE1] == 1).all() and (st0[longer] == ld).all()):
            raise ValueError('all sequences of a layer must share D, dtype, device and row stride')
        self = cls.__new__(cls)
        self.ptrs = np.ascontiguousarray(ptrs, dtype=np.uint64)
        self.ld = ld
        self.n_cols = width
        self.dtype = _dtype_code(t0)
        self.device = t0.device
        self._keep = tensors
        if self.device.type != 'cuda':
            raise ValueError('embeddings must live on the GPU (torch device "cuda")')
        self.n_keep = int(n_keep)
        self.m_keep = int(m_keep)
        return self


def host_pointer(t: torch.Tensor) -> int:
    """The address under which the GPU sees a pinned host tensor (``dctfp_host_device_pointer``); raises when the
    tensor is not page-locked and mapped."""
    if not t.is_pinned():
        raise ValueError('a host tensor used as the result buffer must be pinned (pin_memory=True)')
    dev = C.c_void_p()
    _lib.check(_lib.load().dctfp_host_device_pointer(C.c_void_p(t.data_ptr()), C.byref(dev)))
    return dev.value


def _layer_array(layers: Sequence[LayerBatch], n_data: int):
    arr = (_lib.Layer * len(layers))()
    off = 0
    for i, l in enumerate(layers):
        if len(l.ptrs) != n_data:
            raise ValueError(f'layer {i} has {len(l.ptrs)} matrices, the call speaks of {n_data}')
        arr[i].seq_data = C.cast(C.c_void_p(l.ptrs.ctypes.data), C.POINTER(C.c_void_p))
        arr[i].ld = l.ld
        arr[i].n_cols = l.n_cols
        arr[i].dtype = l.dtype
        arr[i].n_keep = l.n_keep
        arr[i].m_keep = l.m_keep
        arr[i].out_offset = off
        off += l.n_keep * l.m_keep
    return arr


def window_geometry(win_rows, win_counts, overlap: int = 200):
    """(seq_win, stitched rows per sequence) of sequences given as windows: ``dctfp_stitch_sizes`` (the reference's
    ``run[-olp:] = (run[-olp:] + new[:olp]) / 2; cat(new[olp:])``, src/embedding.py:185-187).  ValueError where torch would
    fail to broadcast there."""
    win_rows = np.ascontiguousarray(np.asarray(win_rows, dtype=np.int32))
    counts = np.asarray(win_counts, dtype=np.int64)
    seq_win = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(counts, out=seq_win[1:])
    if seq_win[-1] != len(win_rows):
        raise ValueError(f'{len(win_rows)} windows, the sequences count {int(seq_win[-1])}')
    sizes = np.empty(len(counts), dtype=np.int64)
    lib = _lib.load()
    rc = lib.dctfp_stitch_sizes(win_rows.ctypes.data, seq_win.ctypes.data, len(counts), int(overlap), 0, sizes.ctypes.data)
    _lib.check(rc, lib)
    return seq_win, sizes


def quantize_windows(layers: Sequence[LayerBatch], win_rows, win_counts, table: PieceTable, overlap: int = 200,
                     out: torch.Tensor = None, ctx: _lib.Context = None, stream=None, fallback: bool = True) -> torch.Tensor:
    """``Embedding.embed_seq`` + ``Fingerprint.quantize`` without the stitched matrix (``dctfp_quantize_windows``,
    include/dctfp.h): every ``LayerBatch`` holds one float32 matrix per WINDOW (window w of sequence s at index
    ``sum(win_counts[:s]) + w``), ``table`` speaks of the stitched sequences (``window_geometry`` gives their rows).  Rows two
    windows share are averaged in the kernel's row load; the result is what ``stitch_embeddings_batch`` + ``quantize_batch``
    give, byte for byte.  Calls the one-launch kernel does not take (DCTFP_ERR_UNSUPPORTED: other kept sizes, half-precision
    rows, a handful of jobs, ...) are stitched first when ``fallback`` is set -- torch allocates the stitched matrices."""
    if not layers:
        raise ValueError('no layers')
    device = layers[0].device
    total = sum(l.n_keep * l.m_keep for l in layers)
    if out is None:
        out = torch.empty((table.n_domains, total), dtype=torch.int8, device=device)
    elif out.dtype != torch.int8 or out.dim() != 2 or out.shape[0] < table.n_domains or out.shape[1] < total \
            or out.stride(1) != 1 or out.device != device:
        raise ValueError('out must be an int8 (n_domains, >= sum n*m) tensor on the layers\' device')
    if table.n_domains == 0:
        return out
    if ctx is None:
        ctx = _lib.get_context(device.index if device.index is not None else torch.cuda.current_device())
    win_rows = np.ascontiguousarray(np.asarray(win_rows, dtype=np.int32))
    seq_win, sizes = window_geometry(win_rows, win_counts, overlap)
    if len(sizes) != len(table.seq_rows) or (sizes != table.seq_rows).any():
        raise ValueError('the piece table was not built for the stitched lengths of these windows (window_geometry)')
    arr = _layer_array(layers, len(win_rows))
    if stream is None:
        stream = torch.cuda.current_stream(device)
    rc = ctx._lib.dctfp_quantize_windows(ctx.handle, arr, len(layers), len(sizes), seq_win.ctypes.data, win_rows.ctypes.data,
                                         int(overlap), table.pieces.ctypes.data, len(table.pieces), table.n_domains,
                                         out.data_ptr(), out.stride(0), C.c_void_p(stream.cuda_stream))
    if rc == _lib.DCTFP_ERR_UNSUPPORTED and fallback:
        from .embedding import stitch_windows_flat
        stitched = []
        for l in layers:
            big, first = stitch_windows_flat(l, win_rows, seq_win, sizes, overlap)
            stitched.append(LayerBatch(big, l.n_keep, l.m_keep, row_offsets=first))
        return quantize_batch(stitched, table, out=out, ctx=ctx, stream=stream)
    _lib.check(rc, ctx._lib)
    return out


def quantize_batch(layers: Sequence[LayerBatch], table: PieceTable, out: torch.Tensor = None,
                   ctx: _lib.Context = None, stream=None) -> torch.Tensor:
    """Runs ``dctfp_quantize`` (include/dctfp.h) and returns the int8 tensor
    ``(table.n_domains, sum n_i*m_i)`` on the layers' device.  Asynchronous with respect
    to the host: the result is ordered on the given / current torch stream.  ``out`` may be a
    pinned host tensor: then the result lands in host memory without a copy (valid once the
    stream has been synchronised)."""
    if not layers:
        raise ValueError('no layers')
    device = layers[0].device
    total = sum(l.n_keep * l.m_keep for l in layers)
    out_ptr = None
    if out is None:
        out = torch.empty((table.n_domains, total), dtype=torch.int8, device=device)
    elif out.dtype != torch.int8 or out.dim() != 2 or out.shape[0] < table.n_domains or out.shape[1] < total \
            or out.stride(1) != 1:
        raise ValueError('out must be an int8 (n_domains, >= sum n*m) tensor')
    elif out.device.type == 'cpu':
        # a pinned host tensor: the kernels write the result over PCIe themselves (small calls; see host_pointer)
        out_ptr = host_pointer(out)
    elif out.device != device:
        raise ValueError('out must live on the layers\' device (or be a pinned host tensor)')
    if table.n_domains == 0:
        return out
    if ctx is None:
        ctx = _lib.get_context(device.index if device.index is not None else torch.cuda.current_device())
    n_seq = len(table.seq_rows)
    arr = _layer_array(layers, n_seq)
    if stream is None:
        stream = torch.cuda.current_stream(device)
    rc = ctx._lib.dctfp_quantize(ctx.handle, arr, len(layers), n_seq, table.seq_rows.ctypes.data,
                                 table.pieces.ctypes.data, len(table.pieces), table.n_domains,
                                 out.data_ptr() if out_ptr is None else out_ptr, out.stride(0), C.c_void_p(stream.cuda_stream))
    _lib.check(rc, ctx._lib)
    return out
