"""``Fingerprint`` -- drop-in for mgtools/DCTdomain's ``src/fingerprint.py`` class
(:17-201) with the iDCT quantisation running on MI355X HIP kernels.

Same fields, same methods, same results:

* ``quantize(qdim)``     src/fingerprint.py:174-201  -> ``dctfp_quantize``   (hot path)
* ``idct_quant(vec, n)`` src/fingerprint.py:126-142  -> ``dctfp_idct_quant``
* ``scale(vec)``         src/fingerprint.py:110-123  -> ``dctfp_scale``
* ``get_doms(emb, dom)`` src/fingerprint.py:145-171  -> ``dctfp_gather_rows`` (+ host string rules)
* ``writece`` / ``reccut`` src/fingerprint.py:45-107 -> see ``dctdomain_amd.reccut``

``embed`` values may be numpy arrays (as in the reference; copied to the GPU once) or
torch tensors already on the GPU straight from the ESM-2 forward pass (new).  There is no
CPU implementation behind this class: without libdctfp.so and a GPU it raises.
"""

from __future__ import annotations

import ctypes as C
import logging
import threading
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from .batch import PieceTable, host_pointer
from .domains import split_domain


CONSTANT_CHANNEL_NOTE = (
    'an embedding channel is exactly constant over a domain: its min-max scale is 0/0, so the (layer, domain) block of '
    'the fingerprint is all 0.  The reference (scipy/pocketfft) gives the same block at most domain lengths but scales '
    'its own round-off noise at 225 of the lengths 3..2000 (tests/golden/fence_golden.json, INTEGRATION.md section 5): '
    'for these blocks the result is reported, not matched')


def warn_constant_channel(pids):
    """The one documented deviation from the reference must not be silent in the drop-in: a warning naming the proteins
    (``make_db --out`` puts it into the log file)."""
    logging.warning(f"constant channel in {', '.join(str(p) for p in pids)}: {CONSTANT_CHANNEL_NOTE}")


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('dctdomain_amd needs an MI355X GPU (torch.cuda.is_available() is False); '
                           'there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _to_device_matrix(x, device=None, keep_half: bool = False) -> torch.Tensor:
    """numpy / torch 2-D matrix -> contiguous tensor on the GPU in a storage type the kernels read
    (float32 / float64; float16 / bfloat16 stay as they are for ``quantize`` when ``keep_half``).
    Every dtype the reference accepts is promoted exactly (it computes in float64)."""
    if isinstance(x, torch.Tensor):
        t = x
        if t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and (t.dtype == torch.float32 or t.dtype == torch.float64) \
                and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1]):
            return t                             # straight off the language model: nothing to do
    else:
        a = np.asarray(x)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float32 if a.dtype in (np.float16,) else np.float64)
        t = torch.from_numpy(np.ascontiguousarray(a))
    if t.dim() != 2:
        raise ValueError(f'expected a 2-D matrix, got shape {tuple(t.shape)}')
    if t.dtype in (torch.float16, torch.bfloat16):
        if not keep_half:
            t = t.to(torch.float32)
    elif t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    if t.device.type != 'cuda':
        t = t.to(device if device is not None else _device())
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _like_input(result: torch.Tensor, like):
    """numpy in -> numpy out (the reference's types); tensor in -> tensor out."""
    if isinstance(like, torch.Tensor):
        return result
    return result.cpu().numpy()


@dataclass
class Fingerprint:
    """Carrier of one protein: embeddings, contact map, predicted domains and their
    int8 fingerprints.  Attributes as in the reference (src/fingerprint.py:30-35)."""
    pid: str = field(default_factory=str)
    seq: str = field(default_factory=str)
    embed: dict = field(default_factory=dict)
    contacts: np.array = field(default_factory=list)
    domains: list = field(default_factory=list)
    quants: dict = field(default_factory=dict)

    def __post_init__(self):
        if not isinstance(self.contacts, torch.Tensor):
            self.contacts = np.array(self.contacts)

    # -- domain prediction (src/fingerprint.py:45-107) ---------------------------------
    def writece(self, outfile: str, t: float):
        from .reccut import write_ce
        write_ce(self, outfile, t)

    def reccut(self, threshold: float):
        from .reccut import predict_domains
        domains = predict_domains(self, threshold)
        for dom in domains:
            self.domains.append(dom)
        if len(domains) > 1:
            self.domains.append(f'1-{len(self.seq)}')

    # -- helpers of quantize -------------------------------------------------------------
    def scale(self, vec):
        """(vec - min) / (max - min); max == min gives NaN (src/fingerprint.py:110-123)."""
        if isinstance(vec, torch.Tensor):
            t = vec.to(torch.float64)
            if t.device.type != 'cuda':
                t = t.to(_device())
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(vec, dtype=np.float64))).to(_device())
        t = t.contiguous()
        flat = t.reshape(-1)
        out = torch.empty_like(flat)
        ctx = _lib.get_context(t.device.index)
        stream = torch.cuda.current_stream(t.device)
        _lib.check(ctx._lib.dctfp_scale(ctx.handle, flat.data_ptr(), flat.numel(), out.data_ptr(),
                                        C.c_void_p(stream.cuda_stream)))
        return _like_input(out.reshape(t.shape), vec)

    def idct_quant(self, vec, num: int):
        """DCT-II along axis 0, keep ``num``, inverse DCT of length ``num``, min-max scale per
        column; returns (num, n_cols) float64 (src/fingerprint.py:126-142)."""
        t = _to_device_matrix(vec)
        n_rows, n_cols = t.shape
        k = min(int(num), n_rows)      # f[:, :num] cannot be wider than the transform
        out = torch.empty((k, n_cols), dtype=torch.float64, device=t.device)
        ctx = _lib.get_context(t.device.index)
        stream = torch.cuda.current_stream(t.device)
        _lib.check(ctx._lib.dctfp_idct_quant(ctx.handle, t.data_ptr(), _dtype_code(t), n_rows, n_cols,
                                             t.stride(0) if n_rows > 1 else n_cols, k, out.data_ptr(), None,
                                             C.c_void_p(stream.cuda_stream)))
        return _like_input(out, vec)

    def dct_coefficients(self, vec, num: int):
        """``f[:, :num]`` of src/fingerprint.py:137 -- (n_cols, num) float64.  Not in the
        reference's API; exposes the intermediate coefficients for parity checks."""
        t = _to_device_matrix(vec)
        n_rows, n_cols = t.shape
        k = min(int(num), n_rows)
        coef = torch.empty((n_cols, k), dtype=torch.float64, device=t.device)
        ctx = _lib.get_context(t.device.index)
        stream = torch.cuda.current_stream(t.device)
        _lib.check(ctx._lib.dctfp_idct_quant(ctx.handle, t.data_ptr(), _dtype_code(t), n_rows, n_cols,
                                             t.stride(0) if n_rows > 1 else n_cols, k, None, coef.data_ptr(),
                                             C.c_void_p(stream.cuda_stream)))
        return _like_input(coef, vec)

    def get_doms(self, embed, dom: str) -> tuple:
        """Rows of a (possibly discontinuous) domain as a float64 matrix + the cleaned
        domain string (src/fingerprint.py:145-171, quirks in ``domains.split_domain``)."""
        t = _to_device_matrix(embed)
        n_rows, n_cols = t.shape
        pieces, key = split_domain(dom, n_rows)
        total = sum(n for _, n in pieces)
        out = torch.empty((total, n_cols), dtype=torch.float64, device=t.device)
        if total:
            rec = np.zeros(len(pieces), dtype=_lib.PIECE_DTYPE)
            rec['row_start'] = [p[0] for p in pieces]
            rec['n_rows'] = [p[1] for p in pieces]
            ctx = _lib.get_context(t.device.index)
            stream = torch.cuda.current_stream(t.device)
            _lib.check(ctx._lib.dctfp_gather_rows(ctx.handle, t.data_ptr(), _dtype_code(t), n_rows, n_cols,
                                                  t.stride(0) if n_rows > 1 else n_cols, rec.ctypes.data, len(rec),
                                                  out.data_ptr(), C.c_void_p(stream.cuda_stream)))
        return _like_input(out, embed), key

    # -- the hot path ----------------------------------------------------------------------
    def quantize(self, qdim: list):
        """iDCT quantisation of every domain of every layer (src/fingerprint.py:174-201).

        ``qdim[2i], qdim[2i+1]`` = kept points along the sequence / channel axis of layer
        ``i`` (dict order).  Afterwards ``quants[dom]`` is an integer array of
        ``sum n_i*m_i`` values 0..127 and ``domains == list(quants)``.  Raises
        ``ValueError`` where the reference's reshape does (a domain shorter than n, a layer
        narrower than m) -- before touching ``quants``.

        One protein per call is the reference's calling pattern, not the fast one: a call costs a
        table upload, one or two kernel launches and a device -> host copy of 480 bytes per domain
        (about 70 us in all).  ``make_db.fingerprint_batch`` / ``batch.quantize_batch`` take many
        proteins per call and are the throughput path."""
        embeds = list(self.embed.values())
        if len(qdim) < 2 * len(embeds):
            raise IndexError('list index out of range')      # qdim[i*2] in the reference
        device = None
        for e in embeds:
            if isinstance(e, torch.Tensor) and e.device.type == 'cuda':
                device = e.device
                break
        mats = [_to_device_matrix(e, device, keep_half=True) for e in embeds]
        if mats and self._quantize_one(mats, qdim):
            return

        # layers are grouped while they share the row count (one piece table per group); every group is
        # enqueued first, the results land in ONE pinned buffer and are read after one synchronisation.
        # One protein per call is bound by host time: the C ABI is called directly (no LayerBatch / quantize_batch
        # objects), on torch's current raw stream, and waited for with dctfp_stream_synchronize.
        groups = []           # (table, offset into the pinned buffer, width, first layer, last layer)
        dev = mats[0].device if mats else None
        stream = _raw_stream(dev) if mats else 0
        ctx = _lib.get_context(dev.index if dev.index is not None else torch.cuda.current_device()) if mats else None
        if ctx is not None:
            # the constant-channel flag belongs to the context (include/dctfp.h): whatever an earlier caller left unread
            # (quantize_batch users, tools) is not about THIS protein
            ctx.get_option('degenerate_seen')
        i = 0
        while i < len(mats):
            j = i
            while j + 1 < len(mats) and mats[j + 1].shape[0] == mats[i].shape[0] \
                    and mats[j + 1].device == mats[i].device:
                j += 1
            table = PieceTable([mats[i].shape[0]], [self.domains])
            if table.n_domains:
                width = sum(qdim[2 * k] * qdim[2 * k + 1] for k in range(i, j + 1))
                buf, off, out_ptr = _result_slot(table.n_domains * width)
                _enqueue_group(mats[i:j + 1], qdim[2 * i:2 * j + 2], table, out_ptr, width, stream)
                groups.append((table, (buf, off), width, i, j))
            i = j + 1
        hosts = _fetch_results(groups, stream)
        if groups and ctx.get_option('degenerate_seen'):
            warn_constant_channel([self.pid])

        # quants[key] = the blocks of every layer, layer-major, then domain order (:184-196); a key that occurs
        # twice (two domain strings cleaned to the same key) is extended twice, as there
        pieces = {}
        for key, value in self.quants.items():
            pieces[key] = [np.asarray(value)]
        for (table, _, _, first, last), host in zip(groups, hosts):
            off = 0
            for k in range(first, last + 1):
                nm = qdim[2 * k] * qdim[2 * k + 1]
                for row, key in enumerate(table.keys):
                    pieces.setdefault(key, []).append(host[row, off:off + nm])
                off += nm
        for key, parts in pieces.items():
            self.quants[key] = np.concatenate(parts).astype(np.int64) if len(parts) > 1 or parts[0].dtype != np.int64 \
                else parts[0]
        self.domains = list(self.quants.keys())


    def _quantize_one(self, mats, qdim) -> bool:
        """The common shape of a call -- every layer the same rows on one GPU, plain ``b-e[,b-e]`` domain strings -- through ONE
        foreign call (``dctfp_quantize_one``: strings -> pieces -> kernels -> wait; include/dctfp.h).  False: not that shape,
        nothing was launched, the general path below takes the call."""
        k, doms = len(mats), self.domains
        st = _ONE
        if k > st.MAX_LAYERS or len(doms) > st.MAX_STRINGS or not doms:
            return False
        t0 = mats[0]
        rows, dev = t0.shape[0], t0.device
        for t in mats:
            if t.shape[0] != rows or t.device != dev:
                return False
        try:
            text = '\n'.join(doms).encode('ascii')
        except (UnicodeEncodeError, TypeError):
            return False
        if len(text) + len(doms) + 1 > st.KEY_CAP:
            return False
        arr, ptrs = st.arr, st.ptrs
        width = 0
        for i, t in enumerate(mats):
            ptrs[i] = t.data_ptr()
            a = arr[i]
            a.ld = t.stride(0) if rows > 1 else t.shape[1]
            a.n_cols = t.shape[1]
            a.dtype = _DTYPE_CODE[t.dtype]
            a.n_keep = n = int(qdim[2 * i])
            a.m_keep = m = int(qdim[2 * i + 1])
            a.out_offset = width
            width += n * m
        n_str = len(doms)
        buf, off, out_ptr = _result_slot(n_str * width)
        ctx = _lib.get_context(dev.index if dev.index is not None else torch.cuda.current_device())
        rc = ctx._lib.dctfp_quantize_one(ctx.handle, arr, k, rows, text, len(text), n_str, out_ptr, n_str, width, st.str_row_p, st.changed_p,
                                        st.key_p, st.KEY_CAP, st.key_len_p, st.n_dom_p, st.n_other_p, st.flag_p, _raw_stream(dev))
        _RESULTS.used = 0
        if rc != 0:
            _lib.check(rc, ctx._lib)          # ValueError where the reference's reshape fails -- before quants is touched
        if st.n_other.value:
            return False                      # (a string for Python's own int() / split: the general path)
        nd = st.n_dom.value
        if st.flag.value:
            warn_constant_channel([self.pid])
        if nd:
            host = buf[off:off + nd * width].reshape(nd, width)
            changed = st.changed[:n_str]
            cleaned = iter(st.key_buf.raw[:st.key_len.value].decode('ascii').split('\n')) if st.key_len.value else None
            keys = []
            for i in range(n_str):             # (every string with a piece removed has its cleaned key in key_text, kept or not)
                key = next(cleaned) if changed[i] == 1 else doms[i]
                if st.str_row[i] >= 0:
                    keys.append(key)
            if not self.quants and len(set(keys)) == nd:
                host64 = host.astype(np.int64)             # a row already is layer 0's block, layer 1's block, ... (:184-196)
                self.quants = dict(zip(keys, host64))
            else:                                          # quants already holds entries / a key twice: extended layer by layer, as there
                held = {key: [np.asarray(v)] for key, v in self.quants.items()}
                o = 0
                for i in range(k):
                    nm = int(qdim[2 * i]) * int(qdim[2 * i + 1])
                    for r, key in enumerate(keys):
                        held.setdefault(key, []).append(host[r, o:o + nm])
                    o += nm
                for key, parts in held.items():
                    self.quants[key] = np.concatenate(parts).astype(np.int64) if len(parts) > 1 or parts[0].dtype != np.int64 else parts[0]
        self.domains = list(self.quants.keys())
        return True


class _OneCall(threading.local):
    """Per thread: the argument blocks of ``dctfp_quantize_one`` (filled in place call after call)."""
    MAX_LAYERS, MAX_STRINGS, KEY_CAP = 8, 64, 8192

    def __init__(self):
        self.arr = (_lib.Layer * self.MAX_LAYERS)()
        self.ptrs = (C.c_void_p * self.MAX_LAYERS)()
        for i in range(self.MAX_LAYERS):
            self.arr[i].seq_data = C.cast(C.byref(self.ptrs, i * C.sizeof(C.c_void_p)), C.POINTER(C.c_void_p))
        self.str_row = (C.c_int32 * self.MAX_STRINGS)()
        self.changed = (C.c_uint8 * self.MAX_STRINGS)()
        self.key_buf = C.create_string_buffer(self.KEY_CAP)
        self.key_len, self.n_dom, self.n_other, self.flag = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        self.str_row_p, self.changed_p, self.key_p = C.addressof(self.str_row), C.addressof(self.changed), C.addressof(self.key_buf)
        self.key_len_p, self.n_dom_p, self.n_other_p, self.flag_p = (C.addressof(x) for x in (self.key_len, self.n_dom, self.n_other, self.flag))


_ONE = _OneCall()


# scratch of the one-protein-per-call path: a pinned host buffer the kernels write their int8 results into directly
# (480 bytes per domain over PCIe: no device buffer, no copy engine in the chain -- its latency was most of a call)
class _Results(threading.local):
    """Per thread: the pinned buffer, its numpy view, the address the GPU sees it under, bytes handed out."""
    pin = None
    np = None
    dev = 0
    used = 0


_RESULTS = _Results()


def _raw_stream(device) -> int:
    """torch's current stream on ``device`` as a hipStream_t value."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    try:
        return int(torch._C._cuda_getCurrentRawStream(idx))
    except AttributeError:                                   # (older / newer torch without the private accessor)
        return int(torch.cuda.current_stream(device).cuda_stream)


def _result_slot(nbytes: int):
    """``nbytes`` of the pinned result buffer: (numpy view of the buffer, offset, address under which the GPU sees the
    slot).  Several slots may be handed out between two ``_fetch_results`` calls (layer groups of one protein): they are
    carved one after another; a slot keeps its buffer alive through the numpy view if a larger one replaces it."""
    st = _RESULTS
    if st.pin is None or st.used + nbytes > st.pin.numel():
        st.pin = torch.empty(max(1 << 16, 2 * nbytes), dtype=torch.int8, pin_memory=True)
        st.np = st.pin.numpy()
        st.dev = host_pointer(st.pin)
        st.used = 0
    off = st.used
    st.used += nbytes
    return st.np, off, st.dev + off


def _enqueue_group(mats, qd, table, out_ptr: int, width: int, stream: int):
    """``dctfp_quantize`` (include/dctfp.h) for the layers of one protein that share the row count."""
    k = len(mats)
    arr = (_lib.Layer * k)()
    ptrs = (C.c_void_p * k)()
    off = 0
    for i, t in enumerate(mats):
        ptrs[i] = t.data_ptr()
        a = arr[i]
        a.seq_data = C.cast(C.byref(ptrs, i * C.sizeof(C.c_void_p)), C.POINTER(C.c_void_p))
        a.ld = t.stride(0) if t.shape[0] > 1 else t.shape[1]
        a.n_cols = t.shape[1]
        a.dtype = _DTYPE_CODE[t.dtype]
        a.n_keep = int(qd[2 * i])
        a.m_keep = int(qd[2 * i + 1])
        a.out_offset = off
        off += a.n_keep * a.m_keep
    d = mats[0].device
    ctx = _lib.get_context(d.index if d.index is not None else torch.cuda.current_device())
    lib = ctx._lib
    _lib.check(lib.dctfp_quantize(ctx.handle, arr, k, 1, table.seq_rows.ctypes.data, table.pieces.ctypes.data,
                                  len(table.pieces), table.n_domains, out_ptr, width, stream))


def _fetch_results(groups, stream: int):
    """Slots handed out by ``_result_slot`` -> numpy arrays (copies) after ONE stream synchronisation."""
    if not groups:
        return []
    _lib.check(_lib.load().dctfp_stream_synchronize(stream))
    out = [buf[off:off + table.n_domains * width].reshape(table.n_domains, width).copy() for table, (buf, off), width, _, _ in groups]
    _RESULTS.used = 0
    return out


_DTYPE_CODE = {torch.float32: _lib.DCTFP_F32, torch.float64: _lib.DCTFP_F64, torch.float16: _lib.DCTFP_F16,
               torch.bfloat16: _lib.DCTFP_BF16}


def _dtype_code(t: torch.Tensor) -> int:
    return _lib.DCTFP_F32 if t.dtype == torch.float32 else _lib.DCTFP_F64
