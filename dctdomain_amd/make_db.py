"""Drop-in for mgtools/DCTdomain ``src/make_db.py``: FASTA -> SQLite ``.db`` (+ ``.index``,
``-dct.npz``, ``.dom``) with the same command line (:167-177) and the same output layout, the
fingerprinting running on MI355X.

    python -m dctdomain_amd.make_db --fafile X.fasta --dbfile X [--maxlen 500] [--cpu N] [--gpu G]
                                    [--noindex] [--nonpz] [--nodom] [--out LOG] [--model esm|synthetic]

What changes underneath: the reference embeds on the GPU, copies every embedding to
the host, pickles it into a ``multiprocessing.Pool`` and fingerprints one protein per worker call
(:36-51).  Here the embeddings stay on the GPU; a flush of many proteins is fingerprinted by one
batched contact top-k, one threaded in-process RecCut call and one ``dctfp_quantize`` launch
(``fingerprint_batch``; round 5: the domain cutter's recursion runs on the GPU too, ``dctfp_reccut``).  ``--cpu`` sizes the thread pool of
the host cutter (what the GPU one hands back: proteins above 2 048 residues), ``--gpu G`` starts one process
per GPU over disjoint, length-balanced shards; a single writer (the parent) fills the database in
sequence order, one transaction per flush (``OrderedWriter``), so the files are identical for every G and an
interrupted build resumes where it stopped (``fpcount``, src/database.py:150, :211-213).
"""

from __future__ import annotations

import argparse
import ctypes as C
import datetime
import logging
import os
import threading
from typing import List

import numpy as np
import torch

from .database import Database
from .fingerprint import Fingerprint

import time
from collections import defaultdict
from contextlib import contextmanager

#: wall seconds per stage of the last ``run`` in this process (workers report theirs to the parent, which keeps the
#: slowest worker's figure per stage): a build at scale says where its time went (profiles/r03/db_build_1M.txt)
STAGE_SECONDS = defaultdict(float)


@contextmanager
def stage(name: str):
    t0 = time.perf_counter()
    try:
        yield
    finally:
        STAGE_SECONDS[name] += time.perf_counter() - t0


from operator import attrgetter as _attrgetter
_IS_CUDA, _DTYPE = _attrgetter('is_cuda'), _attrgetter('dtype')
_KEPT_DTYPES = {torch.float32, torch.float64, torch.float16, torch.bfloat16}

#: set to a list and ``fingerprint_batch`` appends (name, perf_counter()) at its stage boundaries (tools/flush_timeline.py)
MARKS = None


def _mark(name: str):
    if MARKS is not None:
        MARKS.append((name, time.perf_counter()))


LAYERS = [15, 21]                 # src/make_db.py:78, :138
QDIM = [3, 80, 3, 80]             # src/make_db.py:30
THRESHOLD = 2.6                   # src/make_db.py:29


def queue_cpu(fp: Fingerprint) -> Fingerprint:
    """One protein: predict domains, quantise (reference ``queue_cpu``, src/make_db.py:19-33)."""
    fp.reccut(THRESHOLD)
    fp.quantize(QDIM)
    logging.info(f'{datetime.datetime.now()} Fingerprinted {fp.pid}')
    return fp


def fingerprint_batch(fps: List[Fingerprint], threads: int = 1, qdim=QDIM, threshold: float = THRESHOLD):
    """``queue_cpu`` for many proteins at once: same ``domains`` / ``quants`` per object as calling it one by one
    (src/make_db.py:19-33 inside the pool of :36-51).  The flush as ``_Flush`` runs it -- geometry of all tensors in one pass,
    contact selection + domain cutter on the GPU, strings and piece table straight from the cutter's integers, one
    ``dctfp_quantize``, the objects filled while the kernels run; anything out of the ordinary (domains or fingerprints already
    in an object, embeddings that are not GPU tensors of one layout, a protein whose domain strings need Python's own parser)
    goes through ``_fingerprint_batch_generic``, which takes every input the reference takes."""
    if not fps:
        return fps
    fl = _Flush(fps, threads, qdim, threshold)
    if fl.start():
        if fl.finish(objects=True) is not None:
            LAST_PATH[0] = 'flush'
            return fps
    LAST_PATH[0] = 'generic'
    return _fingerprint_batch_generic(fps, threads, qdim, threshold)


#: which way the last ``fingerprint_batch`` / ``flush_records`` of this process went ('flush' or 'generic'): tests assert it
LAST_PATH = [None]


_SEQ, _CONTACTS, _EMBED, _DOMAINS, _QUANTS, _PID = (_attrgetter(a) for a in ('seq', 'contacts', 'embed', 'domains', 'quants', 'pid'))
_SIDE_STREAMS = {}
#: which of the two sets of page-locked result buffers the next flush takes (two flushes are alive at a time in process_sequences: the one
#: whose cutter runs and the one being finished; the buffers themselves are per thread, reccut._pinned)
_FLUSH_SLOT = [0]


def _side_stream(device):
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = _SIDE_STREAMS[device] = torch.cuda.Stream(device)
    return s


class _Flush:
    """One flush of a database build in two halves, so that the build can do something else in between
    (``process_sequences`` embeds the next proteins; the domain cutter is a latency, not a load: a few long proteins, one
    workgroup each):

    ``start()``   geometry of every tensor of the flush in one pass (``_geom``), contact selection + domain cutter enqueued on a
                  side stream, results on their way into page-locked memory; the embedding tables while the GPU works.
    ``finish()``  the cutter's integers -> strings + piece table (``dctfp_reccut_pieces``), one ``dctfp_quantize``, the
                  per-protein results assembled while the kernels run.  ``objects=True`` fills ``domains`` / ``quants`` of every
                  ``Fingerprint`` as ``queue_cpu`` does (int64 rows, the reference's dtype) and returns the list;
                  ``objects=False`` returns what the writer stores -- (pid, domains, int8 rows) -- and leaves the objects alone.

    Both return None / False where the flush is not of the plain kind; the caller then runs ``_fingerprint_batch_generic``."""

    def __init__(self, fps, threads=1, qdim=QDIM, threshold=THRESHOLD):
        self.fps, self.threads, self.qdim, self.threshold = fps, max(1, threads), list(qdim), threshold
        self.cut = None

    def start(self) -> bool:
        from . import _geom, reccut
        from .batch import LayerBatch
        fps = self.fps
        n = len(fps)
        _mark('start')
        if any(map(_DOMAINS, fps)) or any(map(_QUANTS, fps)):
            return False                       # (domains given by the caller, or a second quantize: the general bookkeeping)
        seqs = list(map(_SEQ, fps))
        lens = np.fromiter(map(len, seqs), dtype=np.int64, count=n)
        cts = list(map(_CONTACTS, fps))
        ptrs, meta = _geom.tensor_table(cts)
        first = cts[0]
        if not (torch.is_tensor(first) and first.is_cuda and first.dtype == torch.float32):
            return False
        side = lens > 1
        if not ((meta[:, _geom.DIM] == 2).all() and (meta[:, _geom.CODE] == meta[0, _geom.CODE]).all()
                and (meta[:, _geom.SIZE0] == lens).all() and (meta[:, _geom.SIZE1] == lens).all()
                and (meta[side, _geom.STRIDE1] == 1).all() and (lens >= 1).all() and lens.max() < (1 << 31)):
            return False                       # (maps given as numpy / flat / transposed: _contact_tensor judges them one by one)
        device = first.device
        lds = np.where(side, meta[:, _geom.STRIDE0], np.maximum(lens, 1))
        self.lens = lens
        keys0 = list(fps[0].embed.keys())
        if not keys0:
            return False
        embeds = list(map(_EMBED, fps))
        _mark('maps')
        ctx = _lib_mod().get_context(device.index)
        self.ctx, self.device = ctx, device
        slot = _FLUSH_SLOT[0] = _FLUSH_SLOT[0] ^ 1
        main = torch.cuda.current_stream(device)
        stream = _side_stream(device)
        stream.wait_stream(main)               # (the maps were written on the caller's stream)
        self.cut = reccut.CutInFlight(ptrs, lds, lens.astype(np.int32), device, self.threshold, stream=stream, slot=slot,
                                      timing=MARKS is not None)
        self._maps = cts
        _mark('enqueued top-k + cutter')
        # ---- the embedding tables, while the GPU selects and cuts
        layers, mats = [], []
        try:
            for i, k in enumerate(keys0):
                vals = [e[k] for e in embeds]
                p, m = _geom.tensor_table(vals)
                if not (m[:, _geom.SIZE0] == lens).all():
                    return self._abandon()     # (rows of an embedding differ from the sequence's length: the general path names it)
                t0 = vals[0]
                if not (t0.is_cuda and t0.dtype in _KEPT_DTYPES and t0.device == device):
                    return self._abandon()
                layers.append(LayerBatch.from_table(vals, p, m, self.qdim[2 * i], self.qdim[2 * i + 1]))
                mats.append(vals)
        except (ValueError, KeyError, AttributeError, TypeError, IndexError):
            return self._abandon()
        self.layers, self.mats, self.keys0 = layers, mats, keys0
        _mark('embedding tables')
        return True

    def _abandon(self):
        if self.cut is not None:
            self.cut.done.synchronize()        # (its buffers are this thread's: nothing may still be writing them)
            self.cut = None
        return False

    def finish(self, objects: bool):
        from .batch import PieceTable, quantize_batch
        fps, n, lens = self.fps, len(self.fps), self.lens
        lib = _lib_mod().load()
        enc = self.cut.wait()
        _mark('cutter waited for')
        enc_off = self.cut.enc_off
        n_enc = int(enc_off[-1])
        # (segments + proteins bound the pieces; 24 bytes per segment + 32 per protein the text: an encoded record spends two
        #  ints per segment)
        piece_cap = n_enc // 2 + n + 1
        text_cap = 12 * n_enc + 32 * n + 64
        pieces = np.empty(piece_cap, dtype=_lib_mod().PIECE_DTYPE)
        text = np.empty(text_cap, dtype=np.uint8)
        counts = np.empty(n, dtype=np.int32)
        text_len, n_pieces, n_dom, n_undone = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        enc = np.ascontiguousarray(enc)
        _lib_mod().check(lib.dctfp_reccut_pieces(n, enc.ctypes.data, enc_off.ctypes.data, lens.ctypes.data, text.ctypes.data, text_cap,
                                                C.byref(text_len), counts.ctypes.data, pieces.ctypes.data, piece_cap, C.byref(n_pieces),
                                                C.byref(n_dom), C.byref(n_undone)), lib)
        if n_undone.value:
            return None                        # (a protein whose strings Python's own parser must judge: never seen; the general path)
        flat = text[:text_len.value].tobytes().decode('ascii').split(';')
        flat.pop()
        nd = n_dom.value
        table = PieceTable.from_pieces(lens, pieces[:n_pieces.value], nd, flat, counts)
        _mark('strings + piece table')
        total = sum(l.n_keep * l.m_keep for l in self.layers)
        self.ctx.get_option('degenerate_seen')  # (the flag is the context's: drop what earlier callers left unread)
        # On the flush's own stream, not the caller's: inside a build the caller's stream still holds the language model's kernels for
        # the NEXT proteins, and this flush's embeddings were complete before its cutter started (which waited for the caller's stream,
        # and has been waited for above) -- the fingerprint kernel runs beside them instead of behind them.
        stream = self.cut.stream
        with torch.cuda.stream(stream):
            out = quantize_batch(self.layers, table, ctx=self.ctx)
            pin = getattr(_PINNED, 'buf', None)
            if pin is None or pin.numel() < nd * total:
                pin = _PINNED.buf = torch.empty(max(nd * total + nd * total // 4, 1 << 20), dtype=torch.int8, pin_memory=True)
            view = pin[:nd * total].view(nd, total)
            view.copy_(out, non_blocking=True)
        _mark('quantize enqueued')
        # ---- everything that needs only the SHAPE of the result, while the kernels run
        bounds = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=bounds[1:])
        bl = bounds.tolist()
        host8 = np.empty((nd, total), dtype=np.int8)
        result = None
        if objects:
            host64 = np.empty((nd, total), dtype=np.int64)      # np.array(list of ints) in the reference: int64
            rows64 = list(host64)                               # (row views, made in one C-level pass)
            dup = []
            for s, fp in enumerate(fps):
                a, b = bl[s], bl[s + 1]
                ks = flat[a:b]
                q = dict(zip(ks, rows64[a:b]))                  # a row already is layer 0's block, layer 1's block, ... (:184-196)
                if len(q) != b - a:                             # (a key twice: never from the cutter -- its domains are disjoint)
                    dup.append(s)
                    continue
                fp.quants = q
                fp.domains = ks
                fp._rows8 = host8[a:b]                          # what the writer stores (see _records)
            result = fps
        else:
            result = [(pid, flat[a:b], host8[a:b]) for pid, a, b in zip(map(_PID, fps), bl[:-1], bl[1:])]
        _mark('results laid out')
        stream.synchronize()
        np.copyto(host8, view.numpy())
        if objects:
            _widen(host64, host8)
            for s in dup:
                a, b = bl[s], bl[s + 1]
                fps[s].domains.extend(flat[a:b])
                _extend_quants(fps[s], flat[a:b], host64[a:b], self.qdim, len(self.keys0))
        _mark('results on the host')
        if nd and self.ctx.get_option('degenerate_seen'):
            # rare: the flush saw an exactly constant channel (0/0 -> all-zero block, the documented deviation).  Name the proteins.
            from .fingerprint import warn_constant_channel
            warn_constant_channel(_constant_channel_pids(fps, self.mats, table) or [fp.pid for fp in fps])
        if logging.getLogger().isEnabledFor(logging.INFO):      # one line per protein, as the reference writes them
            now = datetime.datetime.now()
            logging.info('\n'.join(f'{now} Fingerprinted {fp.pid}' for fp in fps))
        self.cut = None
        return result


#: the two halves of a flush for callers outside this module
Flush = _Flush


def _lib_mod():
    from . import _lib
    return _lib


_WIDEN_POOL = []


def _widen(dst64: np.ndarray, src8: np.ndarray):
    """int8 rows -> the int64 rows ``quants`` holds (the reference's dtype): 35 MB written per flush of 2 048 proteins, 2 ms on
    one core -- four threads share it (numpy releases the lock inside the copy)."""
    n = len(src8)
    if n * src8.shape[1] < (1 << 21):
        np.copyto(dst64, src8)
        return
    if not _WIDEN_POOL:
        from concurrent.futures import ThreadPoolExecutor
        _WIDEN_POOL.append(ThreadPoolExecutor(max_workers=4, thread_name_prefix='dctfp-widen'))
    step = (n + 3) // 4
    list(_WIDEN_POOL[0].map(lambda a: np.copyto(dst64[a:a + step], src8[a:a + step]), range(0, n, step)))


def flush_records(fps: List[Fingerprint], threads: int = 1, qdim=QDIM, threshold: float = THRESHOLD):
    """What the writer stores for a flush -- (pid, domains, int8 rows) per protein, the files' content -- without filling the
    ``Fingerprint`` objects (``make_db`` never looks at them again).  Same values as ``_records(fingerprint_batch(fps))``."""
    if not fps:
        return []
    fl = _Flush(fps, threads, qdim, threshold)
    if fl.start():
        recs = fl.finish(objects=False)
        if recs is not None:
            LAST_PATH[0] = 'flush'
            return recs
    LAST_PATH[0] = 'generic'
    return _records(_fingerprint_batch_generic(fps, threads, qdim, threshold))


def _fingerprint_batch_generic(fps: List[Fingerprint], threads: int = 1, qdim=QDIM, threshold: float = THRESHOLD):
    """The flush for every input the reference takes (numpy or CPU tensors, domains already present, a second ``quantize``
    on the same objects, strings only Python's parser can judge): what ``fingerprint_batch`` was through round 5's first half.

    Nothing in here loops over domains in Python (the flush used to cost 117 us per protein against < 1 us
    of kernels): the contact selection is one GPU call, RecCut one threaded C call that runs while this thread prepares
    the embedding tables, the piece table is built by ``dctfp_build_pieces``, and every ``fp.quants[key]`` is a row VIEW of
    one int64 copy of the flush's result (the reference's dtype; keep ``fp`` alive and the flush's array stays alive)."""
    from . import reccut
    from .batch import LayerBatch, PieceTable, quantize_batch
    from . import _lib
    from .fingerprint import _to_device_matrix, warn_constant_channel
    if not fps:
        return fps
    _mark('start')
    lens = [len(fp.seq) for fp in fps]
    maps = [reccut._contact_tensor(fp.contacts, n) for fp, n in zip(fps, lens)]
    ctx = _lib.get_context(maps[0].device.index)
    ctx.get_option('degenerate_seen')                          # (the flag is the context's: drop what earlier callers left unread)
    # Contact selection and the domain cutter's recursion both run on the GPU (round 5: dctfp_contact_topk + dctfp_reccut; the
    # host library cost 9 us per protein on 16 threads, and the selected contacts had to come over for it); while the kernels
    # run, this thread builds the embedding tables.  What comes back is a few ints per domain.
    built = {}
    _mark('maps')

    def build_tables():
        _mark('enqueued top-k + cutter')
        keys0 = list(fps[0].embed.keys())
        mats = []
        for k in keys0:
            vals = [fp.embed[k] for fp in fps]
            # straight off the language model (float tensors on the GPU, rows contiguous): nothing to convert -- judged in a few
            # C-level passes over the list instead of a Python call per tensor; LayerBatch checks shapes and strides itself
            if not (all(map(torch.is_tensor, vals)) and all(map(_IS_CUDA, vals)) and set(map(_DTYPE, vals)) <= _KEPT_DTYPES):
                vals = [_to_device_matrix(v, keep_half=True) for v in vals]
            mats.append(vals)
        layers = []
        for i in range(len(keys0)):
            try:
                layers.append(LayerBatch(mats[i], qdim[2 * i], qdim[2 * i + 1]))
            except ValueError:       # (row-strided or transposed views, mixed strides: made contiguous one by one, then judged again)
                mats[i] = [_to_device_matrix(v, keep_half=True).contiguous() for v in mats[i]]
                layers.append(LayerBatch(mats[i], qdim[2 * i], qdim[2 * i + 1]))
        built['keys0'], built['mats'], built['layers'] = keys0, mats, layers
        _mark('embedding tables')

    cut = {'doms': reccut.domains_from_maps(maps, threshold, threads=max(1, threads), before_wait=build_tables)}
    _mark('cutter waited for + strings')
    if 'layers' not in built:
        build_tables()
    keys0, mats, layers = built['keys0'], built['mats'], built['layers']
    for fp, d, n in zip(fps, cut['doms'], lens):
        if len(d) > 1:
            d.append(f'1-{n}')                                 # src/fingerprint.py:106-107
        fp.domains.extend(d)
    rows = [m.shape[0] for m in mats[0]]
    _mark('domains extended')
    table = PieceTable(rows, [fp.domains for fp in fps])
    _mark('piece table')
    out = quantize_batch(layers, table)
    _mark('quantize enqueued')
    host8 = _to_host(out) if table.n_domains else np.zeros((0, 0), np.int8)
    _mark('results on the host')
    if table.n_domains and ctx.get_option('degenerate_seen'):
        # rare: the flush saw an exactly constant channel (0/0 -> all-zero block, the documented deviation).  Name the proteins.
        warn_constant_channel(_constant_channel_pids(fps, mats, table) or [fp.pid for fp in fps])
    host64 = host8.astype(np.int64)                            # np.array(list of ints) in the reference: int64
    keys = table.keys
    bounds = np.searchsorted(table.owner, np.arange(len(fps) + 1)).tolist()   # output rows of protein s: [bounds[s], bounds[s+1])
    for s, fp in enumerate(fps):
        a, b = bounds[s], bounds[s + 1]
        ks = keys[a:b]
        if not fp.quants:
            q = dict(zip(ks, host64[a:b]))                     # a row already is layer 0's block, layer 1's block, ... (:184-196)
            if len(q) == b - a:                                # (no key twice: RecCut's domains are disjoint)
                fp.quants = q
                fp.domains = ks
                fp._rows8 = host8[a:b]                         # what the writer stores (see _records)
                continue
        _extend_quants(fp, ks, host64[a:b], qdim, len(keys0))
    _mark('unpacked into the objects')
    if logging.getLogger().isEnabledFor(logging.INFO):         # one line per protein, as the reference writes them
        now = datetime.datetime.now()
        logging.info('\n'.join(f'{now} Fingerprinted {fp.pid}' for fp in fps))
    return fps


def _extend_quants(fp, ks, rows, qdim, n_layers):
    """The general case of the reference's bookkeeping (src/fingerprint.py:184-196) for one protein: ``quants`` already has
    entries, or two domain strings were cleaned to the same key -- every layer's block is appended to whatever the key
    holds, layer-major, then domain order."""
    held = {k: list(np.asarray(v).tolist()) for k, v in fp.quants.items()}
    off = 0
    for i in range(n_layers):
        nm = qdim[2 * i] * qdim[2 * i + 1]
        for k, row in zip(ks, rows):
            held.setdefault(k, []).extend(row[off:off + nm].tolist())
        off += nm
    fp.quants = {k: np.array(v) for k, v in held.items()}
    fp.domains = list(fp.quants.keys())
    fp.__dict__.pop('_rows8', None)


_PINNED = threading.local()


def _to_host(t: torch.Tensor) -> np.ndarray:
    """Device tensor -> numpy through a page-locked staging buffer (a pageable ``.cpu()`` of a flush's results runs at a
    fraction of the PCIe rate).  Returns a copy: the staging buffer is reused by the next call of this thread."""
    n = t.numel()
    pin = getattr(_PINNED, 'buf', None)
    if pin is None or pin.dtype != t.dtype or pin.numel() < n:
        pin = torch.empty(max(n + n // 4, 1 << 20), dtype=t.dtype, pin_memory=True)
        _PINNED.buf = pin
    view = pin[:n].view(t.shape)
    view.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return view.numpy().copy()


def _constant_channel_pids(fps, mats, table):
    """Proteins of a flush with a domain in which some channel of some layer is exactly constant (only run after the
    kernels flagged one; a handful of torch reductions per domain)."""
    hit = []
    pieces = table.pieces
    first = 0
    for d in range(table.n_domains):
        last = first
        while last < len(pieces) and pieces['domain'][last] == d:
            last += 1
        s = table.owner[d]
        if not hit or hit[-1] != fps[s].pid:
            for layer in mats:
                rows = torch.cat([layer[s][int(p['row_start']):int(p['row_start']) + int(p['n_rows'])] for p in pieces[first:last]])
                if bool((rows.max(dim=0).values == rows.min(dim=0).values).any()):
                    hit.append(fps[s].pid)
                    break
        first = last
    return hit


def load_model(name: str, device):
    from .embedding import Model, SyntheticModel
    if name == 'synthetic':
        model = SyntheticModel()
    else:
        model = Model()           # raises ImportError when fair-esm is not installed
    model.to_device(device)
    return model


def _records(fps: List[Fingerprint]):
    """Picklable (pid, domains, int8 matrix) triples for the writer.  A protein that came out of ``fingerprint_batch``'s
    fast path hands over the int8 rows of the flush as they left the GPU (no per-domain conversion)."""
    out = []
    for fp in fps:
        rows = fp.__dict__.get('_rows8')
        if rows is None or len(rows) != len(fp.domains):
            rows = np.array([fp.quants[d] for d in fp.domains], dtype=np.int8)
        out.append((fp.pid, list(fp.domains), rows))
    return out


class _Rec:
    def __init__(self, pid, domains, mat):
        self.pid, self.domains = pid, domains
        self.quants = {d: mat[i] for i, d in enumerate(domains)}


def process_sequences(seqs, model, device, maxlen: int, cpu: int, flush: int, sink):
    """Embeds and fingerprints ``seqs`` [(pid, sequence)] on ``device``; ``sink(records)`` receives the
    results of every flush."""
    from .embedding import Batch
    queue: List[Fingerprint] = []
    batch, cur = [], 0

    def run_batch(b):
        with stage('embed (language model + window stitching)'):
            bt = Batch(b, model, device)
            bt.embed_batch(LAYERS, maxlen)
            for emb in bt.embeds:
                queue.append(Fingerprint(pid=emb.pid, seq=emb.seq, embed=emb.embed, contacts=emb.contacts))

    # A flush in two halves (``_Flush``): its contact selection and domain cutter are enqueued when the queue is full, its second
    # half -- strings, piece table, dctfp_quantize, records -- runs when the NEXT queue is full (or the input ends): the cutter is
    # a latency (a few long proteins, one workgroup each, 3-5 ms per 2 048 proteins), and it passes while the next proteins are
    # embedded.  Records reach the writer in sequence order all the same.
    pending = []

    def finish_pending():
        while pending:
            fl, q = pending.pop(0)
            with stage('fingerprint (contact top-k + RecCut + dctfp_quantize)'):
                recs = fl.finish(objects=False) if fl is not None else None
                if recs is None:
                    recs = _records(_fingerprint_batch_generic(q, threads=cpu))
            with stage('hand over to the writer'):
                sink(recs)

    def flush_queue():
        q = list(queue)
        queue.clear()
        # (the flush before this one first: its fingerprint kernel goes onto the side stream, and behind this flush's cutter it would
        #  wait for exactly the latency the two halves are apart to avoid)
        finish_pending()
        with stage('fingerprint (contact top-k + RecCut + dctfp_quantize)'):
            fl = _Flush(q, threads=cpu)
            if not fl.start():
                fl = None
        pending.append((fl, q))

    for pid, seq in seqs:                          # same packing rule as Database.yield_seqs
        if batch and (cur + len(seq) > maxlen or len(batch) > cpu):
            run_batch(batch)
            batch, cur = [], 0
            if len(queue) >= flush:
                flush_queue()
        batch.append((pid, seq))
        cur += len(seq)
    if batch:
        run_batch(batch)
    if queue:
        flush_queue()
    finish_pending()


def _gpu_worker(rank: int, n_gpu: int, shards, model_name: str, maxlen: int, cpu: int, flush: int, out_q):
    """One worker process per GPU (reference ``queue_gpu``, src/make_db.py:54-92).  Whatever happens in here, the
    parent hears about it: results as ``('recs', rank, records)``, a Python error as ``('error', rank, traceback)`` and
    always a closing ``('done', rank)``; a worker that dies without it (signal, abort) is noticed by ``_run_workers``."""
    try:
        n_dev = torch.cuda.device_count()
        dev = torch.device('cuda', rank % max(1, n_dev))
        torch.cuda.set_device(dev)
        if n_dev > 1:           # this worker's host threads next to its GPU (one GPU: nothing to choose)
            from .dist import pin_to_gpu_numa
            where = pin_to_gpu_numa(dev.index)
            logging.info(f"GPU worker {rank}: {torch.cuda.get_device_name(dev)} at {where['pci']}, NUMA node {where['numa_node']}, "
                         f"{where['pinned'] or 'no'} CPUs pinned")
        model = load_model(model_name, dev)
        process_sequences(shards[rank], model, dev, maxlen, cpu, flush, lambda recs: out_q.put(('recs', rank, recs)))
        out_q.put(('stats', rank, dict(STAGE_SECONDS)))
    except BaseException:       # noqa: BLE001 -- reported to the parent, which stops the build
        import traceback
        out_q.put(('error', rank, traceback.format_exc()))
    finally:
        out_q.put(('done', rank))


def _run_workers(n_gpu: int, worker_args: tuple, sink, target=None, poll_s: float = 2.0):
    """Starts one process per GPU, hands every batch of records to ``sink`` as it arrives and returns when all
    workers have said 'done'.  A worker that reports an error, or that is gone without its closing message, ends the
    build: the other workers are terminated and RuntimeError is raised (what was written so far stays in the
    database, ``fpcount`` marks it, the next run resumes)."""
    import queue as _queue
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out_q = ctx.Queue()
    procs = [ctx.Process(target=target or _gpu_worker, args=(r, n_gpu) + tuple(worker_args) + (out_q,)) for r in range(n_gpu)]
    for p in procs:
        p.start()
    finished, failure = set(), None
    try:
        while len(finished) < n_gpu and failure is None:
            try:
                item = out_q.get(timeout=poll_s)
            except _queue.Empty:
                for r, p in enumerate(procs):
                    if r not in finished and not p.is_alive():
                        # dead without a closing message; drain what it may still have queued before giving up
                        try:
                            while True:
                                late = out_q.get(timeout=0.5)
                                if late[0] == 'recs':
                                    sink(late[2])
                                elif late[0] == 'stats':
                                    pass
                                elif late[0] == 'done':
                                    finished.add(late[1])
                                elif late[0] == 'error':
                                    failure = f'GPU worker {late[1]} failed:\n{late[2]}'
                        except _queue.Empty:
                            pass
                        if r not in finished and failure is None:
                            failure = f'GPU worker {r} died (exit code {p.exitcode}) without finishing its shard'
                continue
            if item[0] == 'recs':
                sink(item[2])
            elif item[0] == 'stats':
                for k, v in item[2].items():
                    STAGE_SECONDS[f'worker: {k}'] = max(STAGE_SECONDS[f'worker: {k}'], v)
            elif item[0] == 'error':
                failure = f'GPU worker {item[1]} failed:\n{item[2]}'
            elif item[0] == 'done':
                finished.add(item[1])
    finally:
        if failure is not None or len(finished) < n_gpu:
            for p in procs:
                if p.is_alive():
                    p.terminate()
        # a worker that has said 'done' may still be tearing down its model / HIP context: wait for it (exitcode None =
        # still running, not a failure); one that is stuck in teardown after delivering everything is ended, not raised
        for p in procs:
            p.join(timeout=30 if failure is not None else 600)
    if failure is not None:
        raise RuntimeError(failure)
    for r, p in enumerate(procs):
        if p.exitcode is None:
            logging.warning(f'GPU worker {r} delivered its shard but did not exit within 600 s; terminating it')
            p.terminate()
            p.join(timeout=30)
        elif p.exitcode != 0:
            raise RuntimeError(f'GPU worker {r} exited with code {p.exitcode}')


class OrderedWriter:
    """The single writer of a build: records arrive per flush -- in pending order from one GPU, interleaved from
    several -- and go into the database in pending order, one transaction per contiguous run that is complete.
    Only what arrived ahead of its turn is held back (O(workers x flush) records), so the ``.db`` is the checkpoint
    *during* a run as it is in the reference (commit per protein + ``fpcount``, src/database.py:211-224, :150):
    an interrupted build resumes after the last committed protein and ends with the same files."""

    def __init__(self, db: Database, pending):
        self.db = db
        self.order = {pid: i for i, (pid, _) in enumerate(pending)}
        if len(self.order) != len(pending):       # (pid is the PRIMARY KEY of `sequences`: cannot happen from a database)
            raise ValueError('duplicate protein ids in the pending list: the ordered writer would stall at the first one')
        self.next = 0                 # pending index of the next protein to write
        self.held = {}                # pending index -> record that arrived early
        self.written = 0

    def add(self, records):
        with stage('writer: SQLite transactions'):
            self._add(records)

    def _add(self, records):
        for pid, domains, mat in records:
            self.held[self.order[pid]] = _Rec(pid, domains, mat)
        run = []
        while self.next in self.held:
            run.append(self.held.pop(self.next))
            self.next += 1
        if run:
            self.db.add_fprints(run)          # one transaction
            self.written += len(run)

    def finish(self):
        """Whatever is still held (gaps are proteins that produced no record) goes in, in pending order."""
        if self.held:
            run = [self.held[i] for i in sorted(self.held)]
            self.held.clear()
            self.db.add_fprints(run)
            self.written += len(run)


def run(args: argparse.Namespace) -> Database:
    if args.out:
        logging.basicConfig(level=logging.INFO, filename=args.out, filemode='w', format='%(message)s', force=True)
    STAGE_SECONDS.clear()
    with stage('open database + read FASTA'):
        db = Database(args.dbfile, args.fafile)
    print('Fingerprinting sequences...\n')
    pending = db.pending()
    n_gpu = int(args.gpu) if args.gpu else 1
    cpu = max(1, int(args.cpu))
    writer = OrderedWriter(db, pending)

    if not torch.cuda.is_available():
        raise RuntimeError('make_db needs an MI355X GPU: the fingerprint path has no CPU fallback')
    if n_gpu <= 1:
        dev = torch.device('cuda', torch.cuda.current_device())
        model = load_model(args.model, dev)
        process_sequences(pending, model, dev, args.maxlen, cpu, args.flush, writer.add)
    else:
        from .dist import balanced_shards
        idx = balanced_shards([len(s) for _, s in pending], n_gpu)
        shards = [[pending[i] for i in ix] for ix in idx]
        _run_workers(n_gpu, (shards, args.model, args.maxlen, cpu, args.flush), writer.add)
    # table order = pending order (ascending length) whatever the number of GPUs
    with stage('writer: SQLite transactions'):
        writer.finish()
    with stage('rename_vid + metadata'):
        db.rename_vid()
        db.update_metadata()
    if not args.noindex:
        os.environ['OMP_NUM_THREADS'] = str(args.cpu)
        print('Creating index...')
        with stage('.index'):
            db.create_index()
    if not args.nonpz:
        with stage('-dct.npz'):
            db.save_fprints(f'{args.dbfile}-dct.npz')
    if not args.nodom:
        with stage('.dom'):
            db.save_doms(f'{args.dbfile}.dom')
    for name, sec in STAGE_SECONDS.items():
        logging.info(f'stage {name}: {sec:.1f} s')
    return db


def build_parser() -> argparse.ArgumentParser:
    """The reference's flags (src/make_db.py:167-177: names, types, defaults) + ``--model`` / ``--flush``."""
    ap = argparse.ArgumentParser(description='FASTA -> fingerprint database (.db, -dct.npz, .dom, .index) on MI355X')
    ap.add_argument('--dbfile', required=True, help='database to create or resume (name without extension)')
    ap.add_argument('--fafile', help='proteins to add (.fa / .fasta); omit to resume an existing database')
    ap.add_argument('--out', default='', help='progress log file (default: console)')
    ap.add_argument('--maxlen', type=int, default=500, help='longest window given to the language model')
    ap.add_argument('--cpu', type=int, default=1, help='host threads for RecCut')
    ap.add_argument('--gpu', type=int, help='GPU worker processes (one per device)')
    for flag, what in (('--noindex', 'the FAISS .index'), ('--nonpz', 'the -dct.npz'), ('--nodom', 'the .dom')):
        ap.add_argument(flag, action='store_true', help=f'do not write {what}')
    ap.add_argument('--model', choices=['esm', 'synthetic'], default='esm',
                    help='language model: fair-esm ESM-2 (as the reference) or the synthetic stand-in')
    ap.add_argument('--flush', type=int, default=1024, help='proteins per batched fingerprint call')
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    db = run(args)
    db.close()


if __name__ == '__main__':
    main()
