"""``Embedding`` / ``Batch`` / ``Model`` -- the producer side of the fingerprint path, mirroring
mgtools/DCTdomain ``src/embedding.py`` (:17-278) with the chunk + overlap-average stitching on
the GPU (``dctfp_stitch``) and the results left on the device for ``Fingerprint``.

The language model is NOT part of this build (fair-esm is third-party and not
installed, no weights).  ``Model`` loads the real ESM-2 when ``esm`` is importable; any object
with the same two callables works (``SyntheticModel`` for tests and benchmarks):

    model.esm_tokenizer([(pid, seq), ...]) -> (_, _, tokens[B, T+2])
    model.esm_encoder(tokens, repr_layers=[15, 21], return_contacts=True)
        -> {'representations': {layer: [B, T+2, D]}, 'contacts': [B, T, T]}
"""

from __future__ import annotations

import ctypes as C
from collections.abc import Sequence
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import _lib

OVERLAP = 200     # hard-coded in the reference, src/embedding.py:163


class StitchJob(C.Structure):
    """``dctfp_stitch_job`` (include/dctfp.h)."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('ld_src', C.c_int64), ('ld_dst', C.c_int64),
                ('n_rows', C.c_int32), ('n_avg', C.c_int32), ('level', C.c_int32), ('reserved', C.c_int32)]


class Model:
    """ESM-2 loader of the reference (src/embedding.py:17-56): esm2_t30_150M_UR50D, layers 15/21.
    Needs the third-party ``esm`` package and its weights."""

    def __init__(self, name: str = 'esm2_t30_150M_UR50D'):
        try:
            import esm  # noqa: F401
        except ImportError as exc:
            raise ImportError('fair-esm is not installed: pass a model object with esm_tokenizer / esm_encoder '
                              '(e.g. dctdomain_amd.embedding.SyntheticModel) instead') from exc
        import esm
        self.esm_encoder, self.alphabet = getattr(esm.pretrained, name)()
        self.esm_tokenizer = self.alphabet.get_batch_converter()
        self.esm_encoder.eval()

    def to_device(self, device):
        self.esm_encoder.to(device)


class SyntheticModel:
    """Deterministic stand-in for the language model (tests, benchmarks, dry runs): embeddings and
    contact maps are cheap functions of the tokens AND of the position inside the window, so that
    overlapping windows disagree on shared residues the way a real model does."""

    AA = 'ACDEFGHIKLMNPQRSTVWYXBZUO'

    def __init__(self, dim: int = 640, layers=(15, 21), seed: int = 0, device=None):
        self.dim = dim
        self.layers = tuple(layers)
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        g = torch.Generator().manual_seed(seed)
        self.table = {l: torch.randn((len(self.AA) + 2, dim), generator=g) * (0.5 + 0.1 * k) for k, l in enumerate(self.layers)}
        self.pos = torch.randn((4096, dim), generator=g) * 0.3
        self.chan = 5.0 * torch.randn((1, dim), generator=g)
        self.padding_idx = len(self.AA) + 1

    def to_device(self, device):
        self.device = torch.device(device)
        self.table = {l: t.to(self.device) for l, t in self.table.items()}
        self.pos = self.pos.to(self.device)
        self.chan = self.chan.to(self.device)

    def esm_tokenizer(self, pairs):
        width = max(len(s) for _, s in pairs) + 2
        tok = torch.full((len(pairs), width), self.padding_idx, dtype=torch.long)
        for b, (_, s) in enumerate(pairs):
            ids = [self.AA.find(ch) if ch in self.AA else 20 for ch in s]
            tok[b, 0] = len(self.AA)
            tok[b, 1:1 + len(s)] = torch.tensor(ids, dtype=torch.long)
            tok[b, 1 + len(s)] = len(self.AA)
        return None, None, tok

    @torch.no_grad()
    def esm_encoder(self, tokens, repr_layers=None, return_contacts=True):
        tokens = tokens.to(self.device)
        b, t = tokens.shape
        reps = {}
        valid = (tokens != self.padding_idx).unsqueeze(-1).float()      # padding must not leak into a sequence
        for l in self.layers:
            x = self.table[l][tokens] + self.pos[torch.arange(t, device=self.device) % self.pos.shape[0]][None]
            mean = (x * valid).sum(dim=1, keepdim=True) / valid.sum(dim=1, keepdim=True)
            x = x + 0.05 * mean + self.chan
            reps[l] = x.float()
        n = t - 2
        i = torch.arange(n, device=self.device)
        near = 0.9 * torch.exp(-(i[:, None] - i[None, :]).abs().float() / 12.0)
        h = (tokens[:, 1:-1, None] * 31 + tokens[:, None, 1:-1] * 17 + i[None, :, None] + i[None, None, :]) % 97
        ct = near[None] + 0.25 * (h.float() / 97.0) * ((i[:, None] // 120) == (i[None, :] // 120))[None]
        ct = 0.5 * (ct + ct.transpose(1, 2))
        return {'representations': reps, 'contacts': ct.clamp(0, 1).float()}


def _stitch(jobs: List[StitchJob], n_cols: int, square: bool, device):
    arr = (StitchJob * len(jobs))(*jobs)
    ctx = _lib.get_context(device.index)
    stream = torch.cuda.current_stream(device)
    _lib.check(ctx._lib.dctfp_stitch(ctx.handle, arr, len(jobs), int(n_cols), 1 if square else 0,
                                     C.c_void_p(stream.cuda_stream)))


def stitch_embeddings(windows: List[torch.Tensor], overlap: int = OVERLAP) -> torch.Tensor:
    """``edata[lay][-olp:] = (edata[lay][-olp:] + emb[:olp]) / 2; cat(emb[olp:])`` over the windows
    of one sequence and one layer (src/embedding.py:185-187), on the GPU."""
    return stitch_embeddings_batch([windows], overlap)[0]


class _Views(Sequence):
    """The per-sequence results of a batch as a read-only sequence of tensors: views of ONE allocation, made when asked
    for (2 000 ``narrow`` calls up front cost more than the kernels of the batch)."""

    def __init__(self, big: torch.Tensor, first, sizes, square: bool):
        self._big, self._first, self._sizes, self._square = big, first, sizes, square

    def __len__(self):
        return len(self._sizes)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        n = len(self._sizes)
        if k < 0:
            k += n
        if not 0 <= k < n:
            raise IndexError(k)
        a, size = int(self._first[k]), int(self._sizes[k])
        if self._square:
            return self._big[a:a + size * size].view(size, size)
        return self._big[a:a + size]


def _float_windows(seq_windows):
    """Flat list of the windows as float32 tensors with contiguous rows + windows per sequence.  The common case (windows
    straight off the model) costs two C-level passes over the list."""
    flat = [w for windows in seq_windows for w in windows]
    if {w.dtype for w in flat} != {torch.float32} or not all(map(torch.Tensor.is_contiguous, flat)):
        fixed = []
        for w in flat:
            if w.dtype != torch.float32:
                w = w.float()
            if w.dim() == 2 and w.stride(-1) != 1:
                w = w.contiguous()
            fixed.append(w)
        flat = fixed
    return flat, np.fromiter(map(len, seq_windows), dtype=np.int64, count=len(seq_windows))


def _stitch_sequences(flat, counts, step: int, square: bool, n_cols: int):
    """Geometry (``dctfp_stitch_sizes``), ONE allocation for all results, launches (``dctfp_stitch_sequences``).  Returns
    the per-sequence results as views of that allocation."""
    device = flat[0].device
    if device.type != 'cuda':
        raise ValueError('windows must be GPU tensors (no CPU fallback)')
    n_seq, n_win = len(counts), len(flat)
    seq_win = np.zeros(n_seq + 1, dtype=np.int64)
    np.cumsum(counts, out=seq_win[1:])
    rows = np.fromiter((w.shape[0] for w in flat), dtype=np.int32, count=n_win)
    numel = np.fromiter(map(torch.Tensor.numel, flat), dtype=np.int64, count=n_win)
    ptrs = np.fromiter(map(torch.Tensor.data_ptr, flat), dtype=np.uint64, count=n_win)
    width = rows.astype(np.int64) if square else np.int64(n_cols)
    if not (numel == rows * width).all() or not all(w.dim() == 2 for w in flat[:1]):
        raise ValueError('a contact window must be square' if square else 'windows of different widths')
    if all(map(torch.Tensor.is_contiguous, flat)):
        lds = np.broadcast_to(width, rows.shape).astype(np.int64)
    else:                   # (row-strided views: the leading dimension of each window)
        lds = np.fromiter((w.stride(0) if w.shape[0] > 1 else w.shape[1] for w in flat), dtype=np.int64, count=n_win)
    lib = _lib.load()
    sizes = np.empty(n_seq, dtype=np.int64)
    rc = lib.dctfp_stitch_sizes(rows.ctypes.data, seq_win.ctypes.data, n_seq, int(step), 1 if square else 0, sizes.ctypes.data)
    if rc == _lib.DCTFP_ERR_SHAPE:
        raise ValueError(lib.dctfp_last_error().decode())          # torch would fail to broadcast in the reference
    _lib.check(rc, lib)
    if square:
        elems = sizes * sizes
        big = torch.zeros(int(elems.sum()), dtype=torch.float32, device=device)      # new_mat = torch.zeros, :143
        dst_ld = sizes
        first = np.zeros(n_seq, dtype=np.int64)
        np.cumsum(elems[:-1], out=first[1:])
        dst = np.uint64(big.data_ptr()) + first.astype(np.uint64) * np.uint64(4)
    else:
        big = torch.empty((int(sizes.sum()), n_cols), dtype=torch.float32, device=device)
        dst_ld = np.full(n_seq, n_cols, dtype=np.int64)
        first = np.zeros(n_seq, dtype=np.int64)
        np.cumsum(sizes[:-1], out=first[1:])
        dst = np.uint64(big.data_ptr()) + first.astype(np.uint64) * np.uint64(4 * n_cols)
    ctx = _lib.get_context(device.index)
    stream = torch.cuda.current_stream(device)
    _lib.check(lib.dctfp_stitch_sequences(ctx.handle, ptrs.ctypes.data, rows.ctypes.data, lds.ctypes.data, seq_win.ctypes.data,
                                          n_seq, dst.ctypes.data, np.ascontiguousarray(dst_ld, dtype=np.int64).ctypes.data,
                                          int(n_cols), int(step), 1 if square else 0, C.c_void_p(stream.cuda_stream)), lib)
    return _Views(big, first, sizes, square)


def stitch_windows_flat(layer, win_rows, seq_win, sizes, overlap: int = OVERLAP):
    """The stitched matrices of ONE layer given as a ``batch.LayerBatch`` of float32 windows (device pointers + common row
    stride): (one (sum sizes, D) tensor, first row of every sequence).  What ``batch.quantize_windows`` falls back to when a call
    is not one the window-averaging kernel takes."""
    if layer.dtype != _lib.DCTFP_F32:
        raise ValueError('windows are stitched in float32, as the reference does (src/embedding.py:185-187)')
    n_seq, n_win = len(sizes), len(win_rows)
    big = torch.empty((int(sizes.sum()), layer.n_cols), dtype=torch.float32, device=layer.device)
    first = np.zeros(n_seq, dtype=np.int64)
    np.cumsum(sizes[:-1], out=first[1:])
    dst = np.uint64(big.data_ptr()) + first.astype(np.uint64) * np.uint64(4 * layer.n_cols)
    lds = np.full(n_win, layer.ld, dtype=np.int64)
    dst_ld = np.full(n_seq, layer.n_cols, dtype=np.int64)
    lib = _lib.load()
    ctx = _lib.get_context(layer.device.index)
    stream = torch.cuda.current_stream(layer.device)
    _lib.check(lib.dctfp_stitch_sequences(ctx.handle, layer.ptrs.ctypes.data, np.ascontiguousarray(win_rows, dtype=np.int32).ctypes.data,
                                          lds.ctypes.data, np.ascontiguousarray(seq_win, dtype=np.int64).ctypes.data, n_seq,
                                          dst.ctypes.data, dst_ld.ctypes.data, int(layer.n_cols), int(overlap), 0,
                                          C.c_void_p(stream.cuda_stream)), lib)
    return big, first


def stitch_embeddings_batch(seq_windows: List[List[torch.Tensor]], overlap: int = OVERLAP):
    """Same for many sequences: one kernel launch per window index for the whole batch.  The window geometry is worked
    out in C (``dctfp_stitch_sequences``); the results come back as a sequence of views of ONE allocation (keep one and
    all stay alive)."""
    if not seq_windows:
        return []
    flat, counts = _float_windows(seq_windows)
    if (counts < 1).any():
        raise ValueError('a sequence without windows')
    if flat[0].dim() != 2:
        raise ValueError(f'a window must be a 2-D tensor, got shape {tuple(flat[0].shape)}')
    return _stitch_sequences(flat, counts, overlap, False, flat[0].shape[1])


def stitch_contacts_batch(seq_windows: List[List[torch.Tensor]], inc: int):
    """``combine_contacts`` applied window after window (src/embedding.py:123-150, :188): window i's
    map lands at offset ``inc * i``; the part overlapping the running map is averaged."""
    if not seq_windows:
        return []
    flat, counts = _float_windows(seq_windows)
    if (counts < 1).any():
        raise ValueError('a sequence without windows')
    return _stitch_sequences(flat, counts, inc, True, 1)


@dataclass
class Embedding:
    """One protein to embed (src/embedding.py:58-192).  ``embed`` / ``contacts`` end up as float32
    torch tensors on the GPU (the reference copies them to numpy; ``Fingerprint`` takes either)."""
    pid: str = field(default_factory=str)
    seq: str = field(default_factory=str)
    embed: dict = field(default_factory=dict)
    contacts: object = field(default_factory=list)

    def __post_init__(self):
        self.contacts = np.array([])

    def split_seq(self, maxlen: int, overlap: int) -> list:
        """Windows of ``maxlen`` every ``maxlen - overlap`` residues; a window not longer than the
        overlap is dropped (src/embedding.py:83-100)."""
        subseqs = []
        for i in range(0, len(self.seq), maxlen - overlap):
            subseq = self.seq[i:i + maxlen]
            if len(subseq) > overlap:
                subseqs.append(subseq)
        return subseqs

    def extract_esm2(self, seq: str, model, device) -> dict:
        """Model forward of one window (src/embedding.py:103-120)."""
        _, _, batch_tokens = model.esm_tokenizer([(self.pid, seq)])
        batch_tokens = batch_tokens.to(device)
        with torch.no_grad():
            return model.esm_encoder(batch_tokens, repr_layers=[15, 21], return_contacts=True)

    def combine_contacts(self, mat1: torch.Tensor, mat2: torch.Tensor, inc: int, times: int) -> torch.Tensor:
        """Running (n x n) map + window (m x m) map at offset ``inc * times`` (src/embedding.py:123-150)."""
        olp = inc * times
        mlen1, mlen2 = mat1.size(0), mat2.size(0)
        out = torch.zeros((olp + mlen2, olp + mlen2), dtype=torch.float32, device=mat1.device)
        jobs = [StitchJob(mat1.data_ptr(), out.data_ptr(), mat1.stride(0), out.stride(0), mlen1, 0, 0, 0),
                StitchJob(mat2.data_ptr(), out.data_ptr() + (olp * out.stride(0) + olp) * 4, mat2.stride(0),
                          out.stride(0), mlen2, min(max(mlen1 - olp, 0), mlen2), 1, 0)]
        _stitch(jobs, 1, True, mat1.device)
        return out

    def embed_seq(self, model, device, layers: list, maxlen: int):
        """Embeds the sequence window by window and stitches the windows (src/embedding.py:153-192)."""
        olp = OVERLAP
        subseqs = self.split_seq(maxlen, olp) if len(self.seq) > maxlen else [self.seq]
        wins = {layer: [] for layer in layers}
        cts = []
        for seq in subseqs:
            res = self.extract_esm2(seq, model, device)
            for layer in layers:
                wins[layer].append(res['representations'][layer][0][1:-1])
            cts.append(res['contacts'][0])
        if len(subseqs) == 1:
            self.embed = {layer: wins[layer][0] for layer in layers}
            self.contacts = cts[0]
            return
        stitched = stitch_embeddings_batch([wins[layer] for layer in layers], olp)
        self.embed = {layer: stitched[k] for k, layer in enumerate(layers)}
        self.contacts = stitch_contacts_batch([cts], maxlen - olp)[0]


@dataclass
class Batch:
    """A batch of (pid, sequence) tuples to embed (src/embedding.py:195-278)."""
    seqs: list = field(default_factory=list)
    model: object = None
    device: object = field(default_factory=str)
    embeds: list = field(default_factory=list)

    def embed_batch(self, layers: list, maxlen: int):
        if len(self.seqs) == 1:
            self.embed_single(layers, maxlen)
        else:
            self.embed_parallel(layers)

    def embed_single(self, layers: list, maxlen: int):
        for seq in self.seqs:
            emb = Embedding(pid=seq[0], seq=seq[1])
            emb.embed_seq(self.model, self.device, layers, maxlen)
            self.embeds.append(emb)

    def embed_parallel(self, layers: list):
        """Several short sequences in one forward, no chunking (src/embedding.py:241-261)."""
        _, _, batch_tokens = self.model.esm_tokenizer(self.seqs)
        pad = getattr(getattr(self.model, 'alphabet', None), 'padding_idx', getattr(self.model, 'padding_idx', None))
        batch_lens = (batch_tokens != pad).sum(1)
        batch_tokens = batch_tokens.to(self.device)
        with torch.no_grad():
            res = self.model.esm_encoder(batch_tokens, repr_layers=[15, 21], return_contacts=True)
        for i, seq in enumerate(self.seqs):
            n = int(batch_lens[i])
            emb = Embedding(pid=seq[0], seq=seq[1])
            emb.contacts = res['contacts'][i][:n - 2, :n - 2]
            for layer in layers:
                emb.embed[layer] = res['representations'][layer][i][1:n - 1]
            self.embeds.append(emb)
