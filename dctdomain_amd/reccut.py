"""Domain prediction step of the reference's ``Fingerprint`` (mgtools/DCTdomain
src/fingerprint.py:45-107) without the Python pair list, the .ce temp file and the RecCut
subprocess:

* the top ``int(t * L)`` contacts with ``j >= i + 5`` are selected on the GPU
  (``dctfp_contact_topk``; the reference builds and sorts an O(L^2) Python list, :54-67);
* the domain boundaries come from ``libreccut.so`` in-process (include/reccut.h; the
  reference spawns ``src/RecCut`` on a text file, :92-100).

``write_ce`` still writes the byte-identical .ce text for users of that file format."""

from __future__ import annotations

import ctypes as C
import threading
from typing import List, Sequence

import numpy as np
import torch

from itertools import repeat

from . import _lib

CUT1_DEFAULT = 0.08      # src/RecCut.cpp:12
CUT2_DEFAULT = 0.07      # src/RecCut.cpp:13


def _contact_tensor(contacts, n_res: int, device=None) -> torch.Tensor:
    if isinstance(contacts, torch.Tensor):
        t = contacts
        if t.dtype == torch.float32 and t.is_cuda and t.dim() == 2 and t.shape[0] == n_res and t.shape[1] == n_res \
                and t.stride(1) == 1:
            return t                            # straight off the language model: nothing to do
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(contacts, dtype=np.float32)))
    t = t.reshape(n_res, n_res)                 # cta = self.contacts.reshape(slen, slen)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if t.device.type != 'cuda':
        if not torch.cuda.is_available():
            raise RuntimeError('dctdomain_amd needs an MI355X GPU; there is no CPU fallback')
        t = t.to(device if device is not None else torch.device('cuda', torch.cuda.current_device()))
    if t.stride(1) != 1:
        t = t.contiguous()
    return t


_PINNED = threading.local()


def _pinned(name: str, dtype, n: int) -> torch.Tensor:
    """A page-locked staging buffer of this thread (grown geometrically): device -> host copies of a flush's contacts run
    at the PCIe rate instead of the pageable one."""
    buf = getattr(_PINNED, name, None)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n + n // 2, 1 << 16), dtype=dtype, pin_memory=True)
        setattr(_PINNED, name, buf)
    return buf[:n]


def top_contacts_batch(maps: Sequence[torch.Tensor], t: float, sort: bool = True, own: bool = True):
    """Top ``int(t*L)`` contacts of each map.  Returns (offs, i, j, v) as numpy arrays: protein
    p's contacts are ``[offs[p], offs[p+1])``; with ``sort`` they are ordered by (-v, i, j) -- the
    order of the reference's CON line (the domain cutter itself does not care about the order).

    Selection (``dctfp_contact_topk``) and order (``dctfp_contact_sort``) both happen on the GPU; what comes back is one
    copy of the selected entries through page-locked buffers.  ``own=False`` returns views of those buffers instead of
    copies of them (64 MB for 4 096 proteins of 500 residues: 5 of the call's 12 ms): valid until this thread's next call,
    which is all a database flush needs (it hands them to the domain cutter and waits for it)."""
    n = len(maps)
    if n == 0:
        return np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)
    device = maps[0].device
    lib = _lib.load()
    n_res = np.fromiter(map(len, maps), dtype=np.int32, count=n)     # (C-level iteration: three generator passes over 4 096 maps cost 3 ms)
    # dctfp_contact_count for all proteins at once: min(int(t * L), pairs with j >= i + 5)
    L64 = n_res.astype(np.int64)
    cand = np.where(L64 >= 6, (L64 - 5) * (L64 - 4) // 2, 0)
    counts = np.minimum(np.maximum((float(t) * L64.astype(np.float64)).astype(np.int64), 0), cand)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=offs[1:])
    total = int(offs[-1])
    oi = torch.empty(max(total, 1), dtype=torch.int32, device=device)
    oj = torch.empty(max(total, 1), dtype=torch.int32, device=device)
    ov = torch.empty(max(total, 1), dtype=torch.float32, device=device)
    on = torch.zeros(n, dtype=torch.int32, device=device)
    ptrs = np.fromiter(map(torch.Tensor.data_ptr, maps), dtype=np.uint64, count=n)
    lds = np.fromiter(map(torch.Tensor.stride, maps, repeat(0)), dtype=np.int64, count=n)
    lds = np.where(n_res > 1, lds, np.maximum(L64, 1))              # (a one-row map may carry any stride)
    ctx = _lib.get_context(device.index)
    stream = torch.cuda.current_stream(device)
    sp = C.c_void_p(stream.cuda_stream)
    _lib.check(lib.dctfp_contact_topk(ctx.handle, ptrs.ctypes.data, lds.ctypes.data, n_res.ctypes.data, n, float(t),
                                      oi.data_ptr(), oj.data_ptr(), ov.data_ptr(), offs.ctypes.data, on.data_ptr(), sp))
    on_device = np.ones(n, dtype=np.uint8)
    if sort:
        _lib.check(lib.dctfp_contact_sort(ctx.handle, ptrs.ctypes.data, lds.ctypes.data, n_res.ctypes.data, n, float(t),
                                          oi.data_ptr(), oj.data_ptr(), ov.data_ptr(), offs.ctypes.data,
                                          on_device.ctypes.data, sp))
    pi, pj, pv, pn = (_pinned('i', torch.int32, total), _pinned('j', torch.int32, total), _pinned('v', torch.float32, total),
                      _pinned('n', torch.int32, n))
    pi.copy_(oi[:total], non_blocking=True)
    pj.copy_(oj[:total], non_blocking=True)
    pv.copy_(ov[:total], non_blocking=True)
    pn.copy_(on, non_blocking=True)
    stream.synchronize()
    hi, hj, hv = pi.numpy(), pj.numpy(), pv.numpy()
    if own:
        hi, hj, hv = hi.copy(), hj.copy(), hv.copy()
    if not (pn.numpy() == counts).all():
        raise RuntimeError('dctfp_contact_topk wrote a different number of contacts than dctfp_contact_count says')
    for p in (np.flatnonzero(on_device == 0) if sort else ()):       # longer than the device network holds (L > 6 301)
        a, b = offs[p], offs[p + 1]
        order = np.lexsort((hj[a:b], hi[a:b], -hv[a:b].astype(np.float64)))
        hi[a:b], hj[a:b], hv[a:b] = hi[a:b][order], hj[a:b][order], hv[a:b][order]
    return offs, hi, hj, hv


def _select_on_device(maps: Sequence[torch.Tensor], t: float):
    """``dctfp_contact_topk`` for a batch, everything left on the device: (n_res, counts, offs [host], oi, oj, ov, on [device])."""
    n = len(maps)
    device = maps[0].device
    lib = _lib.load()
    n_res = np.fromiter(map(len, maps), dtype=np.int32, count=n)
    L64 = n_res.astype(np.int64)
    cand = np.where(L64 >= 6, (L64 - 5) * (L64 - 4) // 2, 0)
    counts = np.minimum(np.maximum((float(t) * L64.astype(np.float64)).astype(np.int64), 0), cand)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=offs[1:])
    total = int(offs[-1])
    oi = torch.empty(max(total, 1), dtype=torch.int32, device=device)
    oj = torch.empty(max(total, 1), dtype=torch.int32, device=device)
    ov = torch.empty(max(total, 1), dtype=torch.float32, device=device)
    on = torch.zeros(n, dtype=torch.int32, device=device)
    ptrs = np.fromiter(map(torch.Tensor.data_ptr, maps), dtype=np.uint64, count=n)
    lds = np.fromiter(map(torch.Tensor.stride, maps, repeat(0)), dtype=np.int64, count=n)
    lds = np.where(n_res > 1, lds, np.maximum(L64, 1))
    ctx = _lib.get_context(device.index)
    sp = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.dctfp_contact_topk(ctx.handle, ptrs.ctypes.data, lds.ctypes.data, n_res.ctypes.data, n, float(t),
                                      oi.data_ptr(), oj.data_ptr(), ov.data_ptr(), offs.ctypes.data, on.data_ptr(), sp))
    return n_res, counts, offs, oi, oj, ov, on


#: proteins of the last ``domains_from_maps`` / ``domains_from_contacts_gpu`` call of this thread that the GPU cutter handed back
#: to the host library (status -1) -- tests and profiles read it
LAST = threading.local()


def _cut_on_device(n_res, counts, offs, oi, oj, ov, on, cut1, cut2, threads, before_wait):
    """``dctfp_reccut`` on contacts that already sit on the device (+ the host library for what it hands back)."""
    n = len(n_res)
    device = oi.device
    lib = _lib.load()
    uniq = np.unique(n_res)
    room = np.fromiter((lib.dctfp_reccut_room(int(v)) for v in uniq), dtype=np.int64, count=len(uniq))[np.searchsorted(uniq, n_res)]
    enc_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(room, out=enc_off[1:])
    enc = torch.empty(int(enc_off[-1]), dtype=torch.int32, device=device)
    ctx = _lib.get_context(device.index)
    stream = torch.cuda.current_stream(device)
    _lib.check(lib.dctfp_reccut(ctx.handle, n_res.ctypes.data, n, oi.data_ptr(), oj.data_ptr(), ov.data_ptr(), offs.ctypes.data,
                                float(cut1), float(cut2), enc.data_ptr(), enc_off.ctypes.data, C.c_void_p(stream.cuda_stream)), lib)
    penc = _pinned('enc', torch.int32, int(enc_off[-1]))
    penc.copy_(enc, non_blocking=True)
    pn = None
    if on is not None:
        pn = _pinned('n', torch.int32, n)
        pn.copy_(on, non_blocking=True)
    if before_wait is not None:
        before_wait()
    stream.synchronize()
    if pn is not None and not (pn.numpy() == counts).all():
        raise RuntimeError('dctfp_contact_topk wrote a different number of contacts than dctfp_contact_count says')
    rlib = _lib.load_reccut()
    cap = 64 * n + 16 * int(n_res.astype(np.int64).sum())
    buf = np.empty(cap, dtype=np.uint8)
    out_off = np.zeros(n + 1, dtype=np.int64)
    nd = np.zeros(n, dtype=np.int32)
    needs = np.zeros(n, dtype=np.uint8)
    ret = rlib.reccut_format_packed(n, penc.numpy().ctypes.data, enc_off.ctypes.data, buf.ctypes.data, cap, out_off.ctypes.data,
                                    nd.ctypes.data, needs.ctypes.data)
    if ret != 0:
        raise RuntimeError(f'reccut_format_packed failed: {ret}')
    text = buf[:int(out_off[-1])].tobytes().decode('ascii')
    bounds = out_off.tolist()
    doms = [text[a:b].split(';')[:-1] for a, b in zip(bounds[:-1], bounds[1:])]
    redo = np.flatnonzero(needs)
    LAST.host_redo = redo.tolist()
    if len(redo):      # the host library on these proteins' own contacts (copied over now: they are few)
        sel_off = np.zeros(len(redo) + 1, dtype=np.int64)
        np.cumsum(counts[redo], out=sel_off[1:])
        pick = np.concatenate([np.arange(offs[p], offs[p + 1]) for p in redo]) if sel_off[-1] else np.zeros(0, np.int64)
        pick_t = torch.from_numpy(pick).to(device)
        hi, hj, hv = (x[pick_t].cpu().numpy() for x in (oi, oj, ov))
        for p, d in zip(redo.tolist(), domains_from_contacts(n_res[redo], sel_off, hi, hj, hv, cut1, cut2, threads=threads)):
            doms[p] = d
    return doms


def reccut_room(n_res: np.ndarray) -> np.ndarray:
    """``dctfp_reccut_room`` for an array of lengths (the C function is the definition; tests hold this to it)."""
    n = np.asarray(n_res, dtype=np.int64)
    doms = np.where(n > 0, n // 22 + 1, 1)
    return 2 + doms + 2 * (2 * doms + 1)


class CutInFlight:
    """Contact selection + domain cutter of a batch, enqueued on ``stream`` (``dctfp_contact_topk`` + ``dctfp_reccut``), the
    encoded results on their way into a page-locked buffer: what a database flush starts early and picks up when it needs
    the domains (``make_db._Flush``).  ``ptrs`` / ``lds`` / ``n_res``: the contact maps' geometry (``_geom.tensor_table``)."""

    def __init__(self, ptrs, lds, n_res, device, t: float, cut1=CUT1_DEFAULT, cut2=CUT2_DEFAULT, stream=None, slot: int = 0,
                 timing: bool = False):
        lib = _lib.load()
        self.events = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timing else None    # (tools/flush_timeline.py)
        n = len(n_res)
        self.n = n
        self.n_res = np.ascontiguousarray(n_res, dtype=np.int32)
        L64 = self.n_res.astype(np.int64)
        cand = np.where(L64 >= 6, (L64 - 5) * (L64 - 4) // 2, 0)
        self.counts = np.minimum(np.maximum((float(t) * L64.astype(np.float64)).astype(np.int64), 0), cand)
        self.offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(self.counts, out=self.offs[1:])
        total = int(self.offs[-1])
        self.cut1, self.cut2 = float(cut1), float(cut2)
        self.stream = stream if stream is not None else torch.cuda.current_stream(device)
        ptrs = np.ascontiguousarray(ptrs, dtype=np.uint64)
        lds = np.ascontiguousarray(lds, dtype=np.int64)
        with torch.cuda.stream(self.stream):
            self.oi = torch.empty(max(total, 1), dtype=torch.int32, device=device)
            self.oj = torch.empty(max(total, 1), dtype=torch.int32, device=device)
            self.ov = torch.empty(max(total, 1), dtype=torch.float32, device=device)
            on = torch.zeros(n, dtype=torch.int32, device=device)
            ctx = _lib.get_context(device.index)
            sp = C.c_void_p(self.stream.cuda_stream)
            if timing:
                self.events[0].record(self.stream)
            _lib.check(lib.dctfp_contact_topk(ctx.handle, ptrs.ctypes.data, lds.ctypes.data, self.n_res.ctypes.data, n, float(t),
                                              self.oi.data_ptr(), self.oj.data_ptr(), self.ov.data_ptr(), self.offs.ctypes.data,
                                              on.data_ptr(), sp), lib)
            if timing:
                self.events[1].record(self.stream)
            room = reccut_room(self.n_res)
            self.enc_off = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(room, out=self.enc_off[1:])
            n_enc = int(self.enc_off[-1])
            enc = torch.empty(n_enc, dtype=torch.int32, device=device)
            _lib.check(lib.dctfp_reccut(ctx.handle, self.n_res.ctypes.data, n, self.oi.data_ptr(), self.oj.data_ptr(), self.ov.data_ptr(),
                                        self.offs.ctypes.data, self.cut1, self.cut2, enc.data_ptr(), self.enc_off.ctypes.data, sp), lib)
            if timing:
                self.events[2].record(self.stream)
            self.penc = _pinned(f'enc{slot}', torch.int32, n_enc)
            self.penc.copy_(enc, non_blocking=True)
            self.pn = _pinned(f'n{slot}', torch.int32, n)
            self.pn.copy_(on, non_blocking=True)
            if timing:
                self.events[3].record(self.stream)
            self.done = torch.cuda.Event()
            self.done.record(self.stream)
        self._keep = (enc, on)

    def wait(self) -> np.ndarray:
        """The encoded results (host view, valid until this thread's next batch in the same slot); proteins the GPU cutter
        handed back (status -1) are redone by the host library here and written into the same encoding."""
        self.done.synchronize()
        if self.events is not None:
            e = self.events
            LAST.gpu_ms = (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3]))   # top-k, cutter, copies
        if not (self.pn.numpy() == self.counts).all():
            raise RuntimeError('dctfp_contact_topk wrote a different number of contacts than dctfp_contact_count says')
        enc = self.penc.numpy()
        redo = np.flatnonzero(enc[self.enc_off[:-1]] < 1)
        LAST.host_redo = redo.tolist()
        self.redo_strings = {}
        if len(redo):
            sel_off = np.zeros(len(redo) + 1, dtype=np.int64)
            np.cumsum(self.counts[redo], out=sel_off[1:])
            pick = np.concatenate([np.arange(self.offs[p], self.offs[p + 1]) for p in redo]) if sel_off[-1] else np.zeros(0, np.int64)
            pick_t = torch.from_numpy(pick).to(self.oi.device)
            hi, hj, hv = (x[pick_t].cpu().numpy() for x in (self.oi, self.oj, self.ov))
            for p, d in zip(redo.tolist(), domains_from_contacts(self.n_res[redo], sel_off, hi, hj, hv, self.cut1, self.cut2)):
                self.redo_strings[p] = d
                rec = _encode_domains(d)
                a, b = int(self.enc_off[p]), int(self.enc_off[p + 1])
                if rec is not None and len(rec) <= b - a:
                    enc[a:a + len(rec)] = rec           # (else: status stays -1 and the caller parses the strings itself)
        return enc


def _encode_domains(doms: List[str]):
    """Domain strings of the binary ("b-e[,b-e]*", 1-based inclusive) in ``dctfp_reccut``'s encoding, or None for anything else."""
    rec = [len(doms)]
    try:
        for d in doms:
            segs = d.split(',')
            rec.append(len(segs))
            for sg in segs:
                b, e = sg.split('-')
                if not (b.isdigit() and e.isdigit()):
                    return None
                rec += [int(b) - 1, int(e) - 1]
    except ValueError:
        return None
    return np.asarray(rec, dtype=np.int32) if doms else None


def domains_from_maps(maps: Sequence[torch.Tensor], t: float, cut1=CUT1_DEFAULT, cut2=CUT2_DEFAULT, threads: int = 1,
                      before_wait=None) -> List[List[str]]:
    """``Fingerprint.reccut``'s domain lists for a batch of contact maps with NOTHING but the answer leaving the GPU: the contact
    selection (``dctfp_contact_topk``) and the domain cutter's recursion (``dctfp_reccut``: src/RecCut.cpp:150-351, one
    workgroup per protein) run back to back on the device; what comes over is a few ints per domain, which libreccut formats
    into the binary's strings.  Proteins the GPU cutter hands back (status -1: longer than its tables, or a step the
    reference leaves undefined) go through the host library on their own contacts.  ``before_wait`` (a callable) runs after
    the kernels are enqueued and before this thread waits for them -- a flush builds its embedding tables there."""
    if len(maps) == 0:
        return []
    n_res, counts, offs, oi, oj, ov, on = _select_on_device(maps, t)
    return _cut_on_device(n_res, counts, offs, oi, oj, ov, on, cut1, cut2, threads, before_wait)


def domains_from_contacts_gpu(n_res: Sequence[int], offs, ci, cj, cv, cut1=CUT1_DEFAULT, cut2=CUT2_DEFAULT, threads: int = 1,
                              device=None) -> List[List[str]]:
    """``domains_from_contacts`` with the recursion on the GPU (``dctfp_reccut``): the same contact lists (host arrays, pairs
    distinct) -> the same strings."""
    n_res = np.ascontiguousarray(n_res, dtype=np.int32)
    if len(n_res) == 0:
        return []
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    oi = torch.from_numpy(np.ascontiguousarray(ci, dtype=np.int32)).to(device)
    oj = torch.from_numpy(np.ascontiguousarray(cj, dtype=np.int32)).to(device)
    ov = torch.from_numpy(np.ascontiguousarray(cv, dtype=np.float32)).to(device)
    if oi.numel() == 0:
        oi, oj, ov = (torch.zeros(1, dtype=d, device=device) for d in (torch.int32, torch.int32, torch.float32))
    return _cut_on_device(n_res, np.diff(offs), offs, oi, oj, ov, None, cut1, cut2, threads, None)


def ce_text(pid: str, seq: str, ci, cj, cv) -> str:
    """The .ce file body of src/fingerprint.py:69-80."""
    slen = len(seq)
    sout = ''
    if len(ci):
        sout = 'CON   ' + ','.join(f'{int(i)} {int(j)} {float(v):.6f}' for i, j, v in zip(ci, cj, cv))
    return f'INF   {pid} {slen}\nSEQ   {seq}\nSS    {"C" * slen}\n{sout}\n'


def write_ce(fp, outfile: str, t: float):
    """``Fingerprint.writece`` (src/fingerprint.py:45-80): same file, byte for byte."""
    slen = len(fp.seq)
    cmap = _contact_tensor(fp.contacts, slen)
    offs, ci, cj, cv = top_contacts_batch([cmap], t)
    with open(outfile, 'w', encoding='utf8') as out_f:
        out_f.write(ce_text(fp.pid, fp.seq, ci, cj, cv))


def domains_from_contacts(n_res: Sequence[int], offs, ci, cj, cv, cut1=CUT1_DEFAULT, cut2=CUT2_DEFAULT,
                          threads: int = 1) -> List[List[str]]:
    """libreccut on a batch: per protein the list the reference gets from
    ``stdout.strip().split()[2].split(';')[:-1]`` (src/fingerprint.py:103)."""
    lib = _lib.load_reccut()
    n = len(n_res)
    n_res = np.ascontiguousarray(n_res, dtype=np.int32)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    cj = np.ascontiguousarray(cj, dtype=np.int32)
    cv = np.ascontiguousarray(cv, dtype=np.float32)
    if n == 0:
        return []
    cap = 64 * n + 16 * int(n_res.astype(np.int64).sum())      # a domain piece "b-e," is at most 12 bytes per residue it holds
    buf = np.empty(cap, dtype=np.uint8)
    out_off = np.zeros(n + 1, dtype=np.int64)
    nd = np.zeros(n, dtype=np.int32)
    rc = np.zeros(n, dtype=np.int32)
    ret = lib.reccut_predict_packed(n, n_res.ctypes.data, offs.ctypes.data, ci.ctypes.data, cj.ctypes.data,
                                    cv.ctypes.data, float(cut1), float(cut2), buf.ctypes.data, cap, out_off.ctypes.data,
                                    nd.ctypes.data, rc.ctypes.data, int(threads))
    if ret != 0:
        raise RuntimeError(f'reccut_predict_packed failed: {ret}')
    if rc.any():
        p = int(np.flatnonzero(rc)[0])
        # the reference would raise CalledProcessError (RecCut exit != 0) or crash
        raise RuntimeError(f'reccut: protein {p}: error {int(rc[p])} '
                           f'({"undefined behaviour in the reference at this input" if rc[p] == -3 else "invalid input"})')
    text = buf[:int(out_off[-1])].tobytes().decode('ascii')
    bounds = out_off.tolist()
    return [text[a:b].split(';')[:-1] for a, b in zip(bounds[:-1], bounds[1:])]


def predict_domains(fp, threshold: float) -> List[str]:
    """The domain list ``Fingerprint.reccut`` appends (src/fingerprint.py:83-104), without the
    trailing whole-protein entry (the caller adds it when there are several domains)."""
    slen = len(fp.seq)
    cmap = _contact_tensor(fp.contacts, slen)
    offs, ci, cj, cv = top_contacts_batch([cmap], threshold, sort=False, own=False)
    return domains_from_contacts([slen], offs, ci, cj, cv)[0]
