"""CPU oracle for the domain-prediction step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates mgtools/DCTdomain src/fingerprint.py:45-107:
* ``top_contacts``  -- the selection inside ``writece`` (:54-67): all pairs (i, j), j >= i+5, sorted
  by value descending with Python's stable sort (ties keep (i, j) ascending order), first
  ``int(t * L)`` kept (all if fewer);
* ``ce_text``       -- the .ce file body (:69-80);
* ``parse_reccut``  -- what ``reccut`` does with the binary's stdout (:103-107).
The domain cutter itself is checked against the reference's own RecCut.cpp, compiled where it
lies into oracle/_ref/RecCut (oracle/Makefile)."""

from __future__ import annotations

import os
import subprocess
import tempfile

import numpy as np

REF_BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'RecCut')


def top_contacts(cmap: np.ndarray, t: float):
    slen = cmap.shape[0]
    ii, jj = np.triu_indices(slen, k=5)              # i-major, j ascending: the reference's append order
    vals = cmap[ii, jj]
    order = np.argsort(-vals.astype(np.float64), kind='stable')
    tot = int(t * slen)
    if tot > len(order):
        tot = len(order)
    sel = order[:tot]
    return ii[sel].astype(np.int32), jj[sel].astype(np.int32), vals[sel].astype(np.float32)


def ce_text(pid: str, seq: str, ci, cj, cv) -> str:
    slen = len(seq)
    sout = ''
    for i, j, v in zip(ci, cj, cv):
        if not sout:
            sout = f'CON   {i} {j} {v:.6f}'
        else:
            sout += f',{i} {j} {v:.6f}'
    return f'INF   {pid} {slen}\nSEQ   {seq}\nSS    {"C" * slen}\n{sout}\n'


def run_ref_binary(text: str, name: str = 'x'):
    """Runs oracle/_ref/RecCut on a .ce text; returns (returncode, stdout)."""
    with tempfile.NamedTemporaryFile('w', suffix='.ce', delete=False) as fh:
        fh.write(text)
        fn = fh.name
    try:
        r = subprocess.run([REF_BIN, '--input', fn, '--name', name], stdout=subprocess.PIPE, text=True)
    finally:
        os.remove(fn)
    return r.returncode, r.stdout


def parse_reccut(stdout: str, seq_len: int):
    """Domain list as ``Fingerprint.reccut`` builds it (:103-107)."""
    domains = stdout.strip().split()[2].split(';')[:-1]
    out = list(domains)
    if len(domains) > 1:
        out.append(f'1-{seq_len}')
    return out
