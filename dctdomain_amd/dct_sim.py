"""Drop-in for mgtools/DCTdomain ``src/dct-sim.py``: similarity between proteins from their DCT
fingerprints (``-dct.npz``), with the L1 distances computed on the GPU.

    python -m dctdomain_amd.dct_sim --dct X-dct.npz [--pair P | --db Y-dct.npz] [--output F]
                                    [--pairfound F] [--top 5] [--threshold 0.25]

Same flags, same output text (src/dct-sim.py:179-211).  DCTdomain = max over all domain pairs of
``1 - min(L1/17000, 1)``, DCTglobal = the same for the two last (whole-protein) fingerprints
(:12-50).

Where the reference loops over protein pairs and, inside, over domain pairs in Python, this module
makes ONE pass on the GPU per run: the int8 L1 matrix of all fingerprints against all fingerprints
(``dctfp_l1_matrix``) reduced per protein x protein block to (minimum, last-last) (``dctfp_block_min``).
The three modes only differ in which blocks they print.  Scores are formed from the integer L1
values with the reference's arithmetic (int64 / 17000 in float64), so the printed floats are identical."""

from __future__ import annotations

import argparse
import sys
import time

import numpy as np

from .similarity import block_min, l1_matrix, to_device_int8

L1_FULL_SCALE = 17000      # src/dct-sim.py:24
HEADER = '#prot1 prot2 sim-domain sim-global'


def _sim(l1):
    """1 - min(L1 / 17000, 1) for arrays of L1 values (ranking only; printing goes through ``_sim1``)."""
    return 1 - np.minimum(np.asarray(l1, dtype=np.int64) / L1_FULL_SCALE, 1)


def _sim1(l1):
    """The same for one value, with the reference's scalar arithmetic (src/dct-sim.py:24-26): Python's
    ``min`` hands back the int 1 once L1 exceeds 17000, so such a score prints as ``0``, not ``0.0``."""
    return 1 - min(np.int64(l1) / L1_FULL_SCALE, 1)


def _scores(mn, last):
    """(DCTdomain, DCTglobal) of one block.  The reference's running maximum starts at the int 0 and
    is replaced only by a strictly larger similarity (:42-50)."""
    best = _sim1(mn)
    return (best if best > 0 else 0), _sim1(last)


def prostSimilarity(emb1, emb2) -> float:
    """Similarity of two single fingerprints (src/dct-sim.py:12-26)."""
    d = l1_matrix(np.asarray(emb1)[None, :], np.asarray(emb2)[None, :]).cpu().numpy()[0, 0]
    return _sim1(d)


def domain_sim(dct_i: np.ndarray, dct_j: np.ndarray) -> tuple:
    """(DCTdomain, DCTglobal) of two proteins' fingerprint sets (src/dct-sim.py:28-50)."""
    mn, last = block_min(l1_matrix(dct_i, dct_j), [0, dct_i.shape[0]], [0, dct_j.shape[0]])
    return _scores(mn[0, 0], last[0, 0])


def load_dct(filename: str, asmap=True) -> tuple:
    """npz -> ({sid: fingerprints} | [fingerprints], sid) (src/dct-sim.py:52-84)."""
    t0 = time.time()
    with np.load(filename) as data:
        seqid, bounds, rows = data['sid'], data['idx'], data['dct']
    per_protein = [rows[a:b, :] for a, b in zip(bounds[:-1], bounds[1:])]
    print(f"dct loaded for {len(seqid)} sequences, time used: {time.time() - t0:.1f}s")
    return (dict(zip(seqid, per_protein)) if asmap else per_protein), seqid


def _protein_groups(idx, max_rows: int):
    """[p0, p1) ranges of consecutive proteins whose fingerprints (idx = prefix offsets) number at most ``max_rows``
    -- at least one protein per range, however many fingerprints it has."""
    p0, n = 0, len(idx) - 1
    while p0 < n:
        p1 = p0 + 1
        while p1 < n and idx[p1 + 1] - idx[p0] <= max_rows:
            p1 += 1
        yield p0, p1
        p0 = p1


class Blocks:
    """All protein-vs-protein (minimum, last-last) L1 blocks between two ``-dct.npz`` files."""

    def __init__(self, file_a: str, file_b: str = None):
        load_dct(file_a, asmap=False)                       # the reference's progress line(s)
        if file_b is not None:
            load_dct(file_b, asmap=False)
        a = np.load(file_a)
        b = a if file_b is None else np.load(file_b)
        self.rows, self.cols = a['sid'], b['sid']
        # protein stripes of `a`: the int32 distance matrix of a stripe stays within ~1 GiB (the protein x protein result
        # is what is kept; the reference loops pair by pair, src/dct-sim.py:126-176)
        # ... and protein groups of `b` of at most COL_ROWS fingerprints, so that neither the uploaded part of `b` nor
        # the distance matrix grows with the size of the files
        da, ia = a['dct'], np.asarray(a['idx'], dtype=np.int64)
        dbm, ib = b['dct'], np.asarray(b['idx'], dtype=np.int64)
        na, nb = len(ia) - 1, len(ib) - 1
        self.mn = np.full((na, nb), 0x7fffffff, dtype=np.int32)      # (an empty block keeps this: block_min_kernel's fill)
        self.last = np.full((na, nb), 0x7fffffff, dtype=np.int32)
        for q0, q1 in _protein_groups(ib, self.COL_ROWS):
            db_dev = to_device_int8(dbm[ib[q0]:ib[q1]])
            budget = max(1, self.TILE_INTS // max(1, db_dev.shape[0]))
            for p0, p1 in _protein_groups(ia, budget):
                if ia[p1] > ia[p0] and db_dev.shape[0] > 0:           # (a stripe of proteins without fingerprints: nothing to launch)
                    mn_t, last_t = block_min(l1_matrix(da[ia[p0]:ia[p1]], db_dev), ia[p0:p1 + 1] - ia[p0], ib[q0:q1 + 1] - ib[q0])
                    self.mn[p0:p1, q0:q1] = mn_t
                    self.last[p0:p1, q0:q1] = last_t
            del db_dev

    COL_ROWS = 1 << 22      # fingerprints of `b` on the device at a time (2 GB of int8 at 480 columns)
    TILE_INTS = 1 << 28     # int32 entries of one distance matrix (1 GiB)

    def scores(self, i: int, j: int) -> tuple:
        return _scores(self.mn[i, j], self.last[i, j])


class Report:
    """Result lines to a file (header first) or to stdout."""

    def __init__(self, path: str = None):
        self.path = path
        self.out = open(path, 'w', encoding='utf8') if path else sys.stdout
        self.line(HEADER)

    def line(self, text: str):
        self.out.write(text + '\n')

    def close(self):
        if self.path:
            self.out.close()
            print('results saved to', self.path)
        else:
            self.out.flush()


def _reporting(fn):
    """The mode functions keep the reference's signature -- the last argument may be an output path or
    ``None`` (stdout) -- and also take an open ``Report``."""
    def run(*args):
        *head, output = args
        if isinstance(output, Report):
            return fn(*head, output)
        report = Report(output)
        try:
            return fn(*head, report)
        finally:
            report.close()
    run.__doc__, run.__name__ = fn.__doc__, fn.__name__
    return run


@_reporting
def pair_sim(npzfile: str, pairfile: str, pairfound: str, report: Report):
    """Similarity of every listed protein pair (src/dct-sim.py:86-124).  Lines starting with ``#``
    are comments (copied to ``pairfound``); pairs with an unknown protein are counted, not printed."""
    blk = Blocks(npzfile)
    where = {name: i for i, name in enumerate(blk.rows)}      # a repeated id: the later one, like a dict of arrays
    listed = found = 0
    kept = []
    with open(pairfile, encoding='utf8') as pairs:
        for text in pairs:
            if text.startswith('#'):
                kept.append(text)
                continue
            first, second = text.split()[:2]
            listed += 1
            if first not in where or second not in where:
                continue
            maxs, s = blk.scores(where[first], where[second])
            report.line(f'{first} {second} {maxs} {s}')
            kept.append(text)
            found += 1
    print(f'total pair {pairfile} found {found} (not found: {listed - found})')
    if pairfound:
        with open(pairfound, 'w', encoding='utf8') as out:
            out.writelines(kept)
        print(f'pairs saved to file {pairfound}')


@_reporting
def db_search(npzfile: str, dbfile: str, top: int, threshold: float, report: Report):
    """Hits of every query protein in a fingerprint database, best DCTglobal first (stable); the first
    ``top`` always, further ones while they reach ``threshold`` (src/dct-sim.py:126-156)."""
    blk = Blocks(npzfile, dbfile)
    glob = _sim(blk.last)                                     # (n_query, n_db) float64
    for i, query in enumerate(blk.rows):
        order = np.argsort(-glob[i], kind='stable')
        for rank, q in enumerate(order):
            if rank >= top and glob[i, q] < threshold:
                break
            maxs, s = blk.scores(i, q)
            report.line(f'{query} {blk.cols[q]} {maxs} {s}')


@_reporting
def all_sim(npzfile: str, report: Report):
    """All-against-all, upper triangle (src/dct-sim.py:158-176)."""
    blk = Blocks(npzfile)
    n = len(blk.rows)
    for i, j in zip(*np.triu_indices(n, k=1)):
        maxs, s = blk.scores(i, j)
        report.line(f'{blk.rows[i]} {blk.rows[j]} {maxs:.3f} {s:.3f}')


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description='protein similarity from DCT fingerprints (GPU L1)')
    ap.add_argument('--dct', required=True, help='fingerprints of the proteins to compare (-dct.npz)')
    ap.add_argument('--output', help='write the result lines here instead of stdout')
    ap.add_argument('--pair', help='file of protein pairs to score (two ids per line)')
    ap.add_argument('--pairfound', help='copy of --pair restricted to the pairs that were scored')
    ap.add_argument('--db', help='search every protein of --dct in this -dct.npz')
    ap.add_argument('--top', type=int, default=5, help='database search: hits always reported per query')
    ap.add_argument('--threshold', type=float, default=0.25, help='database search: further hits down to this DCTglobal')
    return ap


def main(argv=None):
    t_start = time.time()
    args = build_parser().parse_args(argv)
    report = Report(args.output)
    t_work = time.time()
    if args.pair:
        pair_sim(args.dct, args.pair, args.pairfound, report)
    elif args.db:
        db_search(args.dct, args.db, args.top, args.threshold, report)
    else:
        all_sim(args.dct, report)
    report.close()
    t_end = time.time()
    print(f'total time used {t_end - t_start:.1f}s')
    print(f'distance calculation used {t_end - t_work:.1f}s')


if __name__ == '__main__':
    main()
