"""Drop-in for the search half of mgtools/DCTdomain ``src/query_db.py`` (:17-91): for every query
protein the ``khits`` nearest database fingerprints by L1 distance, printed as
``round(1 - L1/17000, 4)``.  The reference uses a FAISS flat index forced to METRIC_L1 (:75-76);
here the exact L1 matrix comes from the GPU kernel and needs no index file.

    python -m dctdomain_amd.query_db --query Q.fasta|Q.db --db X.db [--out F] [--khits 100]
                                     [--maxlen 500] [--cpu N] [--gpu G] [--model esm|synthetic]
"""

from __future__ import annotations

import argparse
import logging
import os
from io import BytesIO

import numpy as np

from .database import Database
from .similarity import l1_matrix, order_pairs, row_select, to_device_int8


def _load_all(db: Database):
    """Every fingerprint row of a database as ((vid, pid, domain) tuples, int8 matrix), table order."""
    meta, blobs = [], []
    for vid, pid, domain, blob in db.cur.execute('SELECT vid, pid, domain, fingerprint FROM fingerprints'):
        meta.append((vid, pid, domain))
        blobs.append(np.load(BytesIO(blob), allow_pickle=False))      # plain int8 vectors: no reason to unpickle a user's file
    return meta, np.array(blobs, dtype=np.int8)


COL_ROWS = 1 << 22      # database fingerprints on the device at a time (2 GB of int8 at 480 columns)
TILE_INTS = 1 << 28     # int32 entries of one distance matrix (1 GiB)


def ranked_hits(d_rows: np.ndarray, khits: int):
    """The first ``khits`` of all hits of one query protein -- ``d_rows[i, j]`` = distance of hit j of its fingerprint i --
    ranked by distance, stable in (fingerprint, hit) order: the reference's ``sorted(top_hits.items(), key=distance)`` over
    the items as it inserts them (src/query_db.py:33-40).  Returns (fingerprint index, hit index, score) lists, the scores
    as ``round(1 - (d / 17000), 4)`` of numpy scalars (:57), rounded all at once."""
    k = d_rows.shape[1]
    d_all = d_rows.ravel()
    sel = np.argsort(d_all, kind='stable')[:khits]
    scores = np.round(1 - (d_all[sel] / 17000), 4).tolist()
    ii, jj = np.divmod(sel, k)
    return ii.tolist(), jj.tolist(), scores


def search(query_rows, query_fps, db_rows, db_fps, khits: int):
    """Yields the reference's log lines.  Per query protein (pids in the order ``SELECT pid FROM
    sequences`` returns them: by primary key, i.e. sorted): the ``khits`` nearest database
    fingerprints of each of its fingerprints (ties: lower vid first, as a flat index scans), then
    all of those ranked by distance (stable) and the first ``khits`` printed (:33-59)."""
    # query fingerprints in tiles: the (tile, ndb) int32 distance matrix stays within ~1 GiB however large the database
    # is (the reference streams one query protein at a time, src/query_db.py:75-87)
    # ... and the database in column blocks of at most COL_ROWS fingerprints (uploaded one at a time): the k nearest of
    # every block are candidates, the k nearest of the candidates the answer (ties: lower database row first, as before)
    ndb = db_fps.shape[0]
    k = min(khits, ndb)
    nq = len(query_fps)
    cand_d, cand_i = [], []                                 # per column block: (nq, k_block) distances / database rows
    for c0 in range(0, ndb, COL_ROWS):
        db_dev = to_device_int8(db_fps[c0:c0 + COL_ROWS])
        kb = min(k, db_dev.shape[0])
        tile = max(1, min(8192, TILE_INTS // max(1, db_dev.shape[0])))
        dms, ims = [], []
        for q0 in range(0, nq, tile):
            dist = l1_matrix(query_fps[q0:q0 + tile], db_dev)            # (tile, block) int32 on the GPU
            dm_t, im_t = row_select(dist, kb)                             # k nearest per query fingerprint
            dms.append(dm_t)
            ims.append(im_t + c0)
            del dist
        cand_d.append(np.concatenate(dms) if dms else np.zeros((0, kb), np.int64))
        cand_i.append(np.concatenate(ims) if ims else np.zeros((0, kb), np.int64))
        del db_dev
    if len(cand_d) == 1:
        dm, im = cand_d[0], cand_i[0]
    elif cand_d:
        dm, im = order_pairs(np.concatenate(cand_d, axis=1), np.concatenate(cand_i, axis=1))
        dm, im = dm[:, :k], im[:, :k]
    else:
        dm, im = np.zeros((nq, 0), np.int64), np.zeros((nq, 0), np.int64)
    by_pid = {}
    for qi, r in enumerate(query_rows):
        by_pid.setdefault(r[1], []).append(qi)
    for pid in sorted(by_pid):
        qis = by_pid[pid]
        # all hits of the protein's fingerprints ranked by distance, stable in (fingerprint, hit) order -- the reference's
        # list.sort(key=distance) over the items as it appends them (:52-59)
        ii, jj, scores = ranked_hits(dm[qis], khits)
        for rank, (i, j, score) in enumerate(zip(ii, jj, scores)):
            qrow = query_rows[qis[i]]
            drow = db_rows[im[qis[i], j]]
            yield f'Query: {qrow[1]} {qrow[2]}, Result {rank + 1}: {drow[1]} {drow[2]}, Similarity: {score}'


def search_db(args: argparse.Namespace, query_db: str, fp_db: str):
    qdb = Database(query_db)
    fdb = Database(fp_db)
    print('Querying database...\n')
    qrows, qfps = _load_all(qdb)
    drows, dfps = _load_all(fdb)
    for line in search(qrows, qfps, drows, dfps, args.khits):
        logging.info(line)
    qdb.close()
    fdb.close()


def build_parser() -> argparse.ArgumentParser:
    """Flags of src/query_db.py:94-110, plus ``--model`` (see make_db)."""
    ap = argparse.ArgumentParser(description='nearest database fingerprints of every query protein (GPU L1)')
    for flag, kind, default, text in (('--maxlen', int, 500, 'longest window given to the language model'),
                                      ('--khits', int, 100, 'hits reported per query protein'),
                                      ('--cpu', int, 1, 'host processes for RecCut'),
                                      ('--gpu', int, False, 'GPU worker processes')):
        ap.add_argument(flag, type=kind, default=default, help=text)
    ap.add_argument('--query', required=True, help='query proteins: FASTA (.fa/.fasta) or an existing .db')
    ap.add_argument('--db', required=True, help='database to search (.db)')
    ap.add_argument('--out', default=False, help='write the hit lines here instead of the console')
    ap.add_argument('--model', choices=['esm', 'synthetic'], default='esm')
    return ap


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.out:
        logging.basicConfig(level=logging.INFO, filename=args.out, filemode='w', format='%(message)s', force=True)
    else:
        logging.basicConfig(level=logging.INFO, format='%(message)s', force=True)
    query_db = os.path.splitext(args.query)[0] + '.db'
    if not args.query.endswith('.db'):
        from . import make_db
        ns = make_db.build_parser().parse_args(['--fafile', args.query, '--dbfile', os.path.splitext(args.query)[0],
                                                '--maxlen', str(args.maxlen), '--cpu', str(args.cpu), '--model', args.model,
                                                '--noindex', '--nonpz', '--nodom'] + (['--gpu', str(args.gpu)] if args.gpu else []))
        make_db.run(ns).close()
    search_db(args, query_db, args.db)


if __name__ == '__main__':
    main()
