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
from .similarity import l1_matrix, row_select


def _load_all(db: Database):
    rows = db.cur.execute(""" SELECT vid, pid, domain, fingerprint FROM fingerprints """).fetchall()
    fps = np.array([np.load(BytesIO(r[3]), allow_pickle=True) for r in rows], dtype=np.int8)
    return rows, fps


def search(query_rows, query_fps, db_rows, db_fps, khits: int):
    """Yields the reference's log lines.  Per query protein (pids in the order ``SELECT pid FROM
    sequences`` returns them: by primary key, i.e. sorted): the ``khits`` nearest database
    fingerprints of each of its fingerprints (ties: lower vid first, as a flat index scans), then
    all of those ranked by distance (stable) and the first ``khits`` printed (:33-59)."""
    dist = l1_matrix(query_fps, db_fps)                                   # (nq, ndb) int32 on the GPU
    k = min(khits, db_fps.shape[0])
    dm, im = row_select(dist, k)                                          # k nearest per query fingerprint
    by_pid = {}
    for qi, r in enumerate(query_rows):
        by_pid.setdefault(r[1], []).append(qi)
    for pid in sorted(by_pid):
        qis = by_pid[pid]
        items = [((i, j), dm[qi, j]) for i, qi in enumerate(qis) for j in range(k)]
        items.sort(key=lambda x: x[1])
        for rank, ((i, j), d) in enumerate(items[:khits]):
            qrow = query_rows[qis[i]]
            drow = db_rows[im[qis[i], j]]
            score = round(1 - (d / 17000), 4)
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


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--query', type=str, required=True, help='can be .fa or .db file')
    parser.add_argument('--db', type=str, required=True, help='fingerprint database (.db)')
    parser.add_argument('--out', type=str, default=False, help='output file')
    parser.add_argument('--maxlen', type=int, default=500, help='max sequence length to embed')
    parser.add_argument('--khits', type=int, default=100, help='number of hits to return')
    parser.add_argument('--cpu', type=int, default=1, help='number of cpus to use')
    parser.add_argument('--gpu', type=int, default=False, help='number of gpus to use')
    parser.add_argument('--model', choices=['esm', 'synthetic'], default='esm')
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
