"""Drop-in for mgtools/DCTdomain ``src/dct-sim.py``: similarity between proteins from their DCT
fingerprints (``-dct.npz``), with the L1 distances computed on the GPU.

    python -m dctdomain_amd.dct_sim --dct X-dct.npz [--pair P | --db Y-dct.npz] [--output F]
                                    [--pairfound F] [--top 5] [--threshold 0.25]

Same flags, same output text (src/dct-sim.py:179-211).  DCTdomain = max over all domain pairs of
``1 - min(L1/17000, 1)``, DCTglobal = the same for the two last (whole-protein) fingerprints
(:12-50).  Scores are formed from the integer L1 distances with the reference's own Python
expressions, so the printed floats are identical."""

from __future__ import annotations

import argparse
import time
from operator import itemgetter

import numpy as np

from .similarity import block_min, l1_matrix


def prostSimilarity(emb1, emb2) -> float:
    """1 - min(L1 / 17000, 1) of two fingerprints (src/dct-sim.py:12-26)."""
    d = l1_matrix(np.asarray(emb1)[None, :], np.asarray(emb2)[None, :]).cpu().numpy()[0, 0].astype(np.int64)
    return _sim(d)


def _sim(d):
    d = d / 17000
    d = min(d, 1)
    return 1 - d


def _scores(mn, last):
    """(maxs, s) as ``domain_sim`` returns them: ``maxs`` starts at int 0 and only a strictly larger
    similarity replaces it (src/dct-sim.py:42-50)."""
    best = _sim(np.int64(mn))
    maxs = best if best > 0 else 0
    return maxs, _sim(np.int64(last))


def domain_sim(dct_i: np.ndarray, dct_j: np.ndarray) -> tuple:
    """(DCTdomain, DCTglobal) of two proteins' fingerprint sets (src/dct-sim.py:28-50)."""
    dist = l1_matrix(dct_i, dct_j)
    mn, last = block_min(dist, [0, dct_i.shape[0]], [0, dct_j.shape[0]])
    return _scores(mn[0, 0], last[0, 0])


def load_dct(filename: str, asmap=True) -> tuple:
    """npz -> ({sid: fingerprints} | [fingerprints], sid) (src/dct-sim.py:52-84)."""
    start = time.time()
    data = np.load(filename)
    seqid, domidx, dct_all = data['sid'], data['idx'], data['dct']
    dct = {} if asmap else []
    for i in range(len(seqid)):
        ai = dct_all[domidx[i]:domidx[i + 1], :]
        if asmap:
            dct[seqid[i]] = ai
        else:
            dct.append(ai)
    print(f"dct loaded for {len(seqid)} sequences, time used: {time.time() - start:.1f}s")
    return dct, seqid


def _block_scores(file_a: str, file_b: str = None):
    """All protein-vs-protein (min, last) L1 blocks between two npz files, one GPU pass."""
    da = np.load(file_a)
    db = da if file_b is None else np.load(file_b)
    dist = l1_matrix(da['dct'], db['dct'])
    mn, last = block_min(dist, da['idx'], db['idx'])
    return da['sid'], db['sid'], mn, last


def _emit(line: str, output: str):
    if output:
        with open(output, 'a', encoding='utf8') as out:
            out.write(line + '\n')
    else:
        print(line)


def pair_sim(npzfile: str, pairfile: str, pairfound: str, output: str):
    """Similarity of every listed protein pair (src/dct-sim.py:86-124)."""
    load_dct(npzfile, asmap=True)
    sid, _, mn, last = _block_scores(npzfile)
    pos = {}
    for i, s in enumerate(sid):
        pos[s] = i                      # a later duplicate id wins, like the dict in load_dct
    tot, totfound = 0, 0
    out2 = open(pairfound, 'w', encoding='utf8') if pairfound else None
    with open(pairfile, 'r', encoding='utf8') as inf:
        for aline in inf:
            if aline[0] == '#':
                if out2:
                    out2.write(aline)
                continue
            subs = aline.split()
            s1, s2 = subs[0], subs[1]
            tot += 1
            if s1 in pos and s2 in pos:
                maxs, s = _scores(mn[pos[s1], pos[s2]], last[pos[s1], pos[s2]])
                _emit(f'{s1} {s2} {maxs} {s}', output)
                if out2:
                    out2.write(aline)
                totfound += 1
    print(f'total pair {pairfile} found {totfound} (not found: {tot - totfound})')
    if out2:
        print(f'pairs saved to file {pairfound}')
        out2.close()


def db_search(npzfile: str, dbfile: str, top: int, threshold: float, output: str):
    """Top hits of every query protein in a fingerprint database (src/dct-sim.py:126-156)."""
    load_dct(npzfile, asmap=False)
    load_dct(dbfile, asmap=False)
    seqid, db_seqid, mn, last = _block_scores(npzfile, dbfile)
    for i in range(len(seqid)):
        results = []
        for q in range(len(db_seqid)):
            maxs, s = _scores(mn[i, q], last[i, q])
            results.append([db_seqid[q], maxs, s])
        results_sorted = sorted(results, key=itemgetter(2), reverse=True)
        for q in range(len(db_seqid)):
            if (q >= top) and (results_sorted[q][2] < threshold):
                break
            hit = results_sorted[q]
            _emit(f'{seqid[i]} {hit[0]} {hit[1]} {hit[2]}', output)


def all_sim(npzfile: str, output: str):
    """All-against-all (src/dct-sim.py:158-176)."""
    load_dct(npzfile, asmap=False)
    seqid, _, mn, last = _block_scores(npzfile)
    n = len(seqid)
    for i in range(n - 1):
        for j in range(i + 1, n):
            maxs, s = _scores(mn[i, j], last[i, j])
            _emit(f'{seqid[i]} {seqid[j]} {maxs:.3f} {s:.3f}', output)


def main(argv=None):
    start = time.time()
    parser = argparse.ArgumentParser()
    parser.add_argument('--dct', help='dct in a npz file', required=True)
    parser.add_argument('--output', help='save results to a file', required=False)
    parser.add_argument('--pair', help='calculate distance between the proteins in the given file', required=False)
    parser.add_argument('--pairfound', help='pairs of proteins with similarity computed', required=False)
    parser.add_argument('--db', help='search query dct against this db', required=False)
    parser.add_argument('--top', help='report at most this many hits for database search', default=5, type=int)
    parser.add_argument('--threshold', help='similarity threshold for reporting hits for database search',
                        default=0.25, type=float)
    args = parser.parse_args(argv)
    if args.output:
        with open(args.output, 'w', encoding='utf8') as out:
            out.write('#prot1 prot2 sim-domain sim-global\n')
    else:
        print('#prot1 prot2 sim-domain sim-global')
    nowt = time.time()
    if args.pair:
        pair_sim(args.dct, args.pair, args.pairfound, args.output)
    elif args.db:
        db_search(args.dct, args.db, args.top, args.threshold, args.output)
    else:
        all_sim(args.dct, args.output)
    if args.output:
        print('results saved to', args.output)
    end = time.time()
    print(f'total time used {end - start:.1f}s')
    print(f'distance calculation used {end - nowt:.1f}s')


if __name__ == '__main__':
    main()
