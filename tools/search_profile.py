"""query_db.search end to end on synthetic fingerprints: n_query fingerprints (four per protein) against n_db, --khits 100 --
kernels, ordering, copies and the reference's per-protein ranking and result lines.   usage: python tools/search_profile.py [n_query] [n_db]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dctdomain_amd import query_db
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 6700
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
rng = np.random.default_rng(5)
def fps(n):
    x = np.cumsum(rng.standard_normal((n, 6, 80)), axis=2)
    mn, mx = x.min(axis=2, keepdims=True), x.max(axis=2, keepdims=True)
    return np.round((x - mn) / (mx - mn) * 254 - 127).astype(np.int8).reshape(n, 480)
q, db = fps(nq), fps(nd)
qrows = [(i + 1, f'q{i // 4:05d}', f'{1 + 10 * (i % 4)}-{9 + 10 * (i % 4)}') for i in range(nq)]
drows = [(i + 1, f'd{i // 4:05d}', f'{1 + 10 * (i % 4)}-{9 + 10 * (i % 4)}') for i in range(nd)]
n = sum(1 for _ in query_db.search(qrows, q, drows, db, 100))          # warm-up
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); n = sum(1 for _ in query_db.search(qrows, q, drows, db, 100)); pr.disable()
print(f'{nq} query fingerprints x {nd}: {n} result lines in {time.perf_counter() - t0:.2f} s')
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print('\n'.join(s.getvalue().splitlines()[4:26]))
