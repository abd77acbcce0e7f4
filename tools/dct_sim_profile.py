"""Where the time of the dct-sim consumers goes on a synthetic -dct.npz of n proteins (about four fingerprints each): the block
matrix (Blocks: uploads, L1 matrix, block minima, copies back) and the database search mode (ranking + lines).
usage: python tools/dct_sim_profile.py [n_proteins]"""
import cProfile, io, os, pstats, sys, tempfile, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dctdomain_amd import dct_sim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
rng = np.random.default_rng(3)
counts = rng.integers(1, 8, size=n)
idx = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
x = np.cumsum(rng.standard_normal((int(idx[-1]), 6, 80)), axis=2)
mn, mx = x.min(axis=2, keepdims=True), x.max(axis=2, keepdims=True)
dct = np.round((x - mn) / (mx - mn) * 254 - 127).astype(np.int8).reshape(-1, 480)
d = tempfile.mkdtemp()
f = os.path.join(d, 'x-dct.npz')
np.savez(f, sid=np.array([f'p{i}' for i in range(n)]), idx=idx, dom=np.array(['1-9'] * int(idx[-1])), dct=dct)
print(n, 'proteins', int(idx[-1]), 'fingerprints')
with contextlib.redirect_stdout(io.StringIO()):
    dct_sim.Blocks(f)                      # warm-up (context, kernels)
for name, fn in (('Blocks (all against all)', lambda: dct_sim.Blocks(f)),
                 ('db_search --top 10 --threshold 0.9', lambda: dct_sim.db_search(f, f, 10, 0.9, os.path.join(d, 'out.txt')))):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable(); fn(); pr.disable()
    print(f'== {name}: {time.perf_counter() - t0:.2f} s')
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(12); print('\n'.join(s.getvalue().splitlines()[4:22]))
