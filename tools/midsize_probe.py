"""Calls between "a protein" and "a batch": n proteins of L = 500, D = 1280, 2 layers per call (2 n jobs), time per call with
the MFMA stage B (packed Y') against the slab stage B of the small-call path, and against the walk kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._experiments  # noqa: F401  (engineering knobs: libdctfp_experiments.so unless DCTFP_LIBRARY says otherwise)
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
L, D = 500, 1280
layers = [torch.randn((1024 * L, D), device=dev) for _ in range(2)]
for n in (4, 16, 64, 128, 256, 384, 512, 768, 1024):
    lengths = np.full(n, L, dtype=np.int64)
    offs = np.arange(n, dtype=np.int64) * L
    table = dd.PieceTable.whole_sequences(lengths)
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    out = torch.empty((n, 480), dtype=torch.int8, device=dev)
    res = []
    ref = None
    for name, opts in (('mfma B', dict(path=1, small_b_jobs=0)), ('slab B', dict(path=1, small_b_jobs=1 << 20)), ('walk', dict(path=2, small_b_jobs=32, ab_run_jobs=0)), ('walk run=2', dict(path=2, ab_run_jobs=2)),
                       ('walk run=1', dict(path=2, ab_run_jobs=1))):
        for k, v in opts.items():
            ctx.set_option(k, v)
        for _ in range(5):
            dd.quantize_batch(lbs, table, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            dd.quantize_batch(lbs, table, out=out)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        if ref is None: ref = out.clone()
        res.append(f'{name} {1e6 * dt:7.1f} us ({"=" if bool((ref == out).all()) else "DIFF"})')
    print(f'{n:5d} proteins ({2 * n:5d} jobs): ' + '   '.join(res), flush=True)
ctx.set_option('path', 0); ctx.set_option('small_b_jobs', 512); ctx.set_option('ab_run_jobs', 0)
