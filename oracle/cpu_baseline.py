"""CPU baseline worker for bench.py -- TEST/MEASUREMENT INFRASTRUCTURE, not product code.

Times the *faithful* oracle form (same scipy.fft calls and per-row scale loop as the
reference's Fingerprint.quantize, src/fingerprint.py:174-201) on the bench workload's
shape, one process per core like the reference's multiprocessing.Pool
(src/make_db.py:48-49).  Inputs are generated inside each worker, so no embedding is
pickled in the timed region (which favours the CPU side relative to make_db.py)."""

from __future__ import annotations

import os
import time


def _worker(args):
    seed, n_rows, n_cols, n_layers, qdim, seconds, form = args
    os.environ['OMP_NUM_THREADS'] = '1'
    os.environ['OPENBLAS_NUM_THREADS'] = '1'
    import numpy as np
    from oracle import dct_oracle as orc
    rng = np.random.default_rng(seed)
    pool = []
    for _ in range(4):        # a few distinct proteins, cycled
        layers = [(rng.standard_normal((n_rows, n_cols)) * np.exp(rng.standard_normal(n_cols))
                   + 5 * rng.standard_normal(n_cols)).astype(np.float32) for _ in range(n_layers)]
        pool.append(layers)
    dom = [f'1-{n_rows}']
    quantize = orc.quantize if form == 'faithful' else orc.quantize_matrix
    quantize(pool[0], dom, qdim)              # warm-up
    done = 0
    t0 = time.perf_counter()
    while True:
        quantize(pool[done % len(pool)], dom, qdim)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            return done, el


def usable_cores(cap=64):
    """Cores this process may really use: affinity mask, clipped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fh2:
                        n = min(n, max(1, q // int(fh2.read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, cap))


def _timed(pool, procs, n_rows, n_cols, n_layers, qdim, seconds, form):
    res = pool.map(_worker, [(100 + i, n_rows, n_cols, n_layers, list(qdim), seconds, form) for i in range(procs)])
    return sum(r[0] for r in res), sum(r[0] / r[1] for r in res)


def run(n_rows=500, n_cols=1280, n_layers=2, qdim=(3, 80, 3, 80), seconds=12.0, procs=None, matrix_seconds=4.0):
    """Returns dict(value=fingerprints/s over all workers, cores=procs, sample=...).  ``value`` is the faithful form (what
    the reference runs); ``matrix_form`` is the same result computed the way a tuned CPU code would (two small float64
    matrix products per layer through numpy/BLAS, one thread per process) -- the fairer "good CPU" line of SURVEY 8d."""
    import multiprocessing as mp
    if procs is None:
        procs = usable_cores()
    ctx = mp.get_context('spawn')
    with ctx.Pool(procs) as pool:
        total, rate = _timed(pool, procs, n_rows, n_cols, n_layers, qdim, seconds, 'faithful')
        extra = None
        if matrix_seconds > 0:
            mtotal, mrate = _timed(pool, procs, n_rows, n_cols, n_layers, qdim, matrix_seconds, 'matrix')
            extra = {'value': mrate, 'unit': 'fingerprints/s', 'cores': procs,
                     'sample': f'{mtotal} fingerprints in {matrix_seconds:.0f} s; oracle matrix form (numpy float64 matmul)'}
    out = {'value': rate, 'unit': 'fingerprints/s', 'cores': procs, 'kind': 'port',
           'sample': f'{total} fingerprints (L={n_rows}, D={n_cols}, {n_layers} layers, qdim {list(qdim)}) in '
                     f'{seconds:.0f} s on {procs} processes; oracle faithful form (scipy.fft dct/idct + per-row scale loop)'}
    if extra:
        out['matrix_form'] = extra
    return out
