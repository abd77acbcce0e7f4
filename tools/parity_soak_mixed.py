"""Parity soak of the multi-domain path: N random proteins with RecCut-shaped domain lists (parts that tile the protein,
discontinuous parts, + the whole protein) and pfam-like lengths through the GPU path (fused walks of the walk kernel at
D = 640 / 1280 / 2560) and through the faithful CPU oracle (scipy.fft, like the reference) on
the host cores; counts mismatching int8 values.  Checker use of oracle/ only (a test tool, not product code).
usage: python tools/parity_soak_mixed.py [n_proteins] [procs] [D] [m]   (m: kept channels, default 80; 85 = PROST's, walk_ab_kernel with six column groups)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def make_protein(seed, D):
    rng = np.random.default_rng(seed)
    L = int(np.clip(rng.gamma(2.2, 170.0), 81, 1330)) if D <= 1280 else int(rng.integers(100, 501))
    k = max(1, min(int(round(L / 110 + rng.normal(0, 0.7))), L // 30))
    if k == 1:
        doms = [f'1-{L}']
    else:
        cuts = sorted(set(int(c) for c in rng.integers(25, L - 25, size=k - 1)))
        edges = [0] + cuts + [L]
        edges = [e for i, e in enumerate(edges) if i == 0 or e == L or e - edges[i - 1] >= 22]
        if L - edges[-2] < 22:
            edges.pop(-2)
        parts = [f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:])]
        if len(parts) >= 3 and rng.random() < 0.3:
            parts = [parts[-1] + ',' + parts[0]] + parts[1:-1]
        doms = parts + [f'1-{L}'] if len(parts) > 1 else [f'1-{L}']
    ls = []
    for _ in range(2):
        x = rng.standard_normal((L, D)) * np.exp(rng.standard_normal(D)) + 5 * rng.standard_normal(D)
        idx = rng.choice(D, size=D // 100, replace=False)
        x[:, idx] += 200.0 * rng.choice([-1.0, 1.0], size=len(idx))
        ls.append(x.astype(np.float32))
    return L, doms, ls


def _oracle_chunk(args):
    seed0, count, D, M = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import dct_oracle as orc
    rows = []
    for i in range(count):
        L, doms, ls = make_protein(seed0 + i, D)
        q = orc.quantize(ls, doms, [3, M, 3, M])
        rows.extend(np.asarray(q[k]).astype(np.int8) for k in q)
    return np.stack(rows)


if __name__ == '__main__':
    import multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    M = int(sys.argv[4]) if len(sys.argv) > 4 else 80
    t0 = time.time()
    B = 512                                   # proteins per GPU call (several thousand jobs: the batch path)
    jobs = [(50_000 + b0, min(B, n - b0), D, M) for b0 in range(0, n, B)]
    with mp.get_context('spawn').Pool(procs) as pool:
        async_res = pool.map_async(_oracle_chunk, jobs, chunksize=1)
        import torch
        import dctdomain_amd as dd
        got = []
        paths = set()
        for seed0, count, _, _ in jobs:
            prots = [make_protein(seed0 + i, D) for i in range(count)]
            lens = [p[0] for p in prots]
            layers = [torch.from_numpy(np.concatenate([p[2][k] for p in prots])).cuda() for k in range(2)]
            table = dd.PieceTable(lens, [p[1] for p in prots])
            offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            got.append(dd.quantize_batch([dd.LayerBatch(t, 3, M, row_offsets=offs) for t in layers], table).cpu().numpy())
            paths.add(dd.get_context(0).get_option('last_path'))
            print(f'  GPU side: {seed0 - 50_000 + count} / {n} proteins, {time.time() - t0:.0f} s', flush=True)
        got = np.concatenate(got)
        exp = np.concatenate(async_res.get())
    assert got.shape == exp.shape, (got.shape, exp.shape)
    bad_vals = int((got != exp).sum())
    bad_fps = int((got != exp).any(axis=1).sum())
    print(f'{n} proteins -> {len(got)} fingerprints ({got.size} int8 values), D={D}, qdim [3, {M}] x 2 layers, RecCut-shaped domain lists, ESM-like '
          f'with +-200 offset channels, quantize path(s) {sorted(paths)} (2 = walk kernel): {bad_fps} mismatching fingerprints, '
          f'{bad_vals} mismatching values; {time.time() - t0:.0f} s')
