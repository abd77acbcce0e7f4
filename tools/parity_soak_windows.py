"""Parity soak of BASELINE config 3 as stated: N random sequences of L ~ U[50, 2000] given as the language model's windows
(maxlen 500, overlap 200; every window its own ESM-like random matrix, so overlapping windows disagree on the rows they share),
multi-domain lists on a third of them (cuts anywhere, also inside shared rows: fused walks), through `dctfp_quantize_windows`
on the GPU and through the CPU oracles chained (stitch_oracle.stitch_embeddings -> dct_oracle.quantize, the faithful scipy form)
on the host cores; counts mismatching int8 values.  Checker use of oracle/ only.
usage: python tools/parity_soak_windows.py [n] [procs] [D]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

MAXLEN, OVERLAP = 500, 200


def windows_of(L):
    if L <= MAXLEN:
        return [L]
    w = [min(MAXLEN, L - i) for i in range(0, L, MAXLEN - OVERLAP)]
    return [v for v in w if v > OVERLAP]


def make_case(seed, D):
    """(L, window matrices per layer, domain strings) of sequence `seed`."""
    rng = np.random.default_rng(seed)
    L = int(rng.integers(50, 2001))
    rows = windows_of(L)
    scale = np.exp(rng.standard_normal(D))
    off = 5 * rng.standard_normal(D)
    off[rng.choice(D, size=D // 100, replace=False)] += 200.0
    layers = [[(rng.standard_normal((r, D)) * scale + off).astype(np.float32) for r in rows] for _ in range(2)]
    doms = [f'1-{L}']
    if seed % 3 == 0 and L >= 60:
        k = int(rng.integers(2, 7))
        cuts = sorted(set(int(c) for c in rng.integers(8, L - 8, size=k - 1)))
        edges = [0] + cuts + [L]
        edges = [e for i, e in enumerate(edges) if i == 0 or e - edges[i - 1] >= 3 or e == L]
        doms = [f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:]) if b - a >= 3] + [f'1-{L}']
    return L, rows, layers, doms


def _oracle_chunk(args):
    seed0, count, D = args
    os.environ['OMP_NUM_THREADS'] = '1'
    import torch
    torch.set_num_threads(1)
    from oracle import dct_oracle as orc
    from oracle import stitch_oracle as sto
    out = []
    for i in range(count):
        L, rows, layers, doms = make_case(seed0 + i, D)
        mats = [sto.stitch_embeddings([torch.from_numpy(w) for w in lay], OVERLAP).numpy() for lay in layers]
        q = orc.quantize(mats, doms, [3, 80, 3, 80])
        out.append(np.stack([q[k].astype(np.int8) for k in q]))
    return out


if __name__ == '__main__':
    import multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
    t0 = time.time()
    per = (n + procs - 1) // procs
    jobs = [(50_000 + p * per, min(per, n - p * per), D) for p in range(procs) if p * per < n]
    with mp.get_context('spawn').Pool(procs) as pool:
        async_res = pool.map_async(_oracle_chunk, jobs)
        import torch
        import dctdomain_amd as dd
        from dctdomain_amd.batch import window_geometry
        ctx = dd.get_context(0)
        got = []
        B = 256
        n_two = 0
        for b0 in range(0, n, B):
            bn = min(B, n - b0)
            cases = [make_case(50_000 + b0 + i, D) for i in range(bn)]
            win_rows = [r for c in cases for r in c[1]]
            counts = [len(c[1]) for c in cases]
            lbs = [dd.LayerBatch([torch.from_numpy(w).cuda() for c in cases for w in c[2][k]], 3, 80) for k in range(2)]
            _, sizes = window_geometry(win_rows, counts, OVERLAP)
            assert sizes.tolist() == [c[0] for c in cases]
            table = dd.PieceTable(sizes, [c[3] for c in cases])
            out = dd.quantize_windows(lbs, win_rows, counts, table, overlap=OVERLAP, fallback=False).cpu().numpy()
            assert ctx.get_option('last_path') == 2
            bounds = np.searchsorted(table.owner, np.arange(bn + 1))
            got += [out[bounds[s]:bounds[s + 1]] for s in range(bn)]
            n_two += sum(1 for c in counts if c > 1)
            print(f'  GPU side: {b0 + bn} / {n} done, {time.time() - t0:.0f} s', flush=True)
        exp = [e for chunk in async_res.get() for e in chunk]
    bad_fp = bad_val = n_fp = 0
    for g, e in zip(got, exp):
        assert g.shape == e.shape, (g.shape, e.shape)
        n_fp += len(e)
        bad_val += int((g != e).sum())
        bad_fp += int((g != e).any(axis=1).sum())
    print(f'{n} sequences of L ~ U[50, 2000] as windows (maxlen {MAXLEN}, overlap {OVERLAP}; {n_two} of them in more than one window), D={D}, '
          f'2 layers, a third with multi-domain lists: {n_fp} fingerprints ({n_fp * 480} int8 values) through dctfp_quantize_windows '
          f'against stitch_oracle -> dct_oracle: {bad_fp} mismatching fingerprints, {bad_val} mismatching values; {time.time() - t0:.0f} s')
