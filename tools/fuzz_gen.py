"""Seeded fuzz of the general walk kernel (walk_gen_kernel): any kept size n = 2 .. 8 x m = 2 .. 128, widths that are not a
multiple of 4 / of 64, float32 and float64 rows, rows taken as column slices of a wider matrix (row stride > D, base not
16-byte aligned: one channel per lane), whole proteins / parts + whole / discontinuous parts.  Every call runs three times:
default dispatch, path = 2 (the general kernel wherever its LDS slot fits) and path = 1 (stage A -> Y' -> stage B, the
path the round-1..3 soaks pinned); all rows of the three must be the same bytes, and a sample of the proteins is checked
against the faithful CPU oracle on the host cores.
Checker use of oracle/ only (a test tool, not product code).   usage: python tools/fuzz_gen.py [n_cases] [procs] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np


def _input(L, D, seed, dtype):
    from recipes import make_input
    x = make_input('esm' if seed % 3 else 'gauss', L, D, seed).astype(dtype)
    if dtype == np.float64:   # bits a float32 does not have: the kernel must read all 8 bytes
        x = x * (1.0 + np.random.default_rng(seed).random(x.shape) * 2.0 ** -30)
    return x


def _oracle(args):
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import dct_oracle as orc
    seeds, L, D, doms, qd, dtype = args
    xs = [_input(L, D, sd, dtype) for sd in seeds]
    q = orc.quantize(xs, doms, qd)
    return [(k, v) for k, v in q.items()]


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    import multiprocessing as mp
    pool = mp.get_context('spawn').Pool(procs)
    import torch
    import dctdomain_amd as dd
    ctx = dd.get_context(0)
    rng = np.random.default_rng(seed)
    bad = bad_paths = total = checked = 0
    used = {}
    t0 = time.time()
    for case in range(n_cases):
        dtype = np.float64 if rng.random() < 0.3 else np.float32
        D = int(rng.choice([130, 200, 257, 320, 333, 512, 640, 641, 777, 1000, 1280, 1283, 2050, 2560]))
        same = rng.random() < 0.6
        qd = []
        for li in range(2):
            if li == 1 and same:
                qd += qd[:2]
            else:
                qd += [int(rng.integers(2, 9)), int(min(D, rng.choice([2, 5, 16, 33, 44, 64, 80, 85, 100, 127, 128])))]
        n_max = max(qd[0::2])
        n_seq = int(rng.choice([140, 200, 300]))
        style = rng.random()          # whole proteins / parts + whole / arbitrary windows
        sliced = rng.random() < 0.35  # rows = columns [c0, c0 + D) of a wider matrix
        lens, doms, seeds = [], [], []
        for s in range(n_seq):
            L = int(rng.integers(n_max + 4, 160 if D <= 1280 else 90))
            if style < 0.4:
                d = [f'1-{L}']
            elif style < 0.8:
                k = int(rng.integers(2, 5))
                lo = n_max + 1
                cuts = sorted(set(int(c) for c in rng.integers(lo, max(lo + 1, L - lo), size=k - 1)))
                e = [0] + [c for c in cuts if c < L] + [L]
                e = [v for i, v in enumerate(e) if i == 0 or v == L or v - e[i - 1] >= lo]
                if len(e) > 2 and L - e[-2] < lo:
                    e.pop(-2)
                parts = [f'{a + 1}-{b}' for a, b in zip(e[:-1], e[1:])]
                if len(parts) >= 3 and rng.random() < 0.4:
                    parts = [parts[-1] + ',' + parts[0]] + parts[1:-1]
                d = parts + [f'1-{L}'] if len(parts) > 1 else [f'1-{L}']
            else:
                a = int(rng.integers(1, max(2, L - n_max - 2)))
                d = [f'{a}-{L}', f'1-{max(n_max + 1, L - 3)}']
            lens.append(L)
            doms.append(d)
            seeds.append([7_000_000 * seed + 1000 * case + 2 * s, 7_000_000 * seed + 1000 * case + 2 * s + 1])
        c0 = int(rng.integers(1, 8)) if sliced else 0
        pad = int(rng.integers(1, 9)) if sliced else 0
        lbs, keep = [], []
        for li in range(2):
            ts = []
            for L, sd in zip(lens, seeds):
                x = torch.from_numpy(_input(L, D, sd[li], dtype))
                if sliced:
                    wide = torch.zeros((L, c0 + D + pad), dtype=x.dtype)
                    wide[:, c0:c0 + D] = x
                    wide = wide.cuda()
                    keep.append(wide)
                    ts.append(wide[:, c0:c0 + D])
                else:
                    ts.append(x.cuda())
            lbs.append(dd.LayerBatch(ts, qd[2 * li], qd[2 * li + 1]))
        table = dd.PieceTable(lens, doms)
        outs, paths = [], []
        for path in (0, 2, 1):
            ctx.set_option('path', path)
            try:
                outs.append(dd.quantize_batch(lbs, table, ctx=ctx).cpu().numpy())
                paths.append(ctx.get_option('last_path'))
            finally:
                ctx.set_option('path', 0)
        used[tuple(paths)] = used.get(tuple(paths), 0) + 1
        differ = int((outs[0] != outs[2]).any(axis=1).sum()) + int((outs[1] != outs[2]).any(axis=1).sum())
        bad_paths += differ
        sample = list(range(0, n_seq, 6))
        res = pool.map(_oracle, [(seeds[s], lens[s], D, doms[s], qd, dtype) for s in sample])
        first = np.zeros(n_seq + 1, dtype=np.int64)
        np.cumsum([len(d) for d in doms], out=first[1:])   # (every domain here is kept: no empty piece lists)
        mism = 0
        for s, rows in zip(sample, res):
            for j, (key, exp) in enumerate(rows):
                r = int(first[s]) + j
                assert table.keys[r] == key
                mism += int((outs[1][r].astype(np.int64) != exp).any())
                checked += 1
        bad += mism
        total += table.n_domains
        print(f'case {case:3d}: D={D:4d} {np.dtype(dtype).name} qd={qd} {"sliced" if sliced else "dense "} {n_seq} proteins {table.n_domains:4d} fingerprints '
              f'paths (default, 2, 1) = {paths}: {differ} rows differ between paths, {mism} of the sampled rows differ from the oracle   ({time.time() - t0:.0f} s)', flush=True)
    print(f'{n_cases} calls x 3 dispatches, {total} fingerprints per dispatch, paths used {used}: {bad_paths} rows differ between dispatches; '
          f'{checked} rows checked against the oracle: {bad} mismatching')
    pool.close()


if __name__ == '__main__':
    main()
