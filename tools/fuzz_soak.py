"""Seeded fuzz over call sizes and shapes that meet every dispatch rule of dctfp_quantize: one protein ... 300 proteins per
call (row-split small calls, stage B over channel slabs, the walk kernel with 1 / 2 / 4 jobs per workgroup), widths around
the walk kernel's limits (D % 8 == 4, D % 32 != 0, 512, 2560, 2564), kept columns 65..80 and shapes the walk kernel does not
take, RecCut-shaped (fused) / arbitrary / discontinuous domain lists -- against the faithful CPU oracle on the host cores.
Checker use of oracle/ only (a test tool, not product code).   usage: python tools/fuzz_soak.py [n_cases] [procs] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np


def _oracle(args):
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import dct_oracle as orc
    from recipes import make_input
    seeds, L, D, dom, qd = args
    out = []
    for li, sd in enumerate(seeds):
        x = make_input('esm', L, D, sd)
        out.append(orc.quantize_matrix([x], [dom], qd[2 * li:2 * li + 2])[orc.split_domain(dom, L)[1]])
    return np.concatenate(out)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    import multiprocessing as mp
    pool = mp.get_context('spawn').Pool(procs)
    import torch
    import dctdomain_amd as dd
    from recipes import make_input
    ctx = dd.get_context(0)
    rng = np.random.default_rng(seed)
    bad = total = 0
    paths = {}
    t0 = time.time()
    for case in range(n_cases):
        D = int(rng.choice([512, 516, 640, 644, 768, 772, 1000, 1280, 1284, 2052, 2560, 2564, 320]))
        n_seq = int(rng.choice([1, 2, 5, 20, 45, 90, 150, 300]))
        walkish = rng.random() < 0.8
        qd = []
        for _ in range(2):
            qd += [3, int(rng.integers(65, 81))] if walkish else [int(rng.integers(2, 7)), int(rng.choice([44, 64, 85, 100]))]
        n_max = max(qd[0::2])
        lens, doms, seeds = [], [], []
        for s in range(n_seq):
            L = int(rng.integers(n_max + 25, 260))
            style = rng.random()
            if style < 0.5:
                k = int(rng.integers(2, 5))
                cuts = sorted(set(int(c) for c in rng.integers(n_max + 2, L - n_max - 2, size=k - 1)))
                e = [0] + cuts + [L]
                e = [v for i, v in enumerate(e) if i == 0 or v == L or v - e[i - 1] >= n_max + 2]
                if L - e[-2] < n_max + 2:
                    e.pop(-2)
                parts = [f'{a + 1}-{b}' for a, b in zip(e[:-1], e[1:])]
                if len(parts) >= 3 and rng.random() < 0.3:
                    parts = [parts[-1] + ',' + parts[0]] + parts[1:-1]
                d = parts + [f'1-{L}'] if len(parts) > 1 else [f'1-{L}']
            elif style < 0.7:
                a = int(rng.integers(1, L - n_max - 10))
                d = [f'{a}-{L}', f'1-{L - 3}']
            else:
                d = [f'1-{L}']
            lens.append(L)
            doms.append(d)
            seeds.append([1_000_000 * seed + 1000 * case + 2 * s, 1_000_000 * seed + 1000 * case + 2 * s + 1])
        lbs = []
        for li in range(2):
            lbs.append(dd.LayerBatch([torch.from_numpy(make_input('esm', L, D, sd[li])).cuda() for L, sd in zip(lens, seeds)], qd[2 * li], qd[2 * li + 1]))
        table = dd.PieceTable(lens, doms)
        out = dd.quantize_batch(lbs, table).cpu().numpy()
        p = ctx.get_option('last_path')
        paths[p] = paths.get(p, 0) + 1
        jobs = [(seeds[s], lens[s], D, dom, qd) for s in range(n_seq) for dom in doms[s]]
        exp = np.stack(pool.map(_oracle, jobs, chunksize=max(1, len(jobs) // (4 * procs))))
        mism = int((out.astype(np.int64) != exp).any(axis=1).sum())
        bad += mism
        total += len(jobs)
        print(f'case {case:3d}: D={D:4d} qd={qd} {n_seq:3d} proteins {len(jobs):4d} fingerprints path {p}: {mism} mismatching   ({time.time() - t0:.0f} s)', flush=True)
    print(f'{n_cases} calls, {total} fingerprints, dispatch {paths} (1 = two kernels / small-call kernels, 2 = walk kernel): {bad} mismatching fingerprints')
    pool.close()


if __name__ == '__main__':
    main()
