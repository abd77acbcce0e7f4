"""Parity soak: N random ESM-like proteins at the headline shape through the GPU path and through the
faithful CPU oracle (scipy.fft, like the reference) on all host cores; counts mismatching int8 values.
Checker use of oracle/ only (this is a test tool, not product code)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def _oracle_chunk(args):
    seed0, count, L, D = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import dct_oracle as orc
    out = np.zeros((count, 480), np.int8)
    for i in range(count):
        ls = make_pair(seed0 + i, L, D)
        out[i] = orc.quantize(ls, [f'1-{L}'], [3, 80, 3, 80])[f'1-{L}'].astype(np.int8)
    return out


def make_pair(seed, L, D):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(2):
        x = rng.standard_normal((L, D)) * np.exp(rng.standard_normal(D)) + 5 * rng.standard_normal(D)
        idx = rng.choice(D, size=D // 100, replace=False)
        x[:, idx] += 200.0 * rng.choice([-1.0, 1.0], size=len(idx))
        out.append(x.astype(np.float32))
    return out


if __name__ == '__main__':
    import multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    L, D = 500, 1280
    t0 = time.time()
    per = (n + procs - 1) // procs
    jobs = [(10_000 + p * per, min(per, n - p * per), L, D) for p in range(procs) if p * per < n]
    with mp.get_context('spawn').Pool(procs) as pool:
        async_res = pool.map_async(_oracle_chunk, jobs)
        import torch
        import dctdomain_amd as dd
        got = np.zeros((n, 480), np.int8)
        B = 256
        for b0 in range(0, n, B):
            bn = min(B, n - b0)
            xs = [make_pair(10_000 + b0 + i, L, D) for i in range(bn)]
            layers = [torch.from_numpy(np.concatenate([x[k] for x in xs])).cuda() for k in range(2)]
            table = dd.PieceTable.whole_sequences([L] * bn)
            offs = np.arange(bn) * L
            got[b0:b0 + bn] = dd.quantize_batch([dd.LayerBatch(t, 3, 80, row_offsets=offs) for t in layers], table).cpu().numpy()
            if (b0 // B) % 8 == 7:
                print(f'  GPU side: {b0 + bn} / {n} done, {time.time() - t0:.0f} s', flush=True)
        exp = np.concatenate(async_res.get())
    bad_vals = int((got != exp).sum())
    bad_fps = int((got != exp).any(axis=1).sum())
    print(f'{n} fingerprints ({n * 480} int8 values), L={L} D={D} 2 layers, ESM-like with +-200 offset channels: '
          f'{bad_fps} mismatching fingerprints, {bad_vals} mismatching values; {time.time() - t0:.0f} s')
