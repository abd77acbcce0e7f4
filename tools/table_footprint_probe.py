"""Does the number of distinct cosine tables matter for whole-protein batches (C3: ~1 950 distinct lengths, one table each,
31 MB of tables behind the scalar cache) -- same row count, lengths drawn from U[50, 2000] against the same lengths rounded
to multiples of 250 (8 tables)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0)
rng = np.random.default_rng(2024)
n_seq, D = 10000, 1280
raw = rng.integers(50, 2001, size=n_seq)
cases = {'U[50,2000] (about 1950 tables)': raw, 'multiples of 250 (8 tables)': np.maximum(250, (raw + 125) // 250 * 250),
         'multiples of 50 (40 tables)': np.maximum(50, (raw + 25) // 50 * 50), 'all 1025 (1 table)': np.full(n_seq, 1025)}
rows_max = int(max(v.sum() for v in cases.values()))
gen = torch.Generator(device=dev); gen.manual_seed(7)
layers = [torch.randn((rows_max, D), device=dev, generator=gen) for _ in range(2)]
for rnd in range(2):
    for name, lengths in cases.items():
        lengths = lengths.astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
        table = dd.PieceTable.whole_sequences(lengths)
        lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
        out = torch.empty((n_seq, 480), dtype=torch.int8, device=dev)
        for _ in range(3):
            dd.quantize_batch(lbs, table, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            dd.quantize_batch(lbs, table, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f'{name:34s} {int(lengths.sum()):9d} rows  step {1e3 * dt:7.3f} ms = {2 * int(lengths.sum()) * D * 4 / dt / 1e9:5.0f} GB/s', flush=True)
