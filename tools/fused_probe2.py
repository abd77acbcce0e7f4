"""Where do fused walks lose bandwidth?  Same embeddings (8000 x L=500 x D, 2 layers), domain lists with a growing number
of parts (+ whole protein); every part has the same length, so only TWO cosine tables are in use -- against the bench mixes,
where hundreds of lengths are.  Both quantize paths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0)
n_seq, L, D = 8000, 500, int(sys.argv[1]) if len(sys.argv) > 1 else 1280
layers = [torch.randn((n_seq * L, D), device=dev) for _ in range(2)]
offs = np.arange(n_seq, dtype=np.int64) * L
lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
ctx = dd.get_context(0)
def parts(k):
    e = [round(i * L / k) for i in range(k + 1)]
    return [f'{a + 1}-{b}' for a, b in zip(e[:-1], e[1:])]
rng = np.random.default_rng(0)
def ragged(k):      # k parts of random lengths (many tables), per protein
    out = []
    for _ in range(n_seq):
        cuts = sorted(set(int(c) for c in rng.integers(25, L - 25, size=k - 1)))
        e = [0] + cuts + [L]
        e = [v for i, v in enumerate(e) if i == 0 or v == L or v - e[i - 1] >= 22]
        if L - e[-2] < 22: e.pop(-2)
        out.append([f'{a + 1}-{b}' for a, b in zip(e[:-1], e[1:])] + [f'1-{L}'])
    return out
cases = {'whole only': [[f'1-{L}']] * n_seq, '2 equal parts + whole': [parts(2) + [f'1-{L}']] * n_seq,
         '5 equal parts + whole': [parts(5) + [f'1-{L}']] * n_seq, '5 ragged parts + whole': ragged(5),
         '10 equal parts + whole': [parts(10) + [f'1-{L}']] * n_seq, '5 equal parts, no whole': [parts(5)] * n_seq,
         '5 ragged parts, no whole': [d[:-1] for d in ragged(5)]}
nbytes = 2 * n_seq * L * D * 4
for name, doms in cases.items():
    table = dd.PieceTable([L] * n_seq, doms)
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    for path in (1, 2):
        ctx.set_option('path', path)
        for _ in range(3):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f'D={D} {name:26s} {table.n_domains:7d} fp  path {path}: step {1e3 * dt:7.3f} ms = {nbytes / dt / 1e9:6.0f} GB/s', flush=True)
ctx.set_option('path', 0)
