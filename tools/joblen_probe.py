"""Stage-A rate against job length and launch shape: 500-row proteins cut into k equal single-piece domains
(no whole protein -> plain, non-fused jobs of 500/k rows), a_waves swept.  Separates the per-job cost of
short jobs from everything else (same bytes, same allocation for every case)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._experiments  # noqa: F401  (engineering knobs: libdctfp_experiments.so unless DCTFP_LIBRARY says otherwise)
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0)
n_seq, L, D = 8000, 500, int(sys.argv[1]) if len(sys.argv) > 1 else 1280
layers = [torch.randn((n_seq * L, D), device=dev) for _ in range(2)]
offs = np.arange(n_seq, dtype=np.int64) * L
lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
ctx = dd.get_context(0)
nbytes = 2 * n_seq * L * D * 4
print('rows/job ' + ' '.join(f'waves={w:<5}' for w in ('auto', 1, 2, 4, 8)) + '  (stage A GB/s | stage B ms)')
for k in (1, 2, 5, 10, 20):
    e = [round(i * L / k) for i in range(k + 1)]
    doms = [f'{a + 1}-{b}' for a, b in zip(e[:-1], e[1:])]
    table = dd.PieceTable([L] * n_seq, [doms] * n_seq)
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    cells = []
    for w in (0, 1, 2, 4, 8):
        ctx.set_option('a_waves', w)
        for _ in range(2):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        ctx.set_option('profile', 1); ctx.profile()
        torch.cuda.synchronize()
        for _ in range(5):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize()
        ms, nl = ctx.profile(); ctx.set_option('profile', 0)
        cells.append(f'{nbytes / (ms[0] / 5 * 1e-3) / 1e9:5.0f}|{ms[1] / 5:5.2f}')
    ctx.set_option('a_waves', 0)
    print(f'{L // k:8d} ' + ' '.join(f'{c:>11}' for c in cells), flush=True)
