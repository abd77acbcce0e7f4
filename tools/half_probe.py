import os, sys, time
sys.path.insert(0, '/root/repo')
import tools._experiments  # noqa: F401  (engineering knobs: libdctfp_experiments.so unless DCTFP_LIBRARY says otherwise)
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
n, L, D = 10000, 500, 1280
layers = [torch.randn((n * L, D), device=dev).to(torch.float16) for _ in range(2)]
lengths = np.full(n, L, dtype=np.int64)
offs = np.arange(n, dtype=np.int64) * L
table = dd.PieceTable.whole_sequences(lengths)
lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
out = torch.empty((n, 480), dtype=torch.int8, device=dev)
nbytes = 2 * n * L * D * 2
for rnd in range(2):
    for cfg in ({'path': 1}, {'path': 2}, {'path': 2, 'ab_run_jobs': 4}, {'path': 2, 'ab_run_jobs': 2}):
        ctx.set_option('ab_run_jobs', 0)
        for k, v in cfg.items(): ctx.set_option(k, v)
        for _ in range(3): dd.quantize_batch(lbs, table, out=out)
        ctx.set_option('profile', 1); ctx.profile()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): dd.quantize_batch(lbs, table, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        ms, nl = ctx.profile(); ctx.set_option('profile', 0)
        print(cfg, f'step {1e3*dt:.3f} ms = {nbytes/dt/1e9:.0f} GB/s  A {ms[0]/10:.3f} ms ({nl[0]//10} launches)  B {ms[1]/10:.3f} ms', flush=True)
