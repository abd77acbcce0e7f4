"""Where does the multi-domain workload lose bandwidth?  Same embeddings (L = 500, D = 1280, 2 layers),
different domain lists: isolates the cost of the fused variant from the cost of short parts."""
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
for arg in sys.argv[2:]:   # an overlap count, or name=value options of the experiments library (ab_mfma_a=1 ...)
    if '=' in arg:
        ctx.set_option(arg.split('=')[0], int(arg.split('=')[1]))
    else:
        ctx.set_option('overlap', int(arg))
def parts(k):
    e = [round(i * L / k) for i in range(k + 1)]
    return [f'{a + 1}-{b}' for a, b in zip(e[:-1], e[1:])]
cases = {'whole only (C2)': [f'1-{L}'], '2 parts + whole': parts(2) + [f'1-{L}'], '5 parts + whole': parts(5) + [f'1-{L}'],
         '10 parts + whole': parts(10) + [f'1-{L}'], '20 parts + whole': parts(20) + [f'1-{L}'],
         '5 parts, no whole': parts(5), '20 parts, no whole': parts(20)}
nbytes = 2 * n_seq * L * D * 4
for name, doms in cases.items():
    table = dd.PieceTable([L] * n_seq, [doms] * n_seq)
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    for _ in range(3):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    ctx.set_option('profile', 1); ctx.profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ms, nl = ctx.profile(); ctx.set_option('profile', 0)
    print(f'{name:22s} {table.n_domains:7d} fingerprints  step {1e3 * dt:7.3f} ms  whole {nbytes / dt / 1e9:6.0f} GB/s  '
          f'stage A sum {ms[0] / 10:7.3f} ms = {nbytes / (ms[0] / 10 * 1e-3) / 1e9:6.0f} GB/s   stage B sum {ms[1] / 10:6.3f} ms', flush=True)
