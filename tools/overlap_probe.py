"""(Every config list runs twice per case, ABAB: the first config after a table change is ~1 % slower.)
Two-kernel path (option path = 1): does stage B really run under stage A?  Same embeddings (8000 x 500 x 1280, 2 layers),
domain lists with a growing number of fingerprints per byte; sweeps the stage-A launch shape and the overlap depth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._experiments  # noqa: F401  (engineering knobs: libdctfp_experiments.so unless DCTFP_LIBRARY says otherwise)
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0)
n_seq, L, D = 8000, 500, 1280
layers = [torch.randn((n_seq * L, D), device=dev) for _ in range(2)]
offs = np.arange(n_seq, dtype=np.int64) * L
lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
ctx = dd.get_context(0)
def parts(k):
    e = [round(i * L / k) for i in range(k + 1)]
    return [f'{a + 1}-{b}' for a, b in zip(e[:-1], e[1:])]
cases = {'whole only (C2)': [f'1-{L}'], '5 parts + whole': parts(5) + [f'1-{L}'], '20 parts + whole': parts(20) + [f'1-{L}'],
         '5 parts, no whole': parts(5), '20 parts, no whole': parts(20)}
workload = None
if len(sys.argv) > 1 and sys.argv[1] in ('c4', 'c5'):     # the bench.py mixes instead of the synthetic cases
    workload = sys.argv.pop(1)
configs = [tuple(int(v) for v in (c + ',4,4096').split(',')[:3]) for c in sys.argv[1:]] or [(0, 4, 4096)]     # (a_waves, overlap, workspace_mb)
ctx.set_option('path', 1)
nbytes = 2 * n_seq * L * D * 4
tables = {name: dd.PieceTable([L] * n_seq, [doms] * n_seq) for name, doms in cases.items()}
if workload:
    del layers, lbs
    torch.cuda.empty_cache()
    argv, sys.argv = sys.argv, ['bench.py', '--workload', workload, '--n-seq', '40000' if workload == 'c5' else '12000']
    import bench
    args = bench.parse()
    sys.argv = argv
    lengths, doms, D = bench.make_workload(args, 0, np)
    layers = [torch.randn((int(lengths.sum()), D), device=dev) for _ in range(2)]
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    nbytes = 2 * int(lengths.sum()) * D * 4
    tables = {f'{workload} mix, pass {i}': dd.PieceTable(lengths, doms) for i in range(3)}
for name, table in tables.items():
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    ref = None
    for waves, ov, ws in configs + configs:
        ctx.set_option('workspace_mb', ws)
        ctx.set_option('a_waves', waves); ctx.set_option('overlap', ov)
        for _ in range(3):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        ctx.set_option('profile', 1); ctx.profile()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        ms, nl = ctx.profile(); ctx.set_option('profile', 0)
        if ref is None: ref = out.clone()
        same = bool((ref == out).all())
        print(f'{name:20s} waves {waves} overlap {ov} ws{ws}: step {1e3 * dt:7.3f} ms = {nbytes / dt / 1e9:5.0f} GB/s   '
              f'A sum {ms[0] / 10:7.3f} ms  B sum {ms[1] / 10:6.3f} ms  same={same}', flush=True)
ctx.set_option('a_waves', 0); ctx.set_option('overlap', 4); ctx.set_option('workspace_mb', 4096); ctx.set_option('path', 0)
