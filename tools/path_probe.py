"""In-process A/B of the two quantize paths on the bench workloads: stage A -> Y' -> stage B (path 1) against the walk
kernel (path 2: both stages in one launch, int8 out), with the walk kernel's knobs.  One context, one allocation per
workload, configurations interleaved (ABAB) so that allocation-to-allocation spread cancels.
usage: python tools/path_probe.py [c2 c3 c4 c5 ...] [--cfg name=value,name=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._experiments  # noqa: F401  (engineering knobs: libdctfp_experiments.so unless DCTFP_LIBRARY says otherwise)
import numpy as np, torch
import dctdomain_amd as dd
import bench

dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
args_w = [a for a in sys.argv[1:] if not a.startswith('--') and '=' not in a] or ['c2', 'c4', 'c5']
cfgs = [a for a in sys.argv[1:] if '=' in a] or ['path=1', 'path=2', 'path=2,ab_unroll=4']
defaults = {k: ctx.get_option(k) for k in ('path', 'ab_group', 'ab_unroll', 'ab_run_jobs', 'ab_longest_first', 'overlap', 'a_waves', 'fuse')}
nseq = {'c2': 10000, 'c3': 10000, 'c4': 12000, 'c5': 40000}
for w in args_w:
    argv, sys.argv = sys.argv, ['bench.py', '--workload', w, '--n-seq', str(nseq[w])]
    a = bench.parse()
    sys.argv = argv
    lengths, doms, D = bench.make_workload(a, 0, np)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    layers = [bench.make_layer(torch, gen, int(lengths.sum()), D, dev) for _ in range(2)]
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    table = dd.PieceTable.whole_sequences(lengths) if doms is None else dd.PieceTable(lengths, doms)
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    nbytes = 2 * int(lengths.sum()) * D * 4
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    ref = None
    for rnd in range(2):
        for cfg in cfgs:
            for k, v in defaults.items():
                ctx.set_option(k, v)
            for kv in cfg.split(','):
                k, v = kv.split('=')
                ctx.set_option(k, int(v))
            for _ in range(3):
                dd.quantize_batch(lbs, table, out=out, ctx=ctx)
            ctx.set_option('profile', 1); ctx.profile()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                dd.quantize_batch(lbs, table, out=out, ctx=ctx)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            ms, nl = ctx.profile(); ctx.set_option('profile', 0)
            if ref is None:
                ref = out.clone()
            same = bool((ref == out).all())
            print(f'{w} {table.n_domains:7d} fp  {cfg:34s} step {1e3 * dt:7.3f} ms = {nbytes / dt / 1e9:5.0f} GB/s  '
                  f'{table.n_domains / dt / 1e6:6.3f} Mfp/s   A {ms[0] / 10:7.3f} ms ({nl[0] // 10} launches)  B {ms[1] / 10:6.3f} ms  same={same}', flush=True)
    del layers, lbs, out, ref
    torch.cuda.empty_cache()
for k, v in defaults.items():
    ctx.set_option(k, v)
