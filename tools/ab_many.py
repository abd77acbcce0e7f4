"""A/B/C... of several builds (and option sets) of libdctfp.so in ONE process on the SAME allocation, interleaved round
after round (boxes and allocations differ by +-4 %, builds often by less):

    python tools/ab_many.py build_variants/base.so dctdomain_amd/libdctfp.so@ab_narrow=2 build_variants/d2.so -- c2 c4 c5

`lib.so@name=value,name=value` sets options on that library's context.  Prints GB/s (algorithmic bytes / whole-path step
time) per round, the median, and whether the int8 output is identical to the first configuration's."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
from dctdomain_amd import _lib
import bench

argv = sys.argv[1:]
split = argv.index('--') if '--' in argv else len(argv)
specs, workloads = argv[:split], argv[split + 1:] or ['c2', 'c4', 'c5']
rounds = int(os.environ.get('AB_ROUNDS', '3'))
nseq = {'c2': 10000, 'c3': 10000, 'c4': 12000, 'c5': 40000}
dev = torch.device('cuda', 0)
cfgs = []
libs = {}
for spec in specs:
    path, _, opts = spec.partition('@')
    if path not in libs:
        libs[path] = _lib.Context(0, _lib.load(os.path.abspath(path)))
    cfgs.append((spec, libs[path], [kv.split('=') for kv in opts.split(',') if kv]))
for w in workloads:
    sys.argv = ['bench.py', '--workload', w, '--n-seq', str(nseq[w])]
    a = bench.parse()
    lengths, doms, D = bench.make_workload(a, 0, np)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    layers = [bench.make_layer(torch, gen, int(lengths.sum()), D, dev) for _ in range(2)]
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    table = dd.PieceTable.whole_sequences(lengths) if doms is None else dd.PieceTable(lengths, doms)
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    nbytes = 2 * int(lengths.sum()) * D * 4
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    res = {spec: [] for spec, _, _ in cfgs}
    ref, same = None, {}
    for rnd in range(rounds):
        for spec, ctx, opts in cfgs:
            saved = {k: ctx.get_option(k) for k, _ in opts}
            for k, v in opts:
                ctx.set_option(k, int(v))
            for _ in range(2):
                dd.quantize_batch(lbs, table, out=out, ctx=ctx)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                dd.quantize_batch(lbs, table, out=out, ctx=ctx)
            torch.cuda.synchronize()
            res[spec].append(10 * nbytes / (time.perf_counter() - t0) / 1e9)
            if rnd == 0:
                if ref is None:
                    ref = out.clone()
                same[spec] = bool((ref == out).all())
            for k, v in saved.items():
                ctx.set_option(k, v)
    print(f'{w}  D={D}  {table.n_domains} fingerprints, {nbytes / 1e9:.1f} GB per step')
    for spec, _, _ in cfgs:
        r = res[spec]
        print(f'  {spec:60s} {statistics.median(r):6.0f} GB/s   rounds {[round(x) for x in r]}   identical={same[spec]}', flush=True)
    del layers, lbs, out, ref
    torch.cuda.empty_cache()
