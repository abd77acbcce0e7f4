"""The general walk kernel against the two-kernel path on the multi-domain mixes (short jobs): which one should the
dispatch pick below which job length?  usage: python tools/gen_probe.py [workload ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
import bench

dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
nseq = {'c2': 10000, 'c3': 10000, 'c4': 12000, 'c5': 40000}
for w in (sys.argv[1:] or ['c5', 'c4', 'c3']):
    sys.argv = ['bench.py', '--workload', w, '--n-seq', str(nseq[w])]
    a = bench.parse()
    lengths, doms, D = bench.make_workload(a, 0, np)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    layers = [bench.make_layer(torch, gen, int(lengths.sum()), D, dev) for _ in range(2)]
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    table = dd.PieceTable.whole_sequences(lengths) if doms is None else dd.PieceTable(lengths, doms)
    nbytes = 2 * int(lengths.sum()) * D * 4
    print(f'{w}  D={D}  {table.n_domains} fingerprints, mean rows per job {float(np.mean(table.lengths)):.0f}')
    for n, m in ((5, 44), (3, 85), (4, 80)):
        lbs = [dd.LayerBatch(x, n, m, row_offsets=offs) for x in layers]
        out = torch.empty((table.n_domains, 2 * n * m), dtype=torch.int8, device=dev)
        res, outs = {}, {}
        for path in (1, 2, 3):                  # 3: the general kernel with every job on its own (round 4's form: "gen_fuse" 0)
            ctx.set_option('path', min(path, 2))
            ctx.set_option('gen_fuse', 0 if path == 3 else 1)
            for _ in range(2):
                dd.quantize_batch(lbs, table, out=out, ctx=ctx)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                dd.quantize_batch(lbs, table, out=out, ctx=ctx)
            torch.cuda.synchronize()
            res[path] = 5 * nbytes / (time.perf_counter() - t0) / 1e9
            assert ctx.get_option('last_path') == min(path, 2)
            fused = ctx.get_option('last_gen_fused')
            if path == 2:
                groups = ctx.get_option('last_walk_groups')
            outs[path] = out.clone()
        ctx.set_option('path', 0)
        ctx.set_option('gen_fuse', 1)
        print(f'   [{n},{m}]  two kernels {res[1]:6.0f} GB/s   ' + (f'walk_ab_kernel ({groups} column groups) {res[2]:6.0f} GB/s   ' if groups else
              f'general walk kernel: fused walks {res[2]:6.0f} GB/s, every job on its own {res[3]:6.0f} GB/s   ') +
              f''
              f'identical={bool((outs[1] == outs[2]).all() and (outs[1] == outs[3]).all())}', flush=True)
