"""A/B of two builds of libdctfp.so in ONE process on the SAME allocation (the only comparison that
is not drowned by the +-4 % allocation-to-allocation spread):
    python tools/ab_libs.py /tmp/libA.so /tmp/libB.so [workload] [n_seq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
from dctdomain_amd import _lib
paths = sys.argv[1:3]
sys.argv = ['bench.py', '--workload', sys.argv[3] if len(sys.argv) > 3 else 'c2', '--n-seq', sys.argv[4] if len(sys.argv) > 4 else '10000']
import bench
args = bench.parse()
lengths, doms, D = bench.make_workload(args, 0, np)
dev = torch.device('cuda', 0)
ctxs = [_lib.Context(0, _lib.load(p)) for p in paths]
offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
table = dd.PieceTable.whole_sequences(lengths) if doms is None else dd.PieceTable(lengths, doms)
out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
nbytes = 2 * int(lengths.sum()) * D * 4
for alloc in range(3):
    layers = [torch.randn((int(lengths.sum()), D), device=dev) for _ in range(2)]
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    res = [[], []]
    outs = []
    for blk in range(6):
        w = blk % 2
        for _ in range(2):
            dd.quantize_batch(lbs, table, out=out, ctx=ctxs[w])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            dd.quantize_batch(lbs, table, out=out, ctx=ctxs[w])
        torch.cuda.synchronize()
        res[w].append(10 * nbytes / (time.perf_counter() - t0) / 1e9)
        if blk < 2:
            outs.append(out.clone())
    same = bool((outs[0] == outs[1]).all())
    print(f'alloc {alloc}: A {[round(r) for r in res[0]]}  B {[round(r) for r in res[1]]} GB/s whole-path; identical output: {same}', flush=True)
    del layers, lbs
    torch.cuda.empty_cache()
