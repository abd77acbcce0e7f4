"""Host-side cost of one dctfp_quantize call (table build + enqueue) vs the GPU time of the batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
sys.argv = ['bench.py', '--workload', sys.argv[1] if len(sys.argv) > 1 else 'c5', '--n-seq', sys.argv[2] if len(sys.argv) > 2 else '40000']
import bench
args = bench.parse()
lengths, doms, D = bench.make_workload(args, 0, np)
dev = torch.device('cuda', 0)
total = int(lengths.sum())
layers = [torch.randn((total, D), device=dev) for _ in range(2)]
offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
t0 = time.perf_counter()
table = dd.PieceTable.whole_sequences(lengths) if doms is None else dd.PieceTable(lengths, doms)
print('PieceTable (python):', round(1e3 * (time.perf_counter() - t0), 1), 'ms for', table.n_domains, 'domains')
lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
ctx = dd.get_context(0)
for _ in range(3):
    dd.quantize_batch(lbs, table, out=out, ctx=ctx)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(10):
    t0 = time.perf_counter()
    dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print('host time per call (ms):', [round(1e3 * h, 2) for h in host])
print('wall per step incl. GPU (ms):', round(1e2 * (time.perf_counter() - t_all), 2))
