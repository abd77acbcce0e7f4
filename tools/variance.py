"""Is the stage-A bandwidth spread between runs a property of the allocation (physical layout)
or of time (clocks)?  Several timed blocks per allocation, several allocations per process."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._experiments  # noqa: F401  (engineering knobs: libdctfp_experiments.so unless DCTFP_LIBRARY says otherwise)
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
n_seq, L, D = 10000, 500, 1280
offs = np.arange(n_seq, dtype=np.int64) * L
table = dd.PieceTable.whole_sequences([L] * n_seq)
out = torch.empty((n_seq, 480), dtype=torch.int8, device=dev)
for alloc in range(4):
    layers = [torch.randn((n_seq * L, D), device=dev) for _ in range(2)]
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    for _ in range(3):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    res = []
    for blk in range(6):
        ctx.set_option('overlap', 1 if blk % 2 == 0 else 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize()
        res.append(10 * n_seq * 5120480 / (time.perf_counter() - t0) / 1e9)
    # reference stream on the same allocation: torch's own read-only reduction
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        for x in layers:
            x.sum()
    torch.cuda.synchronize()
    ref = 5 * 2 * n_seq * L * D * 4 / (time.perf_counter() - t0) / 1e9
    print('   torch.sum read stream GB/s:', round(ref))
    print('alloc', alloc, 'ptrs', [hex(x.data_ptr()) for x in layers], 'whole-path GB/s per block (overlap 1,4,1,4,1,4):', [round(r) for r in res], flush=True)
    del layers, lbs
    torch.cuda.empty_cache()
