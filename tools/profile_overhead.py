"""Does the hipEvent bracketing of the kernels ("profile" option) perturb the step time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0); ctx = dd.get_context(0)
n_seq, L, D = 10000, 500, 1280
layers = [torch.randn((n_seq * L, D), device=dev) for _ in range(2)]
offs = np.arange(n_seq, dtype=np.int64) * L
table = dd.PieceTable.whole_sequences([L] * n_seq)
lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
out = torch.empty((n_seq, 480), dtype=torch.int8, device=dev)
for _ in range(3):
    dd.quantize_batch(lbs, table, out=out, ctx=ctx)
for blk in range(8):
    prof = blk % 2
    ctx.set_option('profile', prof); ctx.profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ctx.profile()
    print(f'profile={prof}: {1e3 * dt:.3f} ms/step', flush=True)
