"""What the similarity consumers pay per stage on the GPU: the L1 matrix, the k nearest per row (search of
query_db: --khits 100), the protein x protein block minima (dct-sim) -- on fingerprints with the value distribution of real
ones (int8 rows of min-max scaled DCT blocks).   usage: python tools/sim_probe.py [n_query] [n_db]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import _lib
from dctdomain_amd.similarity import l1_matrix
import ctypes as C

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 6700
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(7)
def fps(n):   # rows of a min-max scaled smooth block: every row touches -127 .. 127, neighbours correlated
    x = torch.cumsum(torch.randn((n, 6, 80), device=dev, generator=g), dim=2)
    mn, mx = x.amin(dim=2, keepdim=True), x.amax(dim=2, keepdim=True)
    return ((x - mn) / (mx - mn) * 254 - 127).round().to(torch.int8).reshape(n, 480)
q, db = fps(nq), fps(nd)
ctx = _lib.get_context(0)
stream = torch.cuda.current_stream(dev)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

dist = l1_matrix(q, db)
t = timed(lambda: l1_matrix(q, db))
print(f'l1_matrix {nq} x {nd} x 480: {1e3 * t:.3f} ms = {nq * nd * 480 / t / 1e12:.1f} T differences/s', flush=True)
for k in (10, 100, 1000):
    val = torch.empty((nq, k), dtype=torch.int32, device=dev); idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
    def sel():
        _lib.check(ctx._lib.dctfp_row_select(ctx.handle, dist.data_ptr(), nq, nd, dist.stride(0), k, val.data_ptr(), idx.data_ptr(),
                                             C.c_void_p(stream.cuda_stream)))
    t = timed(sel)
    # check against torch.topk on a few rows (values only: ties)
    ref = torch.topk(dist[:64].to(torch.int64), k, dim=1, largest=False).values.sort(dim=1).values
    got = val[:64].to(torch.int64).sort(dim=1).values
    print(f'row_select k={k}: {1e3 * t:.3f} ms = {1e6 * t / nq:.2f} us per row, {nq * nd * 4 / t / 1e9:.0f} GB/s of the matrix per pass-equivalent; '
          f'values equal torch.topk on 64 rows: {bool((ref == got).all())}', flush=True)
# protein blocks: 4 fingerprints per protein
ia = torch.arange(0, nq + 1, 4, device=dev, dtype=torch.int64); ib = torch.arange(0, nd + 1, 4, device=dev, dtype=torch.int64)
npa, npb = len(ia) - 1, len(ib) - 1
mn = torch.empty((npa, npb), dtype=torch.int32, device=dev); last = torch.empty_like(mn)
def bm():
    _lib.check(ctx._lib.dctfp_block_min(ctx.handle, dist.data_ptr(), dist.stride(0), ia.data_ptr(), npa, ib.data_ptr(), npb, mn.data_ptr(),
                                        last.data_ptr(), C.c_void_p(stream.cuda_stream)))
t = timed(bm)
ref = dist[:npa * 4, :npb * 4].reshape(npa, 4, npb, 4).amin(dim=(1, 3))
print(f'block_min {npa} x {npb} blocks of 4 x 4: {1e3 * t:.3f} ms = {nq * nd * 4 / t / 1e9:.0f} GB/s of the matrix; equal to torch: {bool((ref == mn).all())}', flush=True)
# the search as query_db makes it for one tile: matrix, selection, order, one copy to the host -- against the order on the host
from dctdomain_amd.similarity import row_select, order_pairs
t = timed(lambda: row_select(dist, 100), reps=3)
print(f'row_select + row_order + D2H (similarity.row_select, k = 100): {1e3 * t:.2f} ms per tile', flush=True)
val = torch.empty((nq, 100), dtype=torch.int32, device=dev); idx = torch.empty((nq, 100), dtype=torch.int32, device=dev)
_lib.check(ctx._lib.dctfp_row_select(ctx.handle, dist.data_ptr(), nq, nd, dist.stride(0), 100, val.data_ptr(), idx.data_ptr(), C.c_void_p(stream.cuda_stream)))
torch.cuda.synchronize()
hv, hi = val.cpu().numpy().astype(np.int64), idx.cpu().numpy().astype(np.int64)
t0 = time.perf_counter(); order = np.lexsort((hi, hv), axis=1); a = np.take_along_axis(hv, order, axis=1); b = np.take_along_axis(hi, order, axis=1); t_lex = time.perf_counter() - t0
t0 = time.perf_counter(); a2, b2 = order_pairs(hv, hi); t_pack = time.perf_counter() - t0
gv, gi = row_select(dist, 100)
print(f'the same order on the host: np.lexsort + take_along_axis {1e3 * t_lex:.1f} ms (round 3), packed 64-bit keys {1e3 * t_pack:.1f} ms; '
      f'all three equal: {bool((a == a2).all() and (b == b2).all() and (a == gv).all() and (b == gi).all())}', flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); row_select(dist, 100); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(10)
