"""Kernel-level timings of the "next" rows: stitcher, contact top-k, L1 matrix."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
from dctdomain_amd import reccut
from dctdomain_amd.embedding import stitch_embeddings_batch, stitch_contacts_batch
from dctdomain_amd.similarity import l1_matrix
dev = torch.device('cuda', 0)
res = {}
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

# f-1: 2000 sequences of L = 1400 (windows 500,500,500,500), D = 1280
n, D = 2000, 1280
wins = [[torch.randn((500, D), device=dev) for _ in range(4)] for _ in range(64)]
seqw = [wins[i % 64] for i in range(n)]
t = timeit(lambda: stitch_embeddings_batch(seqw))
bytes_moved = n * (4 * 500 * D * 4 + 1400 * D * 4)       # every window row read once, every output row written once (one launch)
res['stitch_embeddings'] = {'sequences': n, 'ms': round(1e3 * t, 3), 'GBps': round(bytes_moved / t / 1e9),
                            'note': 'whole call: window checks in Python, geometry in C, one allocation, ONE launch (round 3: four, 41 GB with the overlap re-reads, 9.0 ms)'}
cw = [[torch.rand((500, 500), device=dev) for _ in range(4)] for _ in range(64)]
t = timeit(lambda: stitch_contacts_batch([cw[i % 64] for i in range(256)], 300))
res['stitch_contacts'] = {'sequences': 256, 'ms': round(1e3 * t, 3), 'us_per_sequence': round(1e6 * t / 256, 1)}

# f-2: top-k over 4096 contact maps of L = 500
maps = [torch.rand((500, 500), device=dev) for _ in range(64)]
mm = [maps[i % 64] for i in range(4096)]
t = timeit(lambda: reccut.top_contacts_batch(mm, 2.6), reps=2)
t_views = timeit(lambda: reccut.top_contacts_batch(mm, 2.6, sort=False, own=False), reps=2)     # what a database flush calls
res['contact_topk'] = {'proteins': 4096, 'L': 500, 'ms_total': round(1e3 * t, 2), 'us_per_protein': round(1e6 * t / 4096, 1),
                       'ms_total_as_the_flush_calls_it': round(1e3 * t_views, 2), 'us_per_protein_as_the_flush_calls_it': round(1e6 * t_views / 4096, 1),
                       'note': 'selection kernel + ordering kernel + one D2H through pinned buffers; reference writece: 107 ms per protein'}

# f-4: 40k x 40k int8 fingerprints
a = torch.randint(0, 128, (40000, 480), dtype=torch.int8, device=dev)
t = timeit(lambda: l1_matrix(a, a), reps=3)
res['l1_matrix'] = {'shape': '40000 x 40000 x 480', 'ms': round(1e3 * t, 2), 'T_abs_diffs_per_s': round(40000 * 40000 * 480 / t / 1e12, 2)}
# f-4: what a search tile pays after the matrix: the 100 nearest per row of a 6 700 x 40 000 tile (1 GiB, query_db.TILE_INTS),
# and the protein x protein minima of dct-sim over the same tile
from dctdomain_amd import _lib
import ctypes as C
dist = l1_matrix(a[:6700], a)
ctx = _lib.get_context(0)
stream = torch.cuda.current_stream(dev)
val = torch.empty((6700, 100), dtype=torch.int32, device=dev); idx = torch.empty_like(val)
t = timeit(lambda: _lib.check(ctx._lib.dctfp_row_select(ctx.handle, dist.data_ptr(), 6700, 40000, dist.stride(0), 100, val.data_ptr(), idx.data_ptr(),
                                                        C.c_void_p(stream.cuda_stream))), reps=5)
res['row_select'] = {'shape': '6700 x 40000, k = 100', 'ms': round(1e3 * t, 3), 'GBps_of_the_matrix': round(6700 * 40000 * 4 / t / 1e9)}
ia = torch.arange(0, 6701, 4, device=dev, dtype=torch.int64); ib = torch.arange(0, 40001, 4, device=dev, dtype=torch.int64)
mn = torch.empty((len(ia) - 1, len(ib) - 1), dtype=torch.int32, device=dev); last = torch.empty_like(mn)
t = timeit(lambda: _lib.check(ctx._lib.dctfp_block_min(ctx.handle, dist.data_ptr(), dist.stride(0), ia.data_ptr(), len(ia) - 1, ib.data_ptr(), len(ib) - 1,
                                                       mn.data_ptr(), last.data_ptr(), C.c_void_p(stream.cuda_stream))), reps=5)
res['block_min'] = {'shape': '1675 x 10000 blocks of 4 x 4', 'ms': round(1e3 * t, 3), 'GBps_of_the_matrix': round(6700 * 40000 * 4 / t / 1e9)}
print(json.dumps(res, indent=1))
