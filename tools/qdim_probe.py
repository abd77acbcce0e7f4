"""The general path (kept sizes other than 3 x 65..80, float64 storage) at the headline shape: which kernels run and
how fast.  usage: python tools/qdim_probe.py [n_seq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
import bench

dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
n_seq, L, D = int(sys.argv[1]) if len(sys.argv) > 1 else 6000, 500, 1280
gen = torch.Generator(device=dev); gen.manual_seed(99)
layers = [bench.make_layer(torch, gen, n_seq * L, D, dev) for _ in range(2)]
offs = np.arange(n_seq, dtype=np.int64) * L
table = dd.PieceTable.whole_sequences([L] * n_seq)
for name, (n, m), store in (('reference [3,80]', (3, 80), 'float32'), ('PROST [5,44]', (5, 44), 'float32'), ('PROST [3,85]', (3, 85), 'float32'),
                             ('[3,64]', (3, 64), 'float32'), ('[4,80]', (4, 80), 'float32'), ('[3,80] float64 rows', (3, 80), 'float64'),
                             ('[8,128]', (8, 128), 'float32')):
    xs = layers if store == 'float32' else [layers[0].double()]
    lbs = [dd.LayerBatch(x, n, m, row_offsets=offs) for x in xs]
    nbytes = len(xs) * n_seq * L * D * xs[0].element_size()
    out = torch.empty((n_seq, n * m * len(xs)), dtype=torch.int8, device=dev)
    for _ in range(2):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    ctx.set_option('profile', 1); ctx.profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    ms, nl = ctx.profile(); ctx.set_option('profile', 0)
    print(f'{name:22s} path {ctx.get_option("last_path")}  step {1e3 * dt:7.3f} ms = {nbytes / dt / 1e9:5.0f} GB/s   '
          f'stage A {ms[0] / 5:7.3f} ms ({nl[0] // 5} launches)  stage B {ms[1] / 5:7.3f} ms ({nl[1] // 5} launches)', flush=True)
    del lbs, out
