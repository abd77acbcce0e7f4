import ctypes, os, sys
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/tools') else os.getcwd())
import tools._experiments  # noqa
import numpy as np, torch
import dctdomain_amd as dd
import bench
dev = torch.device('cuda', 0); ctx = dd.get_context(0)
for w, n in (('c4', 12000), ('c5', 40000)):
    sys.argv = ['bench.py', '--workload', w, '--n-seq', str(n)]
    a = bench.parse(); lengths, doms, D = bench.make_workload(a, 0, np)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    layers = [bench.make_layer(torch, gen, int(lengths.sum()), D, dev) for _ in range(2)]
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    table = dd.PieceTable(lengths, doms)
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    for _ in range(4): dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    torch.cuda.synchronize()
    cap = 1 << 21
    ctx.set_option('walk_trace', cap)
    dd.quantize_batch(lbs, table, out=out, ctx=ctx); torch.cuda.synchronize()
    addr = ctx.get_option('walk_trace_host')
    tr = np.ctypeslib.as_array((ctypes.c_uint64 * (4 * cap)).from_address(addr)).reshape(cap, 4).copy()
    ctx.set_option('walk_trace', 0)
    tr = tr[tr[:, 1] > 0]
    wave = tr[:, 3] & 0xff; wait = (tr[:, 3] >> 32) * 0.01; life = (tr[:, 1] - tr[:, 0]) * 0.01
    hw = tr[:, 2] & 0xffffffff; simd = (hw >> 4) & 3
    print(f'{w}: mean barrier wait per wave lifetime, by wave index (us; share of lifetime)')
    for k in range(int(wave.max()) + 1):
        m = wave == k
        print(f'   wave {k}: wait {wait[m].mean():7.1f} us = {100 * wait[m].sum() / life[m].sum():5.1f} %   (simd of that wave: {np.bincount(simd[m].astype(int), minlength=4)})')
    print('   by SIMD:', [f'{100 * wait[simd == q].sum() / life[simd == q].sum():.1f} %' for q in range(4)])
    del layers, lbs, out; torch.cuda.empty_cache()
