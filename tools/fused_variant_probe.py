import os, sys, time
sys.path.insert(0, '/root/repo')
import tools._experiments  # noqa: F401  (engineering knobs: libdctfp_experiments.so unless DCTFP_LIBRARY says otherwise)
import numpy as np, torch
import dctdomain_amd as dd
dev = torch.device('cuda', 0)
n_seq, L, D = 8000, 500, int(sys.argv[1]) if len(sys.argv) > 1 else 1280
layers = [torch.randn((n_seq * L, D), device=dev) for _ in range(2)]
offs = np.arange(n_seq, dtype=np.int64) * L
lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
ctx = dd.get_context(0)
ctx.set_option('path', 2)
whole = [[f'1-{L}']] * n_seq
one_fused = [[f'1-250', f'251-{L}', f'1-{L}']] + [[f'1-{L}']] * (n_seq - 1)
parts5 = [[f'{100*i+1}-{100*i+100}' for i in range(5)]] * n_seq
parts5_onefused = [[f'1-250', f'251-{L}', f'1-{L}']] + [[f'{100*i+1}-{100*i+100}' for i in range(5)]] * (n_seq - 1)
nbytes = 2 * n_seq * L * D * 4
for name, doms, opts in (('whole only, plain kernel (U8 G4)', whole, {}), ('whole only, plain kernel U4 G3', whole, {'ab_unroll': 4, 'ab_group': 3}),
                         ('whole only but FUSED kernel variant (one fused protein)', one_fused, {}),
                         ('5 x 100-row parts, plain kernel', parts5, {}), ('5 x 100-row parts, FUSED kernel variant', parts5_onefused, {})):
    for k in ('ab_unroll', 'ab_group'):
        ctx.set_option(k, 0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    table = dd.PieceTable([L] * n_seq, doms)
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    for _ in range(3):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f'D={D} {name:58s}: step {1e3 * dt:7.3f} ms = {nbytes / dt / 1e9:6.0f} GB/s', flush=True)
