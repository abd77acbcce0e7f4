"""Where the waves of the walk kernel spend their cycles, per phase, on the bench workloads.
Needs an instrumented build of the library (never the shipped one):

    tools/build_variant.sh timeline -DDCTFP_WALK_TIMELINE -DDCTFP_EXPERIMENTS
    DCTFP_LIBRARY=build_variants/timeline.so python tools/walk_timeline.py c2 c4 c5 [name=value,...]

Every wave adds the time between its phase marks (s_memrealtime, 10 ns ticks) to device counters (kernels.hip.h,
DCTFP_TL_MARK); the table is the share of the summed wave lifetimes, plus microseconds per job / per flush, and how much of
the launch the wave slots of the chip were occupied (sum of lifetimes / (step x resident waves))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._experiments  # noqa: F401  (engineering knobs: libdctfp_experiments.so unless DCTFP_LIBRARY says otherwise)
import numpy as np, torch
import dctdomain_amd as dd
import bench

PHASES = ['stream (job record -> last row)', 'epilogue (scale + pack)', 'flush: unpack + MFMA', 'wait: last arrivals for the others',
          'ticket, cross-wave sum + int8', 'wait: slots of the last flush group', 'other']
dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
workloads = [a for a in sys.argv[1:] if '=' not in a] or ['c2', 'c4', 'c5']
cfgs = [a for a in sys.argv[1:] if '=' in a] or ['path=2']
nseq = {'c2': 10000, 'c3': 10000, 'c4': 12000, 'c5': 40000}
steps = 5
for w in workloads:
    argv, sys.argv = sys.argv, ['bench.py', '--workload', w, '--n-seq', str(nseq[w])]
    a = bench.parse()
    sys.argv = argv
    lengths, doms, D = bench.make_workload(a, 0, np)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    layers = [bench.make_layer(torch, gen, int(lengths.sum()), D, dev) for _ in range(2)]
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    table = dd.PieceTable.whole_sequences(lengths) if doms is None else dd.PieceTable(lengths, doms)
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    nbytes = 2 * int(lengths.sum()) * D * 4
    out = torch.empty((table.n_domains, 480), dtype=torch.int8, device=dev)
    for cfg in cfgs:
        saved = {}
        for kv in cfg.split(','):
            k, v = kv.split('=')
            saved[k] = ctx.get_option(k)
            ctx.set_option(k, int(v))
        for _ in range(2):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        ctx.set_option('degenerate_channels', 0)          # resets every counter
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        c = [ctx.get_option(f'walk_timeline_{i}') for i in range(11)]
        life, waves, jobs, flushes = c[7], c[8], c[9], c[10]
        tick = 1e-8   # s_memrealtime: 100 MHz
        print(f'\n{w}  {cfg}  D={D}  {table.n_domains} fingerprints  step {1e3 * dt:.3f} ms = {nbytes / dt / 1e9:.0f} GB/s   '
              f'{waves // steps} waves, {jobs // steps} wave-jobs, {flushes // steps} wave-flushes per step; '
              f'mean wave lifetime {1e6 * tick * life / max(1, waves):.1f} us; sum of lifetimes / step = {tick * life / steps / dt:.0f} waves resident on average')
        for i, name in enumerate(PHASES):
            per = 1e6 * tick * c[i] / max(1, flushes if i in (2, 3, 4, 5) else jobs)
            print(f'  {name:34s} {100.0 * c[i] / max(1, life):5.1f} %   {per:9.2f} us per {"flush" if i in (2, 3, 4, 5) else "job"}')
        for k, v in saved.items():
            ctx.set_option(k, v)
    del layers, lbs, out
    torch.cuda.empty_cache()
