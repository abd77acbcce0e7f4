"""When and where did every wave of ONE walk-kernel launch run?  Needs the instrumented build (never the shipped library):

    tools/build_variant.sh timeline -DDCTFP_WALK_TIMELINE -DDCTFP_EXPERIMENTS
    DCTFP_LIBRARY=build_variants/timeline.so python tools/walk_trace.py c4 c5 c2 [name=value,...]

Every wave stores {begin, end (s_memrealtime, 10 ns), HW_ID / XCC_ID, workgroup} (kernels.hip.h, option walk_trace).  Printed:
how many waves are resident over the launch, how the compute units are filled (waves per CU over time), the gap between a
wave's end and the start of the next wave in the same wave slot, and the tail of the launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._experiments  # noqa: F401
import numpy as np, torch
import dctdomain_amd as dd
import bench

dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
workloads = [a for a in sys.argv[1:] if '=' not in a] or ['c4', 'c5', 'c2']
cfgs = [a for a in sys.argv[1:] if '=' in a] or ['path=2']
nseq = {'c2': 10000, 'c3': 10000, 'c4': 12000, 'c5': 40000}
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
        for _ in range(5):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize()
        cap = 1 << 21
        ctx.set_option('walk_trace', cap)
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize()
        addr = ctx.get_option('walk_trace_host')
        tr = np.ctypeslib.as_array((ctypes.c_uint64 * (4 * cap)).from_address(addr)).reshape(cap, 4).copy()
        ctx.set_option('walk_trace', 0)
        tr = tr[tr[:, 1] > 0]
        t0 = tr[:, 0].min()
        b, e = (tr[:, 0] - t0) * 0.01, (tr[:, 1] - t0) * 0.01          # us
        hw, xcc = tr[:, 2] & 0xffffffff, tr[:, 2] >> 32
        # gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        cu = ((xcc & 0xf) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
        slot = (cu << 6) | (((hw >> 4) & 3) << 4) | (hw & 0xf)                # (cu, simd, wave slot)
        total = e.max()
        print(f'\n{w}  {cfg}  D={D}: {len(tr)} waves on {len(np.unique(cu))} CUs, launch {total:.0f} us '
              f'(algorithmic {nbytes / total / 1e3:.0f} GB/s over the traced launch); wave lifetime mean {np.mean(e - b):.0f} us, '
              f'median {np.median(e - b):.0f}, max {np.max(e - b):.0f}')
        # resident waves over time
        ev = np.concatenate([np.stack([b, np.ones_like(b)], 1), np.stack([e, -np.ones_like(e)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind='stable')]
        res = np.cumsum(ev[:, 1])
        dt = np.diff(ev[:, 0], append=ev[-1, 0])
        mean_res = float((res * dt).sum() / total)
        peak = int(res.max())
        print(f'  resident waves: mean {mean_res:.0f}, peak {peak} ({peak / len(np.unique(cu)):.1f} per CU); '
              f'mean over the middle half of the launch {float((res * dt)[(ev[:, 0] > total / 4) & (ev[:, 0] < 3 * total / 4)].sum() / (total / 2)):.0f}')
        for lo, hi in ((0, 0.05), (0.05, 0.5), (0.5, 0.9), (0.9, 0.95), (0.95, 1.0)):
            m = (ev[:, 0] >= lo * total) & (ev[:, 0] < hi * total)
            print(f'    {100 * lo:3.0f}-{100 * hi:3.0f} % of the launch: {float((res * dt)[m].sum() / ((hi - lo) * total)):6.0f} waves resident')
        # gap between the end of a wave and the begin of the next one in the same hardware wave slot
        order = np.lexsort((b, slot))
        same = slot[order][1:] == slot[order][:-1]
        gap = (b[order][1:] - e[order][:-1])[same]
        print(f'  same wave slot, end -> next begin: median {np.median(gap):.1f} us, mean {np.mean(gap):.1f}, 90 % {np.percentile(gap, 90):.1f}, '
              f'max {gap.max():.0f}; {len(gap)} hand-overs = {gap.sum() / (mean_res * total) * 100:.1f} % of the resident wave time')
        # per CU: time with no wave at all
        idle = []
        for c in np.unique(cu):
            m = cu == c
            evc = np.concatenate([np.stack([b[m], np.ones(m.sum())], 1), np.stack([e[m], -np.ones(m.sum())], 1)])
            evc = evc[np.argsort(evc[:, 0], kind='stable')]
            rc = np.cumsum(evc[:, 1])
            dtc = np.diff(evc[:, 0], append=evc[-1, 0])
            inside = (evc[:, 0] >= evc[0, 0])
            idle.append((float(dtc[(rc == 0) & inside].sum()), float(evc[0, 0]), float(total - evc[-1, 0]), float((rc * dtc).sum() / total)))
        idle = np.array(idle)
        print(f'  per CU: empty between its first and last wave {idle[:, 0].mean():.0f} us mean ({100 * idle[:, 0].mean() / total:.1f} % of the launch), '
              f'first wave at {idle[:, 1].mean():.0f} us mean / {idle[:, 1].max():.0f} max, last wave ends {idle[:, 2].mean():.0f} us before the launch does '
              f'(max {idle[:, 2].max():.0f}); waves per CU mean {idle[:, 3].mean():.1f}, min {idle[:, 3].min():.1f}, max {idle[:, 3].max():.1f}')
        for k, v in saved.items():
            ctx.set_option(k, v)
    del layers, lbs, out
    torch.cuda.empty_cache()
