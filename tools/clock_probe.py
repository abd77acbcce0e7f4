"""Is the chip at its clock while the walk kernel runs?  Loops one workload for a few seconds and samples
`rocm-smi --showclocks --showpower` from a side thread.  usage: python tools/clock_probe.py c2 c4 c5 [name=value ...]
(options need the experiments library: DCTFP_LIBRARY=dctdomain_amd/libdctfp_experiments.so)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
import bench

dev = torch.device('cuda', 0)
ctx = dd.get_context(0)
for kv in [a for a in sys.argv[1:] if '=' in a]:
    ctx.set_option(kv.split('=')[0], int(kv.split('=')[1]))
sys.argv = [a for a in sys.argv if '=' not in a]
nseq = {'c2': 10000, 'c3': 10000, 'c4': 12000, 'c5': 40000}


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5)
            out.append(r.stdout.strip()[:2000])
        except Exception as e:     # noqa: BLE001
            out.append(f'error {e}')
        time.sleep(0.7)


for w in sys.argv[1:] or ['c2', 'c4', 'c5']:
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
    for _ in range(3):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 6.0:
        for _ in range(20):
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
        torch.cuda.synchronize(); n += 20
    dt = (time.perf_counter() - t0) / n
    stop.set(); th.join()
    print(f'== {w}: {1e3 * dt:.3f} ms per step = {nbytes / dt / 1e9:.0f} GB/s over {n} steps')
    import json
    for s in samples[1:6]:
        try:
            d = json.loads(s)['card0']
            print('    sclk', d.get('sclk clock speed:'), ' power', d.get('Current Socket Graphics Package Power (W)'), 'W')
        except Exception:     # noqa: BLE001
            print('   ', ' '.join(s.split())[:300])
    del layers, lbs, out
    torch.cuda.empty_cache()
