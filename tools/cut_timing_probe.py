"""Where the GPU cutter's time goes: an instrumented build (-DDCTFP_CUT_TIMING -DDCTFP_WALK_TIMELINE) adds thread 0's time per phase
(100 MHz ticks, summed over all proteins of the launch) to the context's spare counters.
usage: DCTFP_LIBRARY=build_variants/cut_timing.so python tools/cut_timing_probe.py [n] [Lmin] [Lmax]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import _lib, make_db, reccut
from dctdomain_amd.embedding import Batch, SyntheticModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 81
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 1330
dev = torch.device('cuda', 0)
rng = np.random.default_rng(7)
lens = np.clip(rng.gamma(2.2, 170.0, size=n).astype(int), lo, hi)
aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
model = SyntheticModel(); model.to_device(dev)
maps = []
for i, L in enumerate(lens):
    bt = Batch([(f'sp{i}', aa[rng.integers(0, 20, size=L)].tobytes().decode())], model, dev)
    bt.embed_batch(make_db.LAYERS, 500)
    maps.append(bt.embeds[0].contacts)
ctx = _lib.get_context(0)
reccut.domains_from_maps(maps, 2.6)
ctx.set_option('degenerate_channels', 0)
t0 = time.perf_counter()
doms = reccut.domains_from_maps(maps, 2.6)
dt = time.perf_counter() - t0
v = [ctx.get_option(f'walk_timeline_{i}') for i in range(11)]
names = ['graph', 'pre/post + forward lists', 'scans + single cut', 'scan: tile fills + final reduce', 'decision', 'hand-over', None, None,
         "scan: wave 0's rows", "scan: wave 0's wait for the block's slowest wave", 'scan: tile clears']
tot = sum(v[:6]) + sum(v[8:11])
print(f'{n} proteins L in [{lens.min()}, {lens.max()}] mean {lens.mean():.0f}: call {1e3 * dt:.2f} ms; {v[6]} nodes, sum V^2 = {v[7]}')
for k, nm in enumerate(names):
    if nm is None:
        continue
    print(f'  {nm:28s} {v[k] / 100:10.0f} us summed over proteins = {100 * v[k] / tot:5.1f} %  ({v[k] / 100 / max(1, v[6]):.2f} us per node)')
