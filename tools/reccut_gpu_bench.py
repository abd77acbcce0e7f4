"""dctfp_contact_topk + dctfp_reccut on a flush's worth of contact maps (pfam-like lengths, the synthetic model's maps): time per
call and per protein; run under rocprofv3 --kernel-trace --stats for the kernels' own durations.
usage: python tools/reccut_gpu_bench.py [n_proteins]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import make_db, reccut
from dctdomain_amd.embedding import Batch, SyntheticModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device('cuda', 0)
rng = np.random.default_rng(7)
lens = np.clip(rng.gamma(2.2, 170.0, size=n).astype(int), 81, 1330)
aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
model = SyntheticModel()
model.to_device(dev)
maps = []
for i, L in enumerate(lens):
    bt = Batch([(f'sp{i}', aa[rng.integers(0, 20, size=L)].tobytes().decode())], model, dev)
    bt.embed_batch(make_db.LAYERS, 500)
    maps.append(bt.embeds[0].contacts)
torch.cuda.synchronize()
for _ in range(2):
    doms = reccut.domains_from_maps(maps, 2.6)
t0 = time.perf_counter()
for _ in range(5):
    doms = reccut.domains_from_maps(maps, 2.6)
dt = (time.perf_counter() - t0) / 5
print(f'{n} proteins (mean L {lens.mean():.0f}): domains_from_maps {1e3 * dt:.2f} ms = {1e6 * dt / n:.2f} us per protein; '
      f'{sum(len(d) for d in doms)} domains, host redo {len(reccut.LAST.host_redo)}', flush=True)
offs, ci, cj, cv = reccut.top_contacts_batch(maps, 2.6, sort=False)
t0 = time.perf_counter()
ref = reccut.domains_from_contacts(lens, offs, ci, cj, cv, threads=16)
print(f'host library on 16 threads: {1e3 * (time.perf_counter() - t0):.2f} ms; same strings: {ref == doms}', flush=True)
