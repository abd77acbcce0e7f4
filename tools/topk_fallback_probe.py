"""Which contact maps of a flush fall through the one-read kernel (and the two-read kernel behind it): dctfp_contact_topk timed
per length bucket."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import make_db, reccut
from dctdomain_amd.embedding import Batch, SyntheticModel
n = 1024
dev = torch.device('cuda', 0)
rng = np.random.default_rng(7)
lens = np.clip(rng.gamma(2.2, 170.0, size=n).astype(int), 81, 1330)
aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
model = SyntheticModel(); model.to_device(dev)
maps = []
for i, L in enumerate(lens):
    bt = Batch([(f'sp{i}', aa[rng.integers(0, 20, size=L)].tobytes().decode())], model, dev)
    bt.embed_batch(make_db.LAYERS, 500)
    maps.append(bt.embeds[0].contacts)
edges = [81, 200, 350, 500, 501, 540, 700, 900, 1154, 1331]
for a, b in zip(edges[:-1], edges[1:]):
    sel = [m for m, L in zip(maps, lens) if a <= L < b]
    if not sel:
        continue
    reccut._select_on_device(sel, 2.6); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        reccut._select_on_device(sel, 2.6)
    torch.cuda.synchronize()
    print(f'L in [{a}, {b}): {len(sel):4d} maps, {1e3 * (time.perf_counter() - t0) / 3:7.3f} ms per call', flush=True)
m = maps[int(np.argmax((lens > 500) & (lens < 560)))] if ((lens > 500) & (lens < 560)).any() else None
if m is not None:
    L = m.shape[0]
    v = m[torch.triu(torch.ones_like(m, dtype=torch.bool), 5)]
    vs, cnt = torch.unique(v, return_counts=True)
    k = int(2.6 * L)
    top = torch.sort(v, descending=True).values[:k]
    print(f'a map of L = {L}: {v.numel()} candidates, {vs.numel()} distinct values; the k-th value {float(top[-1]):.6f} occurs {int((v == top[-1]).sum())} times; '
          f'largest tie class {int(cnt.max())} (value {float(vs[cnt.argmax()]):.6f})')
