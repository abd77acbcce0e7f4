import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import reccut
dev = torch.device('cuda', 0)
for L, n in ((200, 2048), (500, 2048), (1000, 512), (2000, 128), (5000, 8)):
    maps = [torch.rand((L, L), device=dev) for _ in range(min(n, 16))]
    mm = [maps[i % len(maps)] for i in range(n)]
    reccut.top_contacts_batch(mm, 2.6, sort=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reccut.top_contacts_batch(mm, 2.6, sort=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'L={L:5d} x {n:5d} proteins: {1e3 * dt:8.2f} ms total, {1e6 * dt / n:8.1f} us per protein, map bytes read once = {n * L * L * 4 / dt / 1e9:7.1f} GB/s')
