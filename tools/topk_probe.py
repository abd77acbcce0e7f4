"""Contact top-k of one long protein (the multi-workgroup selection) and of a batch of short ones: time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import numpy as np, torch
from dctdomain_amd import reccut
from recipes_contacts import make_contacts
for L in (500, 1400, 1500, 2000, 5000):
    m = torch.from_numpy(make_contacts('blocks', L, 5, nb=max(1, L // 120))).cuda()
    for _ in range(3):
        reccut.top_contacts_batch([m], 2.6, sort=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        reccut.top_contacts_batch([m], 2.6, sort=False)
    torch.cuda.synchronize()
    print(f'L = {L:5d}: {1e3 * (time.perf_counter() - t0) / 20:7.3f} ms per protein (top {int(2.6 * L)} of {(L - 5) * (L - 4) // 2} pairs, D2H included)')
maps = [torch.from_numpy(make_contacts('blocks', 500, 100 + i, nb=4)).cuda() for i in range(256)]
for _ in range(2):
    reccut.top_contacts_batch(maps, 2.6, sort=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    reccut.top_contacts_batch(maps, 2.6, sort=False)
torch.cuda.synchronize()
print(f'batch of 256 x L = 500: {1e6 * (time.perf_counter() - t0) / 5 / 256:7.2f} us per protein')
