"""Where the time of reccut.top_contacts_batch goes for 4096 contact maps of L = 500 (cProfile + a kernel-only timing)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctdomain_amd import reccut
dev = torch.device('cuda', 0)
maps = [torch.rand((500, 500), device=dev) for _ in range(64)]
mm = [maps[i % 64] for i in range(4096)]
for _ in range(2):
    reccut.top_contacts_batch(mm, 2.6)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    reccut.top_contacts_batch(mm, 2.6)
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per call of 4096 proteins')
pr = cProfile.Profile()
pr.enable()
reccut.top_contacts_batch(mm, 2.6)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
