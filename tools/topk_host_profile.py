"""reccut.top_contacts_batch on 4096 contact maps of L = 500: the call as a flush makes it (sort=False, own=False) and with
copies and the reference's order, on maps without structure (uniform random) and on banded ones (contacts along the diagonal,
like a real map: the second read of contact_topk2_kernel skips the chunks that cannot hold a candidate); cProfile of one call.
Under rocprofv3 the kernel stats give the kernel's share."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctdomain_amd import reccut
dev = torch.device('cuda', 0)
i, j = torch.meshgrid(torch.arange(500, device=dev), torch.arange(500, device=dev), indexing='ij')
kinds = {'uniform random': lambda: torch.rand((500, 500), device=dev),
         'banded': lambda: (torch.exp(-(i - j).abs() / 6.0) * (0.6 + 0.4 * torch.rand((500, 500), device=dev))).contiguous()}
for name, make in kinds.items():
    maps = [make() for _ in range(64)]
    mm = [maps[q % 64] for q in range(4096)]
    for kw in (dict(sort=False, own=False), dict()):
        for _ in range(2):
            reccut.top_contacts_batch(mm, 2.6, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            reccut.top_contacts_batch(mm, 2.6, **kw)
        torch.cuda.synchronize()
        print(f'{name:15s} {str(kw):32s} {(time.perf_counter() - t0) / 3 * 1e3:7.2f} ms per call of 4096 proteins', flush=True)
pr = cProfile.Profile()
pr.enable()
reccut.top_contacts_batch(mm, 2.6, sort=False, own=False)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
