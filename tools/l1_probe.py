"""Rate of dctfp_l1_matrix (all-against-all L1 distances of int8 fingerprints, 480 bytes each): T byte-differences per second
against the v_sad_u8 peak (4 differences per lane and instruction: 256 CUs x 64 lanes per clock x 4 = 65 536 per clock,
157 T/s at 2.4 GHz), and the int32 matrix written per second.   usage: python tools/l1_probe.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctdomain_amd.similarity import l1_matrix
dev = torch.device('cuda', 0)
for n in [int(v) for v in sys.argv[1:]] or [8192, 20000, 40000]:
    a = torch.randint(0, 128, (n, 480), dtype=torch.int8, device=dev)
    b = torch.randint(0, 128, (n, 480), dtype=torch.int8, device=dev)
    out = l1_matrix(a, b); torch.cuda.synchronize()
    ref = (a[:64].to(torch.int32)[:, None, :] - b[None, :256].to(torch.int32)).abs().sum(-1)
    assert (out[:64, :256] == ref).all()
    t0 = time.perf_counter()
    for _ in range(5):
        out = l1_matrix(a, b)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    print(f'{n} x {n} x 480: {1e3 * t:.2f} ms = {n * n * 480 / t / 1e12:.1f} T differences/s ({100 * n * n * 480 / t / 157.3e12:.0f} % of the v_sad_u8 peak), '
          f'{4 * n * n / t / 1e9:.0f} GB/s of int32 written', flush=True)
    del out
