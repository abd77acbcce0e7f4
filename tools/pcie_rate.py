"""PCIe-inclusive rate: the reference's own usage, Fingerprint(embed=numpy arrays).quantize(), where
every call copies the embeddings host -> device first.  Never the bench value; noted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
rng = np.random.default_rng(0)
if os.environ.get('DCTFP_SMALL_ONE') == '1':          # A/B: small_call_kernel instead of the three kernels of a small call
    dd.get_context(0).set_option('small_one', 1)
    print('small_one = 1: small_call_kernel (one launch)')
L, D = 500, 1280
embeds = [{15: rng.standard_normal((L, D)).astype(np.float32), 21: rng.standard_normal((L, D)).astype(np.float32)} for _ in range(16)]
def one(e):
    fp = dd.Fingerprint(pid='x', seq='A' * L, embed=e, domains=[f'1-{L}'])
    fp.quantize([3, 80, 3, 80])
    return fp
for e in embeds[:4]:
    one(e)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 3.0:
    one(embeds[n % len(embeds)])
    n += 1
dt = time.perf_counter() - t0
print(f'numpy in / numpy out, one protein per call: {n / dt:.0f} fingerprints/s ({1e3 * dt / n:.3f} ms per call, '
      f'{n * 2 * L * D * 4 / dt / 1e9:.2f} GB/s of host data)')
dev = [{k: torch.from_numpy(v).cuda() for k, v in e.items()} for e in embeds]
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3.0:
    one(dev[n % len(dev)])
    n += 1
dt = time.perf_counter() - t0
print(f'GPU tensors in, one protein per call: {n / dt:.0f} fingerprints/s ({1e3 * dt / n:.3f} ms per call)')
if len(sys.argv) > 1 and sys.argv[1] == 'profile':
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(2000):
        one(dev[i % len(dev)])
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
