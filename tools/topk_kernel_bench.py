"""The contact top-k kernels on 4 096 DISTINCT maps of L = 500 (4.1 GB: nothing is served from a cache): the one-read kernel
of round 5 (option topk_kernel = 0), the two-read kernel of round 4 (2) and the radix select (1), on maps without structure and
on banded ones; results compared entry for entry as sets per protein (the order inside a protein is unspecified).  Run under
`rocprofv3 --kernel-trace --stats` for the kernels' own durations; the loop time printed here includes the launch of the
(empty) redo kernels behind the first one.
usage: python tools/topk_kernel_bench.py [n_maps] [L]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dctdomain_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device('cuda', 0)
ctx = _lib.experiments_context(0)
lib = ctx._lib
k = int(lib.dctfp_contact_count(L, 2.6))
tri = (L - 5) * (L - 4) // 2
print(f'{n} maps of L = {L}: k = {k}, candidate triangle {tri} entries = {tri * 4 * n / 1e9:.3f} GB', flush=True)
i, j = torch.meshgrid(torch.arange(L, device=dev), torch.arange(L, device=dev), indexing='ij')
band = torch.exp(-(i - j).abs() / 6.0)
gen = torch.Generator(device=dev)
gen.manual_seed(5)
for name in ('uniform random', 'banded'):
    big = torch.rand((n, L, L), device=dev, generator=gen)
    if name == 'banded':
        big = (band[None] * (0.6 + 0.4 * big)).contiguous()
    ptrs = (np.uint64(big.data_ptr()) + np.arange(n, dtype=np.uint64) * np.uint64(L * L * 4))
    lds = np.full(n, L, dtype=np.int64)
    n_res = np.full(n, L, dtype=np.int32)
    offs = (np.arange(n + 1, dtype=np.int64) * k)
    outs = {}
    for opt in (0, 2, 1):
        ctx.set_option('topk_kernel', opt)
        oi = torch.zeros(n * k, dtype=torch.int32, device=dev)
        oj = torch.zeros(n * k, dtype=torch.int32, device=dev)
        ov = torch.zeros(n * k, dtype=torch.float32, device=dev)
        on = torch.zeros(n, dtype=torch.int32, device=dev)
        sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def call():
            _lib.check(lib.dctfp_contact_topk(ctx.handle, ptrs.ctypes.data, lds.ctypes.data, n_res.ctypes.data, n, 2.6, oi.data_ptr(),
                                              oj.data_ptr(), ov.data_ptr(), offs.ctypes.data, on.data_ptr(), sp), lib)
        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        assert bool((on == k).all())
        key = (oi.long() * 65536 + oj.long()).view(n, k).sort(dim=1).values
        outs[opt] = key
        print(f'{name:15s} topk_kernel = {opt}: {ms:7.3f} ms per call = {tri * 4 * n / ms / 1e6:7.1f} GB/s of the triangle', flush=True)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[1]), 'the kernels disagree'
    print(f'{name:15s} the three kernels select the same {n * k} contacts', flush=True)
    del big
ctx.set_option('topk_kernel', 0)
