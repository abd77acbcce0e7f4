"""Time of contact_topk1_kernel cut after its phases (builds with -DDCTFP_TOPK1_STOP=0/1/2: results are wrong, the time is the point).
usage: DCTFP_LIBRARY=build_variants/X.so python tools/topk_phase_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import _lib
n, L = 4096, 500
dev = torch.device('cuda', 0)
ctx = _lib.get_context(0)
lib = ctx._lib
k = int(lib.dctfp_contact_count(L, 2.6))
gen = torch.Generator(device=dev); gen.manual_seed(5)
big = torch.rand((n, L, L), device=dev, generator=gen)
ptrs = (np.uint64(big.data_ptr()) + np.arange(n, dtype=np.uint64) * np.uint64(L * L * 4))
lds = np.full(n, L, dtype=np.int64); n_res = np.full(n, L, dtype=np.int32); offs = (np.arange(n + 1, dtype=np.int64) * k)
oi = torch.zeros(n * k, dtype=torch.int32, device=dev); oj = torch.zeros_like(oi); ov = torch.zeros(n * k, dtype=torch.float32, device=dev)
on = torch.zeros(n, dtype=torch.int32, device=dev)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def call():
    _lib.check(lib.dctfp_contact_topk(ctx.handle, ptrs.ctypes.data, lds.ctypes.data, n_res.ctypes.data, n, 2.6, oi.data_ptr(), oj.data_ptr(), ov.data_ptr(), offs.ctypes.data, on.data_ptr(), sp), lib)
call(); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5): call()
ev[1].record(); torch.cuda.synchronize()
print(os.environ.get('DCTFP_LIBRARY', 'default'), f'{ev[0].elapsed_time(ev[1]) / 5:.3f} ms per call (all launches)', 'out_n[:4] =', on[:4].tolist(), flush=True)
