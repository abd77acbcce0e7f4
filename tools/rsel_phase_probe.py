"""row_select_reg_kernel cut after its phases (-DDCTFP_RSEL_STOP=1/2/3: results wrong, the time is the point)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import _lib
nq, nd, k = 6700, 40000, 100
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(7)
dist = torch.randint(2000, 60000, (nq, nd), device=dev, generator=g, dtype=torch.int32)
ctx = _lib.get_context(0)
val = torch.empty((nq, k), dtype=torch.int32, device=dev); idx = torch.empty_like(val)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def sel():
    _lib.check(ctx._lib.dctfp_row_select(ctx.handle, dist.data_ptr(), nq, nd, dist.stride(0), k, val.data_ptr(), idx.data_ptr(), sp))
sel(); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10): sel()
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 10
print(os.environ.get('DCTFP_LIBRARY', 'default').split('/')[-1], f'{ms:.3f} ms = {nq * nd * 4 / ms / 1e6:.0f} GB/s', flush=True)
