"""RCCL sanity on one GPU: a world-size-1 `nccl` process group doing the control-plane calls bench.py uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
import torch, torch.distributed as dist
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from dctdomain_amd import dist as dd
dd.barrier(dev)
print('max', dd.max_over_ranks(1.5, dev), 'sum', dd.sum_over_ranks(2.0, dev))
# the per-rank diagnostics block of the N > 1 bench line goes through all_gather_object (pickled into device tensors under nccl)
print('gather_objects', dd.gather_objects(dict(rank=0, **dd.gpu_numa(0), sclk_mhz=[2000, 1990])))
dist.barrier()
dist.destroy_process_group()
print('rccl ok')
