"""One process per GPU, sequences sharded, no collective on the data path.

Mirrors the reference's data parallelism -- one ``mp.Process`` per GPU pulling sequence
batches from a shared queue (mgtools/DCTdomain src/make_db.py:95-117) and a single SQLite
writer -- with ``torch.distributed`` ranks: ``nccl`` (= RCCL) on GPUs, ``gloo`` on CPU for
tests.  The only communication is control-plane: a barrier, a max over ranks of a timing,
and the gather of the 480-byte fingerprints to the writer rank."""

from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def env_rank():
    """(rank, world_size, local_rank) from the torchrun environment (defaults 0, 1, 0)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def init(backend: str = None, device: torch.device = None):
    """Initialises the default process group when WORLD_SIZE > 1; returns (rank, world)."""
    rank, world, _ = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if (device is not None and device.type == 'cuda') else 'gloo'
        kwargs = {}
        if backend == 'nccl' and device is not None:
            kwargs['device_id'] = device
        dist.init_process_group(backend, **kwargs)
    return rank, world


def balanced_shards(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Deterministic length-balanced partition of sequence indices (work is proportional to
    L * D): longest first onto the least loaded rank, ties to the lowest rank; every shard is
    returned in ascending index order."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(lengths[i])
    for s in shards:
        s.sort()
    return shards


def barrier(device: torch.device = None):
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device: torch.device = None) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_root(obj, root: int = 0):
    """Python objects (pids, domain strings, int8 arrays) to the writer rank; None elsewhere."""
    if not dist.is_initialized():
        return [obj]
    world = dist.get_world_size()
    out = [None] * world if dist.get_rank() == root else None
    dist.gather_object(obj, out, dst=root)
    return out


def sum_over_ranks(value: float, device: torch.device = None) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
