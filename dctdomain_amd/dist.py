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


def gather_objects(obj) -> list:
    """One Python object per rank on every rank (a single-process run: [obj])."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def gpu_numa(device_index: int) -> dict:
    """Where a GPU sits: PCI address, NUMA node and the node's CPU list (from sysfs; empty strings / -1 where the box does
    not say).  No HIP call beyond torch's device properties."""
    info = {'pci': '', 'numa_node': -1, 'cpulist': ''}
    try:
        p = torch.cuda.get_device_properties(device_index)
        info['pci'] = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
        with open(f"/sys/bus/pci/devices/{info['pci']}/numa_node") as f:
            info['numa_node'] = int(f.read().strip())
        if info['numa_node'] >= 0:
            with open(f"/sys/devices/system/node/node{info['numa_node']}/cpulist") as f:
                info['cpulist'] = f.read().strip()
    except Exception:     # noqa: BLE001 -- placement facts are optional: no GPU, no sysfs entry, another torch
        pass
    return info


def _parse_cpulist(text: str) -> set:
    cpus = set()
    for part in text.split(','):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def pin_to_gpu_numa(device_index: int) -> dict:
    """Keeps this process (and the threads it starts: RecCut's pool, torch's) on the CPUs of its GPU's NUMA node -- the
    reference leaves placement to the OS (src/make_db.py:105-116); with eight workers on a two-socket host half of them
    would otherwise stage tables and results through the far socket.  Only narrows the current affinity; a box that does
    not report the node is left alone.  Returns what it found and did."""
    info = gpu_numa(device_index)
    info['pinned'] = 0
    try:
        allowed = os.sched_getaffinity(0)
        want = _parse_cpulist(info['cpulist']) & allowed
        if want and want != allowed:
            os.sched_setaffinity(0, want)
            info['pinned'] = len(want)
    except (OSError, AttributeError, ValueError):
        pass
    return info
