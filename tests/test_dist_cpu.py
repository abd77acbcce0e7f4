"""The N > 1 control plane (sharding, barrier, max-over-ranks timing, gather to the writer)
on CPU with the gloo backend, world size 2."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lengths, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from dctdomain_amd import dist as dd
    r, w = dd.init('gloo')
    assert (r, w) == (rank, world)
    shards = dd.balanced_shards(lengths, world)
    mine = shards[rank]
    # stand-in for the GPU work of this shard: a deterministic 480-byte record per sequence
    fps = np.stack([np.full(480, i % 128, np.int8) for i in mine]) if mine else np.zeros((0, 480), np.int8)
    dd.barrier()
    t = dd.max_over_ranks(1.0 + rank)
    assert dd.sum_over_ranks(10 + rank) == 21.0
    gathered = dd.gather_to_root((mine, fps))
    if rank == 0:
        ret['t'] = t
        ret['idx'] = [g[0] for g in gathered]
        ret['sum'] = [int(g[1].astype(np.int64).sum()) for g in gathered]
    dist.destroy_process_group()


def test_world2_gloo_sharding_and_gather():
    rng = np.random.default_rng(3)
    lengths = [int(v) for v in rng.integers(50, 2000, size=41)]
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, lengths, ret), nprocs=2, join=True)
    assert ret['t'] == 2.0                       # max over ranks
    idx = ret['idx']
    assert sorted(idx[0] + idx[1]) == list(range(41))          # complete, disjoint
    load = [sum(lengths[i] for i in s) for s in idx]
    assert abs(load[0] - load[1]) <= max(lengths)              # balanced by residues
    assert ret['sum'] == [sum((i % 128) * 480 for i in s) for s in idx]


def test_balanced_shards_deterministic():
    from dctdomain_amd.dist import balanced_shards
    lengths = [500] * 16
    s = balanced_shards(lengths, 8)
    assert all(len(x) == 2 for x in s)
    assert balanced_shards(lengths, 8) == s
    assert balanced_shards([], 4) == [[], [], [], []]
    assert balanced_shards([10, 1, 1, 1], 2) == [[0], [1, 2, 3]]
