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


def test_ragged_bench_workloads_are_one_job_dealt_to_the_ranks():
    """`bench.py --workload c5 --gpus N`: the N ranks hold a length-balanced partition of ONE workload of N x n_seq sequences
    (what make_db --gpu N does with a database), not N independent random mixes."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    argv, sys.argv = sys.argv, ['bench.py', '--workload', 'c5', '--n-seq', '500']
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    whole_l, whole_d, D = bench._ragged_workload(args, np, 1500)
    parts = [bench.make_workload(args, r, np, 3) for r in range(3)]
    assert sorted(np.concatenate([p[0] for p in parts]).tolist()) == sorted(whole_l.tolist())
    assert sum(len(p[1]) for p in parts) == len(whole_d) and all(p[2] == D == 640 for p in parts)
    rows = [int(p[0].sum()) for p in parts]
    assert max(rows) - min(rows) <= int(whole_l.max())
    for l, d, _ in parts:
        assert all(dl[-1] == f'1-{L}' for L, dl in zip(l.tolist(), d))        # every sequence kept its own domain list
    # one rank: the workload the single-GPU line has always used
    one = bench.make_workload(args, 0, np, 1)
    assert one[0].tolist() == bench._ragged_workload(args, np, 500)[0].tolist()


def test_numa_helpers_do_not_need_a_gpu():
    from dctdomain_amd import dist as dd_dist
    assert dd_dist._parse_cpulist('0-3,8,10-11') == {0, 1, 2, 3, 8, 10, 11}
    assert dd_dist._parse_cpulist('') == set()
    info = dd_dist.pin_to_gpu_numa(0)            # no GPU here: nothing found, nothing changed, no exception
    assert info['pinned'] == 0 and set(info) >= {'pci', 'numa_node', 'cpulist', 'pinned'}
