#!/usr/bin/env python3
"""Headline benchmark: DCT fingerprints/s on synthetic L=500, D=1280, 2-layer batches
(BASELINE.json config 2) on N MI355X, one process per GPU, sequences sharded, no collective.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (dctfp_quantize: job tables + stage A + stage B) over the
rank's whole batch, inputs resident in HBM.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n-seq', type=int, default=10000, help='sequences per GPU (weak scaling)')
    ap.add_argument('--seq-len', type=int, default=500)
    ap.add_argument('--dim', type=int, default=1280)
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='0 disables the CPU baseline leg')
    ap.add_argument('--cpu-procs', type=int, default=0,
                    help='CPU baseline worker processes (0 = usable cores, at most 16 = one GPU\'s CPU share on the pool)')
    ap.add_argument('--parity-sample', type=int, default=8)
    ap.add_argument('--opt', action='append', default=[], help='name=value passed to dctfp_set_option')
    return ap.parse_args()


def make_layer(torch, gen, n_seq, L, D, device):
    """ESM-like synthetic embeddings, generated on the device in slabs:
    randn * exp(N(0,1) per channel) + N(0,5) per channel, 1 % channels offset by +-200."""
    x = torch.empty((n_seq * L, D), dtype=torch.float32, device=device)
    ch_scale = torch.exp(torch.randn((1, D), generator=gen, device=device))
    ch_off = 5.0 * torch.randn((1, D), generator=gen, device=device)
    n_out = max(1, D // 100)
    idx = torch.randperm(D, generator=gen, device=device)[:n_out]
    sign = (torch.rand(n_out, generator=gen, device=device) < 0.5).float() * 2 - 1
    ch_off[0, idx] += 200.0 * sign
    slab = max(1, (1 << 28) // (L * D))           # about 1 GiB of floats per slab
    for s0 in range(0, n_seq, slab):
        s1 = min(n_seq, s0 + slab)
        v = x[s0 * L:s1 * L]
        torch.randn(v.shape, generator=gen, device=device, out=v)
        v.mul_(ch_scale).add_(ch_off)
    return x


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    # ---- CPU baseline leg first (rank 0, N = 1 only), before this process touches the GPU ----
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import cpu_baseline as cb
        procs = args.cpu_procs if args.cpu_procs > 0 else min(cb.usable_cores(), 16)
        cpu_baseline = cb.run(args.seq_len, args.dim, args.layers, tuple([3, 80] * args.layers), args.cpu_seconds,
                              procs=procs)

    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; the product path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    import dctdomain_amd as dd
    from dctdomain_amd import dist as ddist
    ddist.init('nccl', device)      # "nccl" is RCCL on ROCm; used for the barrier + max-over-ranks only

    ctx = dd.get_context(local_rank)
    for kv in args.opt:
        k, v = kv.split('=')
        ctx.set_option(k, int(v))

    n_seq, L, D = args.n_seq, args.seq_len, args.dim
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    layers = [make_layer(torch, gen, n_seq, L, D, device) for _ in range(args.layers)]
    offs = np.arange(n_seq, dtype=np.int64) * L
    table = dd.PieceTable.whole_sequences([L] * n_seq)
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    out = torch.empty((n_seq, 240 * args.layers), dtype=torch.int8, device=device)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    ctx.set_option('profile', 1)
    barrier()
    ctx.profile()                                   # reset the event accumulators
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ms_k, n_k = ctx.profile()                       # hipEvent time of stage A / stage B on the launch stream
    ctx.set_option('profile', 0)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- parity sample against the oracle (checker only; outside the timed region) ----
    parity = None
    if rank == 0 and args.parity_sample > 0:
        from oracle import dct_oracle as orc
        host = out.cpu().numpy()
        pick = np.linspace(0, n_seq - 1, args.parity_sample).astype(int)
        bad = 0
        for s in pick:
            ls = [x[s * L:(s + 1) * L].cpu().numpy() for x in layers]
            q = orc.quantize(ls, [f'1-{L}'], [3, 80] * args.layers)[f'1-{L}']
            bad += int(np.any(host[s].astype(np.int64) != q))
        parity = {'checked': int(len(pick)), 'mismatching_fingerprints': bad}

    if rank == 0:
        total_fp = n_seq * world * args.steps
        value = total_fp / elapsed
        bytes_per_fp = args.layers * L * D * 4 + 240 * args.layers          # 5,120,480 B at C2
        a_launch_ms = ms_k[0] / max(1, n_k[0])
        a_bytes = bytes_per_fp * n_seq * (args.steps / max(1, n_k[0]))      # units one stage-A launch processes
        achieved = a_bytes / (a_launch_ms * 1e-3) / 1e9 if a_launch_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile) and n_seq == 10000 and L == 500 and D == 1280:
            with open(tfile) as fh:
                traffic = json.load(fh).get('stage_a_hbm_bytes_per_launch')
        line = {
            'metric': 'DCT fingerprints/sec on L=500 D=1280', 'value': value, 'unit': 'fingerprints/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'C2: {n_seq} sequences/GPU x {args.layers} layers of L={L} x D={D} float32 '
                                   f'(ESM-like synthetic), one whole-sequence domain each, qdim [3,80]x{args.layers} '
                                   f'-> {240 * args.layers} int8 per fingerprint',
                       'sequences_per_gpu': n_seq, 'L': L, 'D': D, 'layers': args.layers, 'sharding': f'seq{world}'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'stage_a_kernel', 'avg_launch_ms': a_launch_ms,
                         'algorithmic_bytes_per_launch': a_bytes,
                         'stage_b_avg_launch_ms': ms_k[1] / max(1, n_k[1]),
                         'whole_path_GBps': value / world * bytes_per_fp / 1e9},
            'cpu_baseline': cpu_baseline,
            'parity': parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
