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

# a fatal error names itself in this run's stderr (glibc's heap / stack messages otherwise go to /dev/tty; libdctfp.so
# prints the native frames of a SIGABRT / SIGSEGV): see tests/conftest.py
os.environ.setdefault('LIBC_FATAL_STDERR_', '1')
os.environ.setdefault('PYTHONFAULTHANDLER', '1')
os.environ.setdefault('DCTFP_CRASH_BACKTRACE', '1')

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n-seq', type=int, default=10000, help='sequences per GPU (weak scaling)')
    ap.add_argument('--seq-len', type=int, default=500)
    ap.add_argument('--dim', type=int, default=None, help='embedding width (default: 1280 for c2/c3, 2560 for c4, 640 for c5)')
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='0 disables the CPU baseline leg')
    ap.add_argument('--cpu-procs', type=int, default=0,
                    help='CPU baseline worker processes (0 = every usable core: os.cpu_count() clipped by the affinity mask and the '
                         'cgroup quota, at most 256)')
    ap.add_argument('--parity-sample', type=int, default=64)   # 64 sequences
    ap.add_argument('--opt', action='append', default=[], help='name=value passed to dctfp_set_option')
    ap.add_argument('--storage', choices=['float32', 'float16', 'bfloat16'], default='float32',
                    help='storage type of the synthetic embeddings (the headline metric is float32, as the reference)')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                    help='torch.distributed backend for the barrier / max-over-ranks (gloo: rehearsal of N > 1 on one GPU)')
    ap.add_argument('--qdim', default='3,80', help='kept points n,m of every layer (default: the reference\'s 3,80; 5,44 and 3,85 are '
                    'PROST\'s -- these run through walk_gen_kernel)')
    ap.add_argument('--diag', action='store_true', help='per-rank placement / clock / power block in the line also at N = 1')
    ap.add_argument('--diag-seconds', type=float, default=2.0, help='length of the untimed loop behind that block (N > 1)')
    ap.add_argument('--workload', choices=['c2', 'c3', 'c4', 'c5'], default='c2',
                    help='c2 = headline (fixed L, whole-sequence domains); c3 = ragged L in [50,2000] given as the language '
                         'model\'s windows (maxlen 500, overlap 200), averaged where they overlap and fingerprinted; '
                         'c4 = D=2560, L<=500, several domains per protein + whole protein; '
                         'c5 = database-build mix: pfam-like lengths, D=640, RecCut-like domains')
    ap.add_argument('--c3-form', choices=['fused', 'stitch', 'stitched'], default='fused',
                    help='c3: fused = dctfp_quantize_windows (shared rows averaged in the row load, the stitched matrix never written); '
                         'stitch = dctfp_stitch_sequences + dctfp_quantize per step (the materialising form); stitched = the timed '
                         'region starts from already stitched matrices (rounds 1-4; not BASELINE config 3 as stated)')
    ap.add_argument('--workloads', default='auto',
                    help='N = 1 only: further workloads measured after the headline and reported in the line\'s "workloads" object '
                         '(auto = c3,c4,c5,c4_pipeline behind the default c2 run, none otherwise; "none"; or a comma list; c4_pipeline = '
                         'BASELINE config 4 as stated: contact map -> top-k -> RecCut -> per-domain fingerprints at D = 2560)')
    ap.add_argument('--extra-steps', type=int, default=10, help='timed steps of each of those workloads (warm-up 2)')
    ap.add_argument('--pipeline-proteins', type=int, default=2048, help='proteins per flush of the c4_pipeline workload')
    return ap.parse_args()


def make_layer(torch, gen, n_rows, D, device):
    """ESM-like synthetic embeddings, generated on the device in slabs:
    randn * exp(N(0,1) per channel) + N(0,5) per channel, 1 % channels offset by +-200."""
    x = torch.empty((n_rows, D), dtype=torch.float32, device=device)
    ch_scale = torch.exp(torch.randn((1, D), generator=gen, device=device))
    ch_off = 5.0 * torch.randn((1, D), generator=gen, device=device)
    n_out = max(1, D // 100)
    idx = torch.randperm(D, generator=gen, device=device)[:n_out]
    sign = (torch.rand(n_out, generator=gen, device=device) < 0.5).float() * 2 - 1
    ch_off[0, idx] += 200.0 * sign
    slab = max(1, (1 << 28) // D)                 # about 1 GiB of floats per slab
    for r0 in range(0, n_rows, slab):
        v = x[r0:min(n_rows, r0 + slab)]
        torch.randn(v.shape, generator=gen, device=device, out=v)
        v.mul_(ch_scale).add_(ch_off)
    return x


def make_workload(args, rank, np, world: int = 1):
    """(lengths, domain strings per sequence or None for whole-sequence domains, D) of THIS rank.

    c2 is the same on every rank.  The ragged workloads are ONE job of `world` x n_seq sequences (same seed everywhere),
    dealt to the ranks by `dist.balanced_shards` -- length-balanced, as make_db --gpu N deals a database build -- so that the
    per-rank bytes differ by a sequence at most (independent random mixes per rank differed by the luck of the draw, and the
    slowest rank is what the line reports)."""
    n_seq = args.n_seq
    if args.workload == 'c2':
        return np.full(n_seq, args.seq_len, dtype=np.int64), None, args.dim or 1280
    lengths, doms, dim = _ragged_workload(args, np, n_seq * world)
    if world > 1:
        from dctdomain_amd.dist import balanced_shards
        mine = balanced_shards(lengths.tolist(), world)[rank]
        lengths = lengths[mine]
        doms = None if doms is None else [doms[i] for i in mine]
    return lengths, doms, dim


MAXLEN, OVERLAP = 500, 200      # BASELINE config 3: "maxlen=500 chunk + overlap averaging" (olp: src/embedding.py:163)


def split_windows(lengths, np, maxlen=MAXLEN, overlap=OVERLAP):
    """(window rows, windows per sequence) by Embedding.split_seq's rule (src/embedding.py:83-100): a sequence longer than
    maxlen is cut into windows of maxlen every maxlen - overlap residues, a window not longer than the overlap is dropped."""
    rows, counts = [], []
    for L in lengths.tolist():
        if L <= maxlen:
            w = [L]
        else:
            w = [min(maxlen, L - i) for i in range(0, L, maxlen - overlap)]
            w = [v for v in w if v > overlap]
        rows += w
        counts.append(len(w))
    return np.asarray(rows, dtype=np.int32), np.asarray(counts, dtype=np.int64)


def _ragged_workload(args, np, n_seq):
    rng = np.random.default_rng(2024)
    if args.workload == 'c3':       # BASELINE config 3: ragged lengths, whole-sequence domains
        return rng.integers(50, 2001, size=n_seq).astype(np.int64), None, args.dim or 1280
    if args.workload == 'c5':       # BASELINE config 5 flavour: what a database build feeds the path
        lengths = np.clip(rng.gamma(2.2, 170.0, size=n_seq).astype(np.int64), 81, 1330)
        dim, target = 640, 110      # esm2_t30 width; RecCut domains are ~100 residues
    else:                           # BASELINE config 4 flavour: D = 2560, L <= 500, short domains
        lengths = rng.integers(100, 501, size=n_seq).astype(np.int64)
        dim, target = 2560, 0
    doms = []
    for L in lengths:
        k = int(rng.integers(1, 7)) if target == 0 else max(1, int(round(int(L) / target + rng.normal(0, 0.7))))
        k = max(1, min(k, int(L) // 30))
        if k == 1:
            doms.append([f'1-{L}'])
            continue
        cuts = np.sort(rng.choice(np.arange(1, int(L) // 25), size=k - 1, replace=False)) * 25
        edges = [0] + [int(c) for c in cuts] + [int(L)]
        doms.append([f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:])] + [f'1-{L}'])
    return lengths, doms, args.dim or dim


def sample_clock_power(step, torch, device, dev_index, seconds: float) -> dict:
    """Loops `step` for `seconds` while a side thread asks rocm-smi for this card's shader clock and package power."""
    import subprocess
    import threading
    if seconds <= 0:
        return {}
    stop, sclk, power = threading.Event(), [], []

    def sampler():
        while not stop.is_set():
            try:
                r = subprocess.run(['rocm-smi', '-d', str(dev_index), '--showclocks', '--showpower', '--json'], capture_output=True,
                                   text=True, timeout=5)
                card = next(iter(json.loads(r.stdout).values()))
                for k, v in card.items():
                    if k.startswith('sclk clock speed'):
                        sclk.append(float(str(v).strip('()').lower().replace('mhz', '')))
                    elif 'Package Power' in k or 'Socket Power' in k:
                        power.append(float(v))
            except Exception:     # noqa: BLE001 -- diagnostics only: no rocm-smi, another json layout, a refused call
                pass
            stop.wait(0.4)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            step()
        torch.cuda.synchronize(device)
        n += 10
    loop_ms = 1e3 * (time.perf_counter() - t0) / max(1, n)
    stop.set()
    th.join(timeout=6)
    out = {'diag_loop_step_ms': round(loop_ms, 4)}
    if len(sclk) > 1:
        out['sclk_mhz'] = [round(sum(sclk[1:]) / len(sclk[1:])), round(min(sclk[1:]))]       # [mean, min]; first sample: ramp-up
    if len(power) > 1:
        out['package_power_w'] = round(sum(power[1:]) / len(power[1:]))
    return out


def source_sha256() -> str:
    """sha256 over the kernel sources (every translation unit + the two headers): what a PMC traffic measurement is valid for."""
    import hashlib
    import build_ext
    h = hashlib.sha256()
    for name in build_ext.kernel_sources():
        with open(name, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: start the N ranks as children of this process
    (`python -m torch.distributed.run`, rendezvous on 127.0.0.1), pass rank 0's JSON line through and
    return the launcher's exit status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def workload_text(wl, n_seq, n_fp, L, D, layers, storage, qn, qm, total_rows, extra=''):
    return {
        'c2': f'C2: {n_seq} sequences/GPU x {layers} layers of L={L} x D={D} {storage} (ESM-like synthetic), one whole-sequence '
              f'domain each, qdim [{qn},{qm}]x{layers} -> {qn * qm * layers} int8 per fingerprint',
        'c3': f'C3: {n_seq} sequences/GPU, L ~ U[50,2000] ({total_rows} rows), D={D} {storage}, {layers} layers, whole-sequence '
              f'domains, ragged batch{extra}',
        'c4': f'C4: {n_seq} sequences/GPU, L ~ U[100,500], D={D} {storage}, {layers} layers, 1-6 domains + whole protein '
              f'({n_fp} fingerprints)',
        'c5': f'C5 mix: {n_seq} sequences/GPU, pfam-like lengths 81-1330, D={D} {storage}, {layers} layers, ~110-residue domains + '
              f'whole protein ({n_fp} fingerprints)'}[wl]


def measure(args, wl, steps, warmup, env):
    """One workload on this rank: data resident in HBM, `warmup` untimed steps, exactly `steps` timed ones between barrier +
    synchronize on both sides, max over the ranks; parity sample against the oracle afterwards (rank 0).  Returns what the
    line (or its "workloads" entry) is made of."""
    np, torch, dd, dist, ddist = env['np'], env['torch'], env['dd'], env['dist'], env['ddist']
    rank, world, device, dev_index, ctx, nccl = env['rank'], env['world'], env['device'], env['dev_index'], env['ctx'], env['nccl']
    wargs = argparse.Namespace(**vars(args))
    wargs.workload = wl
    if wl != args.workload:     # a workload behind the headline runs at its own default size
        wargs.n_seq = {'c2': 10000, 'c3': 10000, 'c4': 12000, 'c5': 40000}[wl]
        wargs.dim = None
    L = args.seq_len
    lengths, doms, D = make_workload(wargs, rank, np, world)
    n_seq = len(lengths)
    total_rows = int(lengths.sum())             # rows of the sequences (c3: of the stitched sequences)
    windows = wl == 'c3' and args.c3_form != 'stitched'
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    if windows:
        win_rows, win_counts = split_windows(lengths, np)
        data_rows = int(win_rows.sum())         # what is resident, and read once per step: every window row
        offs = np.concatenate([[0], np.cumsum(win_rows)[:-1]]).astype(np.int64)
    else:
        data_rows = total_rows
        offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    layers = [make_layer(torch, gen, data_rows, D, device) for _ in range(args.layers)]
    if args.storage != 'float32':
        layers = [x.to(getattr(torch, args.storage)) for x in layers]
    t_tab = time.perf_counter()
    table = dd.PieceTable.whole_sequences(lengths) if doms is None else dd.PieceTable(lengths, doms)
    piece_table_ms = 1e3 * (time.perf_counter() - t_tab)      # outside the timed loop: a caller builds it once per batch
    qn, qm = (int(v) for v in args.qdim.split(','))
    lbs = [dd.LayerBatch(x, qn, qm, row_offsets=offs) for x in layers]
    n_fp = table.n_domains
    out = torch.empty((n_fp, qn * qm * args.layers), dtype=torch.int8, device=device)

    if not windows:
        def step():
            dd.quantize_batch(lbs, table, out=out, ctx=ctx)
    elif args.c3_form == 'fused':
        def step():     # BASELINE config 3 in one launch: shared rows averaged in the row load
            dd.quantize_windows(lbs, win_rows, win_counts, table, overlap=OVERLAP, out=out, ctx=ctx, fallback=False)
    else:
        from dctdomain_amd.batch import window_geometry
        from dctdomain_amd.embedding import stitch_windows_flat
        seq_win, sizes = window_geometry(win_rows, win_counts, OVERLAP)

        def step():     # ... and the materialising form: windows -> stitched matrices (HBM) -> fingerprints
            st = []
            for lb in lbs:
                big, first = stitch_windows_flat(lb, win_rows, seq_win, sizes, OVERLAP)
                st.append(dd.LayerBatch(big, qn, qm, row_offsets=first))
            dd.quantize_batch(st, table, out=out, ctx=ctx)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()

    for _ in range(warmup):
        step()
    # host side of one call with the GPU idle (C-side job tables + enqueue; inside the timed loop it runs under the previous
    # step's kernel)
    call_host_ms = []
    for _ in range(3):
        torch.cuda.synchronize(device)
        t_call = time.perf_counter()
        step()
        call_host_ms.append(1e3 * (time.perf_counter() - t_call))
    ctx.set_option('profile', 1)
    barrier()
    ctx.profile()                                   # reset the event accumulators
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ms_k, n_k = ctx.profile()                       # hipEvent time of stage A / stage B on the launch stream
    ctx.set_option('profile', 0)
    own_elapsed = elapsed
    elapsed = ddist.max_over_ranks(elapsed, device if nccl else None)
    # (the line reports max-over-ranks as the contract says; min and the rank-local kernel time make a bad curve readable)
    fastest = -ddist.max_over_ranks(-own_elapsed, device if nccl else None)
    own_kernel_ms = ms_k[0] / max(1, n_k[0])
    slowest_kernel_ms = ddist.max_over_ranks(own_kernel_ms, device if nccl else None)
    fastest_kernel_ms = -ddist.max_over_ranks(-own_kernel_ms, device if nccl else None)
    fp_all_ranks = ddist.sum_over_ranks(n_fp, device if nccl else None)   # ragged workloads differ per rank
    last_path = ctx.get_option('last_path')

    # ---- N > 1: what makes a bad curve readable from the line alone.  Per rank: where it ran, what it streamed, its own step
    # and kernel time -- and clock / package power sampled while the same step loops for two more seconds (the timed region
    # is too short for rocm-smi; the walks run a package into its power limit, eight of them share a chassis).
    per_rank = None
    if (world > 1 or args.diag) and wl == args.workload:
        diag = dict(rank=rank, cuda_index=dev_index, gb_per_step=round(args.layers * data_rows * D * layers[0].element_size() / 1e9, 3),
                    step_ms=round(1e3 * own_elapsed / steps, 4), kernel_launch_ms=round(own_kernel_ms, 4))
        try:        # (diagnostics must never cost the line: whatever fails here is reported in the block, and every rank
            #        still reaches the collective below)
            diag.update(device=torch.cuda.get_device_name(device), **ddist.gpu_numa(dev_index))
            diag.update(sample_clock_power(step, torch, device, dev_index, args.diag_seconds))
        except Exception as exc:     # noqa: BLE001
            diag['diag_error'] = repr(exc)[:200]
        per_rank = ddist.gather_objects(diag)
        if world > 1:
            dist.barrier()

    # ---- parity sample against the oracle (checker only; outside the timed region) ----
    parity = None
    if rank == 0 and args.parity_sample > 0:
        from oracle import dct_oracle as orc
        host = out.cpu().numpy()
        pick = np.linspace(0, n_seq - 1, args.parity_sample).astype(int)
        first_row = {}
        for row, s in enumerate(table.owner):
            first_row.setdefault(s, row)
        if windows:     # the reference's stitching expression on the CPU (torch float32), then its quantize
            from oracle import stitch_oracle as sto
            first_win = np.concatenate([[0], np.cumsum(win_counts)]).astype(np.int64)
        bad = checked = 0
        for s in pick:
            if windows:
                ls = []
                for x in layers:
                    ws = [x[int(offs[w]):int(offs[w] + win_rows[w])].float().cpu() for w in range(int(first_win[s]), int(first_win[s + 1]))]
                    ls.append(sto.stitch_embeddings(ws, OVERLAP).numpy())
                assert ls[0].shape[0] == int(lengths[s])
            else:
                a, b = int(offs[s]), int(offs[s] + lengths[s])
                ls = [x[a:b].float().cpu().numpy() for x in layers]
            dl = [f'1-{int(lengths[s])}'] if doms is None else doms[s]
            q = orc.quantize(ls, dl, [qn, qm] * args.layers)
            for k, key in enumerate(q):
                bad += int(np.any(host[first_row[s] + k].astype(np.int64) != q[key]))
                checked += 1
        parity = {'checked': checked, 'mismatching_fingerprints': bad}

    res = None
    if rank == 0:
        total_fp = fp_all_ranks * steps
        value = total_fp / elapsed
        # algorithmic bytes: every embedding row read once per layer + the int8 output; 5,120,480 B per
        # fingerprint at C2.  c3 from windows: every WINDOW row read once (the rows two windows share are two rows of input).
        esz = layers[0].element_size()
        batch_bytes = args.layers * data_rows * D * esz + qn * qm * args.layers * n_fp
        bytes_per_fp = batch_bytes / n_fp
        a_launch_ms = ms_k[0] / max(1, n_k[0])
        # (c3 in the materialising form: the walk kernel reads the STITCHED rows; the stitch kernel's time is in ms_per_step only)
        kernel_bytes = batch_bytes if not (windows and args.c3_form == 'stitch') else args.layers * total_rows * D * esz + qn * qm * args.layers * n_fp
        a_bytes = kernel_bytes * (steps / max(1, n_k[0]))               # units one stage-A launch processes
        achieved = a_bytes / (a_launch_ms * 1e-3) / 1e9 if a_launch_ms > 0 else 0.0
        # HBM traffic of the dominant kernel from the PMC passes (tools/profile_gpu.sh).  The file is stamped with the sha256
        # of the kernel sources it was measured on; a stamp that does not match the sources of THIS run means the number is
        # stale -> null, never a silently outdated constant.
        # (the walk kernel of the reference's shape, the general walk kernel for every other shape of float32 / float64 rows)
        tuned_shape = qn == 3 and 64 < qm <= (96 if args.storage == 'float32' else 80) and 512 <= D <= 2560 and D % 4 == 0
        kernel = ('walk_ab_kernel' if tuned_shape else 'walk_gen_kernel') if last_path == 2 else 'stage_a_kernel'
        traffic, traffic_note = None, 'no PMC measurement for this workload / source state'
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            with open(tfile) as fh:
                tj = json.load(fh)
            entry = tj.get('workloads', {}).get(wl)
            default_size = n_seq == {'c2': 10000, 'c3': 10000, 'c4': 12000, 'c5': 40000}[wl] and (wl != 'c2' or (L == 500 and D == 1280))
            if entry and tj.get('source_sha256') == source_sha256() and entry.get('kernel') == kernel and default_size \
                    and not args.opt and args.storage == 'float32' and args.qdim == '3,80' and world == 1 \
                    and (wl != 'c3' or args.c3_form == 'fused') and entry.get('algorithmic_bytes', batch_bytes) == batch_bytes:
                traffic = entry['hbm_bytes_per_launch']
                traffic_note = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {tj.get('measured', '')}: 2 x FETCH_SIZE KiB + WRITE_SIZE KiB"
        extra = ''
        if wl == 'c3' and not windows:
            extra = '; ALREADY STITCHED matrices (not BASELINE config 3 as stated)'
        elif wl == 'c3':
            extra = f'; given as {len(win_rows)} windows (maxlen {MAXLEN}, overlap {OVERLAP}: {data_rows} window rows), ' + (
                'the rows two windows share averaged in the row load (dctfp_quantize_windows), nothing but int8 written' if args.c3_form == 'fused'
                else 'stitched into HBM (dctfp_stitch_sequences) and fingerprinted (dctfp_quantize) in every step')
        res = {
            'value': value, 'ms_per_step': 1e3 * elapsed / steps, 'steps': steps, 'warmup': warmup,
            'config': {'workload': workload_text(wl, n_seq, n_fp, L, D, args.layers, args.storage, qn, qm, total_rows, extra),
                       'sequences_per_gpu': n_seq, 'fingerprints_per_gpu': n_fp, 'L': L if wl == 'c2' else None,
                       'D': D, 'layers': args.layers, 'sharding': f'seq{world}'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_note,
                         'kernel': kernel, 'avg_launch_ms': a_launch_ms,
                         'algorithmic_bytes_per_launch': a_bytes,
                         'stage_b_avg_launch_ms': ms_k[1] / max(1, n_k[1]),
                         'whole_path_GBps': value / world * bytes_per_fp / 1e9},
            'per_rank_ms': {'step': [1e3 * fastest / steps, 1e3 * elapsed / steps],
                            'kernel_launch': [fastest_kernel_ms, slowest_kernel_ms]},      # [min, max] over the ranks
            'per_rank': per_rank,
            # outside the timed loop: the piece table of the batch (built once by the caller) and the host side of one call
            'host_table_ms': {'piece_table': round(piece_table_ms, 3), 'quantize_call_idle_gpu': round(min(call_host_ms), 3),
                              'domains': n_fp},
            'parity': parity,
        }
        if wl == 'c3' and args.c3_form == 'stitch':
            # two kernels share the step: the stitch copies (windows read + stitched rows written), the walk kernel reads them back
            res['roofline']['note'] = ('avg_launch_ms / achieved are the walk kernel over the STITCHED bytes it reads; the step also holds '
                                       'the stitch kernel: see ms_per_step and whole_path_GBps (algorithmic bytes of the windows)')
    del layers, lbs, out
    torch.cuda.empty_cache()
    return res


def measure_pipeline(args, steps, env, n_prot: int = 2048, D: int = 2560):
    """BASELINE config 4 AS STATED -- "L x L contact-map (L <= 500) + RecCut domain split, per-domain fingerprint, D = 2560" -- as
    make_db runs a flush of it (make_db.flush_records): contact top-k (dctfp_contact_topk) -> the domain cutter's recursion
    (dctfp_reccut) -> strings + piece table (dctfp_reccut_pieces) -> dctfp_quantize -> (pid, domains, int8 rows) on the host.  Inputs
    (two embedding layers and the contact map per protein) resident in HBM; a step = one flush of `n_prot` proteins, in one piece
    (the wait for the cutter inside).  Parity sample: the CPU chain oracle top-k -> the reference's RecCut binary (oracle/_ref,
    when it is there) -> oracle quantize."""
    np, torch, dd, device = env['np'], env['torch'], env['dd'], env['device']
    from dctdomain_amd import make_db, reccut
    rng = np.random.default_rng(4242)
    lens = rng.integers(100, 501, size=n_prot).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    gen = torch.Generator(device=device)
    gen.manual_seed(99)
    layers = [make_layer(torch, gen, int(offs[-1]), D, device) for _ in range(2)]
    maps = []
    for L in lens.tolist():       # a decaying band + blocks of 90-150 residues that contact inside themselves: what RecCut cuts along
        i = torch.arange(L, device=device)
        blk = int(rng.integers(90, 151))
        near = 0.9 * torch.exp(-(i[:, None] - i[None, :]).abs().float() / 12.0)
        u = torch.rand((L, L), generator=gen, device=device)
        cm = near + 0.3 * u * ((i[:, None] // blk) == (i[None, :] // blk)) + 0.02 * torch.rand((L, L), generator=gen, device=device)
        maps.append((0.5 * (cm + cm.t())).clamp_(0, 1).contiguous())
    seqs = ['A' * int(L) for L in lens]

    def fresh():
        return [dd.Fingerprint(pid=f'p{s:05d}', seq=seqs[s], embed={15: layers[0][offs[s]:offs[s + 1]], 21: layers[1][offs[s]:offs[s + 1]]},
                               contacts=maps[s]) for s in range(n_prot)]
    for _ in range(2):
        recs = make_db.flush_records(fresh(), threads=16)
    path = make_db.LAST_PATH[0]
    batches = [fresh() for _ in range(steps)]
    # (the inputs of ALL steps exist before the loop -- steps x n_prot objects with their dicts: park them where the cyclic collector
    #  does not walk them every time a flush's own few thousand containers trip it; a build holds two flushes' worth, not ten)
    import gc
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for b in batches:
        recs = make_db.flush_records(b, threads=16)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    gc.unfreeze()
    make_db.MARKS = []                     # one more flush with the GPU's own times (events on the side stream) and the host marks
    recs = make_db.flush_records(fresh(), threads=16)
    marks, make_db.MARKS = make_db.MARKS, None
    gpu_ms = getattr(reccut.LAST, 'gpu_ms', None)
    at = {name: t for name, t in marks}
    n_fp = sum(len(r[1]) for r in recs)
    # ---- parity sample (checker only, outside the timed region)
    parity = None
    if args.parity_sample > 0:
        from oracle import contacts_oracle as co
        from oracle import dct_oracle as orc
        have_ref = os.path.exists(co.REF_BIN)
        bad_dom = bad_fp = checked = doms_checked = 0
        for s in np.linspace(0, n_prot - 1, min(args.parity_sample, 32)).astype(int):
            pid, doms, rows8 = recs[s]
            L = int(lens[s])
            if have_ref:
                ci, cj, cv = co.top_contacts(maps[s].cpu().numpy(), 2.6)
                rc, out_txt = co.run_ref_binary(co.ce_text(pid, seqs[s], ci, cj, cv), pid)
                bad_dom += int(rc != 0 or co.parse_reccut(out_txt, L) != doms)
                doms_checked += 1
            q = orc.quantize([x[offs[s]:offs[s + 1]].cpu().numpy() for x in layers], doms, [3, 80, 3, 80])
            for k, key in enumerate(q):
                bad_fp += int(np.any(rows8[k].astype(np.int64) != q[key]))
                checked += 1
        parity = {'checked': checked, 'mismatching_fingerprints': bad_fp, 'domain_lists_checked_against_the_reference_binary': doms_checked,
                  'mismatching_domain_lists': bad_dom}
    res = {
        'value': n_fp * steps / elapsed, 'unit': 'fingerprints/s', 'proteins_per_s': n_prot * steps / elapsed,
        'us_per_protein': 1e6 * elapsed / (steps * n_prot), 'ms_per_step': 1e3 * elapsed / steps, 'steps': steps,
        'config': {'workload': f'C4 as stated: {n_prot} proteins per flush, L ~ U[100,500], per protein an L x L contact map (synthetic: band + '
                               f'90-150-residue blocks) -> top {2.6} L contacts -> RecCut -> per-domain + whole-protein fingerprints from 2 layers of '
                               f'D={D} float32, qdim [3,80]x2; {n_fp} fingerprints per flush; make_db.flush_records (path: {path})',
                   'proteins_per_flush': n_prot, 'fingerprints_per_flush': n_fp, 'D': D, 'layers': 2},
        'gpu_ms': None if not gpu_ms else {'contact_topk': round(gpu_ms[0], 3), 'domain_cutter': round(gpu_ms[1], 3)},
        'host_ms': None if 'cutter waited for' not in at else {
            'first_half': round(1e3 * (at['embedding tables'] - at['start']), 3),
            'wait_for_the_cutter': round(1e3 * (at['cutter waited for'] - at['embedding tables']), 3),
            'second_half': round(1e3 * (at['results on the host'] - at['cutter waited for']), 3)},
        'roofline': None,      # (a latency chain -- a few long recursions, one workgroup each -- not a stream: DESIGN section 5)
        'parity': parity,
    }
    del layers, maps, batches
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # Started without a launcher: be the launcher (the reference starts its own per-GPU processes too,
        # src/make_db.py:105-116).  This parent never touches the GPU; the ranks are fresh child processes.
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    # ---- CPU baseline leg first (rank 0, N = 1 only), before this process touches the GPU ----
    # The faithful oracle form in one process per usable core, as the reference's Pool(cpu) runs it (src/make_db.py:48-49;
    # P = os.cpu_count(), clipped here by the affinity mask and the cgroup quota -- what the box really grants).
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import cpu_baseline as cb
        procs = args.cpu_procs if args.cpu_procs > 0 else cb.usable_cores(cap=256)
        cpu_baseline = cb.run(args.seq_len, args.dim or 1280, args.layers, tuple([3, 80] * args.layers), args.cpu_seconds,
                              procs=procs)
        cpu_baseline['os_cpu_count'] = os.cpu_count()
        cpu_baseline['usable_cores'] = cb.usable_cores(cap=1 << 20)
        if procs > 16 and args.cpu_procs == 0:   # rounds 1-4 quoted 16 processes (one GPU's CPU share on the pool): kept beside it
            few = cb.run(args.seq_len, args.dim or 1280, args.layers, tuple([3, 80] * args.layers), min(6.0, args.cpu_seconds),
                         procs=16, matrix_seconds=0)
            cpu_baseline['sixteen_processes'] = {'value': few['value'], 'cores': 16, 'sample': few['sample']}

    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; the product path has no CPU fallback')
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and args.backend != 'gloo':
        raise SystemExit(f'LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible')
    dev_index = local_rank % n_dev               # gloo rehearsal: several ranks may share the one GPU
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    import dctdomain_amd as dd
    from dctdomain_amd import dist as ddist
    ddist.init(args.backend, device)   # "nccl" is RCCL on ROCm; used for the barrier + max-over-ranks only

    ctx = dd.get_context(dev_index)
    if rank == 0:       # stderr: which HIP / HSA runtime files the kernels run on (stdout carries the one JSON line)
        from dctdomain_amd import _lib as _ddlib
        print(_ddlib.runtime_report(), file=sys.stderr, flush=True)
    for kv in args.opt:
        k, v = kv.split('=')
        ctx.set_option(k, int(v))

    env = dict(np=np, torch=torch, dd=dd, dist=dist, ddist=ddist, rank=rank, world=world, device=device, dev_index=dev_index,
               ctx=ctx, nccl=args.backend == 'nccl')
    head = measure(args, args.workload, args.steps, args.warmup, env)

    # ---- N = 1: the other BASELINE configurations behind the headline, in the same line (what the driver's
    # run attests is then every config, not C2 alone).  Each at its default size, inputs resident, parity-sampled; the
    # headline fields are not touched.  Never at N > 1: that line is the scaling contract's.
    others = None
    want = args.workloads
    if want == 'auto':
        plain = args.workload == 'c2' and args.n_seq == 10000 and args.seq_len == 500 and args.dim in (None, 1280) and not args.opt \
            and args.storage == 'float32' and args.qdim == '3,80' and args.layers == 2
        want = 'c3,c4,c5' if plain else 'none'
    if world == 1 and want != 'none':
        others = {}
        for wl in want.split(','):
            if wl == 'c4_pipeline':
                continue
            r = measure(args, wl, args.extra_steps, 2, env)
            if r is not None:
                others[wl] = {'value': r['value'], 'unit': 'fingerprints/s', 'ms_per_step': r['ms_per_step'], 'steps': r['steps'],
                              'config': r['config'], 'roofline': r['roofline'], 'host_table_ms': r['host_table_ms'], 'parity': r['parity']}
        if args.workloads == 'auto' or 'c4_pipeline' in args.workloads.split(','):
            others['c4_pipeline'] = measure_pipeline(args, args.extra_steps, env, n_prot=args.pipeline_proteins)

    if rank == 0:
        line = {
            'metric': 'DCT fingerprints/sec on L=500 D=1280' if args.workload == 'c2' else f'DCT fingerprints/sec ({args.workload})',
            'value': head['value'], 'unit': 'fingerprints/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': head['config'], 'roofline': head['roofline'], 'per_rank_ms': head['per_rank_ms'], 'per_rank': head['per_rank'],
            'host_table_ms': head['host_table_ms'],
            'cpu_baseline': cpu_baseline,
            'parity': head['parity'],
            'per_layer_fingerprints_per_s': head['value'] * args.layers,   # 240-byte matrix fingerprints, same GB/s
        }
        if others is not None:
            line['workloads'] = others
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
