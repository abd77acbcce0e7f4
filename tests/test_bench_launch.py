"""bench.py --gpus N must work however it is started: with torchrun (the driver's way) or bare, in which case it
starts its own ranks (the reference starts its per-GPU processes itself, src/make_db.py:105-116)."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_bare_gpus2_starts_its_own_ranks_cpu():
    """No GPU here: the two child ranks must start (WORLD_SIZE = 2 each) and stop at the 'needs an MI355X' check --
    not at the old 'launch with torch.distributed.run' refusal of the parent."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('CPU-only check')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
                        '--cpu-seconds', '0', '--n-seq', '8'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'launch with torch.distributed.run' not in r.stderr
    # (the launcher ends the other rank as soon as one has failed: under load only the first may get to print)
    assert r.stderr.count('bench.py needs an MI355X') >= 1, r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize('launcher', ['bare', 'torchrun'])
def test_gpus2_on_one_gpu_over_gloo(launcher):
    """Two ranks sharing the one GPU (gloo control plane): one JSON line, n_gpus 2, parity sample clean."""
    args = ['--gpus', '2', '--backend', 'gloo', '--n-seq', '2000', '--steps', '3', '--warmup', '1', '--cpu-seconds', '0',
            '--parity-sample', '8']
    if launcher == 'bare':
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + args
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', '29731', os.path.join(ROOT, 'bench.py')] + args
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak'
    assert line['config']['sequences_per_gpu'] == 2000
    assert line['parity']['mismatching_fingerprints'] == 0 and line['parity']['checked'] == 8
    assert line['value'] > 0
    # [min, max] over the ranks: what makes a bad scaling curve diagnosable from the line alone
    step, kern = line['per_rank_ms']['step'], line['per_rank_ms']['kernel_launch']
    assert 0 < step[0] <= step[1] == pytest.approx(line['ms_per_step']) and 0 < kern[0] <= kern[1] <= step[1]
    # ... and per rank: where it ran, what it streamed, its own times (clock / power when rocm-smi answers)
    ranks = line['per_rank']
    assert [r['rank'] for r in ranks] == [0, 1] and all(r['gb_per_step'] > 0 and r['step_ms'] > 0 and 'pci' in r for r in ranks)
    assert line['host_table_ms']['piece_table'] >= 0 and line['host_table_ms']['quantize_call_idle_gpu'] > 0


@pytest.mark.gpu
def test_ragged_workload_is_dealt_to_the_ranks():
    """`--workload c5 --gpus 2`: ONE job of 2 x n_seq sequences, length-balanced over the ranks (dist.balanced_shards) -- the
    two ranks stream the same bytes to a sequence; parity sample (multi-domain proteins, fused walks) clean."""
    args = ['--gpus', '2', '--backend', 'gloo', '--workload', 'c5', '--n-seq', '1500', '--steps', '3', '--warmup', '1',
            '--cpu-seconds', '0', '--parity-sample', '6', '--diag-seconds', '0.5']
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert line['n_gpus'] == 2 and line['parity']['mismatching_fingerprints'] == 0 and line['parity']['checked'] >= 6
    gb = [r_['gb_per_step'] for r_ in line['per_rank']]
    assert abs(gb[0] - gb[1]) <= 0.01 * max(gb), gb          # (1 330 rows x 640 channels x 4 B x 2 layers = 0.007 GB at most)


def test_traffic_stamp_matches_the_kernel_sources():
    """`roofline.traffic` is printed only when profiles/traffic.json carries the sha256 of the kernel sources it was
    measured on (bench.source_sha256).  A kernel edit without a new PMC pass would silently print null in the judged
    line: the stamp of the committed file must be the committed sources'."""
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'profiles', 'traffic.json')) as f:
        tj = json.load(f)
    if tj['source_sha256'] != bench.source_sha256():
        msg = ('profiles/traffic.json is stale (the kernel sources changed after the PMC pass it holds): rerun '
               'tools/profile_gpu.sh + tools/collect_profiles.sh -- the judged bench line would print roofline.traffic = null')
        if os.environ.get('DCTFP_DEV_STALE_TRAFFIC_OK') == '1':      # mid-development only, asked for explicitly
            pytest.xfail(msg)
        pytest.fail(msg)
    for w in ('c2', 'c4', 'c5'):
        assert tj['workloads'][w]['kernel'] == 'walk_ab_kernel' and tj['workloads'][w]['hbm_bytes_per_launch'] > 0


@pytest.mark.gpu
def test_pipeline_workload_in_the_line():
    """workloads.c4_pipeline (BASELINE config 4 as stated: contact map -> top-k -> RecCut on the GPU -> fingerprints at D = 2560): in
    the line, through the two-phase flush, its parity sample -- incl. the domain lists against the reference's binary where
    oracle/_ref is built -- clean."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--n-seq', '600', '--steps', '2', '--warmup', '1', '--cpu-seconds', '0',
                        '--parity-sample', '6', '--workloads', 'c4_pipeline', '--extra-steps', '2', '--pipeline-proteins', '192'],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    w = line['workloads']['c4_pipeline']
    assert 'path: flush' in w['config']['workload'] and w['config']['proteins_per_flush'] == 192
    assert w['value'] > 0 and w['us_per_protein'] > 0 and w['gpu_ms']['domain_cutter'] > 0
    assert w['parity']['checked'] >= 6 and w['parity']['mismatching_fingerprints'] == 0 and w['parity']['mismatching_domain_lists'] == 0
    assert line['parity']['mismatching_fingerprints'] == 0
