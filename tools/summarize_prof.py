#!/usr/bin/env python3
"""Condenses the rocprofv3 CSV output of tools/profile_gpu.sh into a small markdown summary
(per-kernel time stats, PMC counters per launch with the gfx950 FETCH_SIZE correction)."""

import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, '**', pat), recursive=True))


def short(name):
    name = name.split('(')[0]
    for key in ('walk_ab_kernel', 'walk_gen_kernel', 'stage_a_kernel', 'stage_b_mfma_kernel', 'stage_b_valu_kernel', 'basis_kernel'):
        if key in name:
            return key
    return name[-60:]


print(f'# rocprofv3 summary ({os.path.basename(out)})\n')
for f in find('trace', '*kernel_stats.csv'):
    print('## kernel stats (rocprofv3 --kernel-trace --stats)\n')
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    with open(f) as fh:
        for r in csv.DictReader(fh):
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                  f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {r['Percentage']} |")
    print()

# Steady state of the dominant kernel from the per-dispatch trace: `--stats` averages the first (cold: code object, page
# faults, first touch of the tables) launch into its mean, which is what the bench line's roofline is NOT computed from --
# bench.py times its K steps after W warm-up steps.  Drop the first WARMUP launches (profile_gpu.sh runs --warmup 1) so that
# algorithmic bytes / this mean / 8 TB/s reproduces roofline.frac of the line printed in the same run (log.txt).
WARMUP = int(os.environ.get('PROF_WARMUP', '1'))
for f in find('trace', '*kernel_trace.csv'):
    runs = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = short(r.get('Kernel_Name', ''))
            if name in ('walk_ab_kernel', 'walk_gen_kernel', 'stage_a_kernel'):
                runs.append((int(r['Start_Timestamp']), name, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    runs.sort()
    if runs:
        main_k = max(set(n for _, n, _ in runs), key=lambda k: sum(d for _, n, d in runs if n == k))
        dur = [d for _, n, d in runs if n == main_k]
        steady = dur[WARMUP:] if len(dur) > WARMUP else dur
        mean = sum(steady) / len(steady)
        med = sorted(steady)[len(steady) // 2]
        print(f'## {main_k}: steady state (first {WARMUP} launch(es) = warm-up, dropped)\n')
        print(f'- launches in order, us: {[round(d, 1) for d in dur]}')
        print(f'- steady-state launches: {len(steady)}, mean {mean:.1f} us, median {med:.1f} us, min {min(steady):.1f} us, max {max(steady):.1f} us')
        line = None
        log = os.path.join(out, 'log.txt')
        if os.path.exists(log):
            for ln in open(log, errors='replace'):
                if ln.startswith('{') and '"roofline"' in ln:
                    line = json.loads(ln)
                    break
        if line:
            rf = line['roofline']
            ab = rf['algorithmic_bytes_per_launch']
            print(f"- algorithmic bytes per launch {ab / 1e9:.4f} GB / steady-state mean = {ab / mean / 1e3:.0f} GB/s = "
                  f"{ab / mean / 1e3 / rf['peak']:.4f} of {rf['peak']:.0f} GB/s; the bench line of the same (profiled) run: "
                  f"avg_launch_ms {rf['avg_launch_ms']:.4f} -> frac {rf['frac']:.4f}")
        print()
    break

counters = defaultdict(lambda: defaultdict(list))
for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2'):
    for f in find(sub, '*counter_collection.csv'):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                counters[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
if counters:
    print('## PMC counters (mean per launch; separate passes)\n')
    print('| kernel | counter | launches | mean |')
    print('|---|---|---|---|')
    res = {}
    for k, cs in sorted(counters.items()):
        for c, vals in sorted(cs.items()):
            m = sum(vals) / len(vals)
            print(f'| {k} | {c} | {len(vals)} | {m:.6g} |')
            res.setdefault(k, {})[c] = m
    print()
    main = 'walk_ab_kernel' if 'walk_ab_kernel' in res else ('walk_gen_kernel' if 'walk_gen_kernel' in res else 'stage_a_kernel')
    if main in res and 'FETCH_SIZE' in res[main]:
        a = res[main]
        # MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
        # reads exactly 1/2 of a wide coalesced stream's bytes -> double it; WRITE_SIZE is exact.
        fetch = 2.0 * a['FETCH_SIZE'] * 1024.0
        write = a.get('WRITE_SIZE', 0.0) * 1024.0
        print(f'## {main} HBM traffic per launch (gfx950 correction: 2 x FETCH_SIZE KiB + WRITE_SIZE KiB)\n')
        print(f'- fetch {fetch / 1e9:.3f} GB, write {write / 1e9:.3f} GB, total {(fetch + write) / 1e9:.3f} GB')
        with open(os.path.join(out, 'traffic.json'), 'w') as fh:
            json.dump({'kernel': main, 'stage_a_hbm_bytes_per_launch': fetch + write, 'fetch_bytes': fetch, 'write_bytes': write,
                       'raw': a}, fh, indent=1)
