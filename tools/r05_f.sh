#!/bin/bash
# round 5, session f: contact top-k in one read: parity, kernel time, and its HBM traffic (FETCH_SIZE in a pass of its own)
cd "$(dirname "$0")/.." && . tools/env.sh
bash tools/r05_c.sh || exit 1
( export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/prof_r05_topk_pmc; cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_r05_topk_pmc -- python3 $R/tools/topk_kernel_bench.py > $R/gpurun_out/r05/topk_kernel_bench_pmc.txt 2>&1 ) || { tail -30 gpurun_out/r05/topk_kernel_bench_pmc.txt; exit 1; }
python3 - <<'PY' | tee gpurun_out/r05/topk_fetch_size.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/prof_r05_topk_pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'topk' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            acc[(r['Kernel_Name'].split('(')[0], r.get('Dispatch_Id'))].append(float(r['Counter_Value']))
per = collections.defaultdict(list)
for (k, d), v in acc.items():
    per[k].append(sum(v))
tri = 4096 * 122760 * 4
for k, v in per.items():
    big = [x for x in v if x > max(v) / 10]
    mean = sum(big) / len(big)
    print(f'{k[:50]:50s} launches {len(v)} (with work: {len(big)}): FETCH_SIZE {mean:.0f} KiB-units -> 2 x = {2 * mean * 1024 / 1e9:.3f} GB per launch = {2 * mean * 1024 / tri:.3f} x the triangle ({tri / 1e9:.3f} GB)')
PY
rm -rf gpurun_out/prof_r05_topk_pmc
