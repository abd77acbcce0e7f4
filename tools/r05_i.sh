#!/bin/bash
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/prof_r05_cut; cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_cut -- python3 $R/tools/reccut_gpu_bench.py > $R/gpurun_out/r05/reccut_gpu_bench.txt 2>&1 ) || { tail -30 gpurun_out/r05/reccut_gpu_bench.txt; exit 1; }
grep -v rocprofv3 gpurun_out/r05/reccut_gpu_bench.txt | tail -5
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_r05_cut/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'topk' in r['Name'] or 'reccut' in r['Name']:
            print(r['Name'].split('(')[0][:70], r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1), 'max', round(float(r['MaxNs']) / 1e3, 1))
PY
rm -rf gpurun_out/prof_r05_cut
