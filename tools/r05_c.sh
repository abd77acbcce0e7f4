#!/bin/bash
# round 5, session c: contact top-k in one read + the sixteen-way selection (also under the row select): parity, then the
# kernels' own times on 4 096 distinct maps
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_reccut.py tests/test_make_db_gpu.py tests/test_similarity_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05/topk1_tests.txt 2>&1 || { cat gpurun_out/r05/topk1_tests.txt; exit 1; }
cat gpurun_out/r05/topk1_tests.txt
( export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/prof_r05_topk; cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_topk -- python3 $R/tools/topk_kernel_bench.py > $R/gpurun_out/r05/topk_kernel_bench.txt 2>&1 ) || { tail -30 gpurun_out/r05/topk_kernel_bench.txt; exit 1; }
cat gpurun_out/r05/topk_kernel_bench.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_r05_topk/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'topk' in r['Name']:
            print(r['Name'].split('(')[0][:60], r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1), 'min', round(float(r['MinNs']) / 1e3, 1), 'max', round(float(r['MaxNs']) / 1e3, 1))
PY
cp gpurun_out/prof_r05_topk/*/*_kernel_stats.csv gpurun_out/r05/topk_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof_r05_topk
