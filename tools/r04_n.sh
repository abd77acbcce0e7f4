#!/bin/bash
# round 4, session n: contact top-k in two reads: parity, kernel times on random and banded maps, the call as a flush makes it
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
( timeout -k 10 600 python -m pytest tests/test_reccut.py tests/test_make_db_gpu.py -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r04/topk_tests.txt 2>&1 || { cat gpurun_out/r04/topk_tests.txt; exit 1; }
cat gpurun_out/r04/topk_tests.txt
( export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/prof_r04_topk2; cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04_topk2 -- python3 $R/tools/topk_host_profile.py > $R/gpurun_out/r04/topk_host_profile.txt 2>&1 )
head -6 gpurun_out/r04/topk_host_profile.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r04_topk2/*/*_kernel_trace.csv')[0]
d = [(r['Kernel_Name'].split('(')[0], int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(f)) if 'contact_topk2' in r['Kernel_Name']]
print('contact_topk2_kernel launches in order (us):', [round(t / 1e3) for _, t in d])
PY
rm -rf gpurun_out/prof_r04_topk2
