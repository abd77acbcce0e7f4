#!/bin/bash
# round 5, session w: which kernels the contact selection of a flush spends its time in, tie-free maps (rocprofv3 kernel stats)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/prof_r05_flush; cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_flush -- python3 $R/tools/flush_timeline.py 2048 tiefree > $R/gpurun_out/r05/flush_under_rocprof.txt 2>&1 ) || { tail -30 gpurun_out/r05/flush_under_rocprof.txt; exit 1; }
python3 - <<'PY' | tee gpurun_out/r05/flush_kernel_stats_tiefree.txt
import csv, glob
for f in glob.glob('gpurun_out/prof_r05_flush/*/*_kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in [x for x in rows if "dctfp" in x["Name"]][:24]:
        print(r['Name'].split('(')[0][:80], r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1), 'total ms', round(float(r['TotalDurationNs']) / 1e6, 2))
PY
rm -rf gpurun_out/prof_r05_flush
