#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# After tools/profile_gpu.sh <round>_c2 / _c3 / _c4 / _c5 ran on the GPU box: stamp profiles/traffic.json with the current
# kernel sources and copy the summaries the documents cite into profiles/<round>/.   usage: tools/collect_profiles.sh r02
set -e
cd "$(dirname "$0")/.."
R=${1:-r02}
python3 tools/stamp_traffic.py c2=gpurun_out/prof_${R}_c2 c3=gpurun_out/prof_${R}_c3 c4=gpurun_out/prof_${R}_c4 c5=gpurun_out/prof_${R}_c5
mkdir -p profiles/$R
for w in c2 c3 c4 c5; do
    d=gpurun_out/prof_${R}_$w
    cp $d/summary.md profiles/$R/rocprof_${w}_summary.md
    cp $d/traffic.json profiles/$R/traffic_$w.json
    cp "$(ls -t $d/trace/*/*_kernel_stats.csv | head -n 1)" profiles/$R/kernel_stats_$w.csv
    grep -h '^{"metric"' $d/log.txt > profiles/$R/bench_lines_under_rocprof_$w.jsonl || true
done
