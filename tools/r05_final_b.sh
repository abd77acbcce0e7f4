#!/bin/bash
# round 5, closing pass b: rocprofv3 kernel stats + PMC passes of bench.py for the workloads named on the command line
cd "$(dirname "$0")/.." && . tools/env.sh
for w in "$@"; do
  extra=""; [ $w != c2 ] && extra="--workload $w"
  [ $w = c4 ] && extra="$extra --n-seq 12000"
  [ $w = c5 ] && extra="$extra --n-seq 40000"
  bash tools/profile_gpu.sh r05_$w $extra > gpurun_out/r05_profile_$w.log 2>&1 || { tail -30 gpurun_out/r05_profile_$w.log; exit 1; }
  tail -12 gpurun_out/prof_r05_$w/summary.md
done
