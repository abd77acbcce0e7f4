#!/bin/bash
# round 5, closing pass t: rocprofv3 kernel stats + PMC traffic of walk_ab_kernel's six-group builds (bench.py --qdim 3,85) on c4 and c5
cd "$(dirname "$0")/.." && . tools/env.sh
bash tools/profile_gpu.sh r05_c4_3x85 --workload c4 --n-seq 12000 --qdim 3,85 > gpurun_out/r05_profile_c4_3x85.log 2>&1 || { tail -30 gpurun_out/r05_profile_c4_3x85.log; exit 1; }
tail -8 gpurun_out/prof_r05_c4_3x85/summary.md
bash tools/profile_gpu.sh r05_c5_3x85 --workload c5 --n-seq 40000 --qdim 3,85 > gpurun_out/r05_profile_c5_3x85.log 2>&1 || { tail -30 gpurun_out/r05_profile_c5_3x85.log; exit 1; }
tail -8 gpurun_out/prof_r05_c5_3x85/summary.md
