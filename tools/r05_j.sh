#!/bin/bash
# round 5, session j: the same-work microbenchmark of the fused walks (ceiling for the c4 / c5 job lengths)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/walk_same_work tools/microbench/walk_same_work.hip || exit 1
timeout -k 10 300 /tmp/walk_same_work 2>&1 | tee gpurun_out/r05/walk_same_work_microbench.txt
