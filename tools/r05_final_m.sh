#!/bin/bash
# round 5, session m2: walk_ab_kernel with six column groups (80 < m <= 96: PROST's [3, 85]): parity, then the mixes at [3, 85]
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_walk_general.py tests/test_gpu_parity.py tests/test_walk_kernel.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r05/walk_six_groups_tests.txt 2>&1 || { cat gpurun_out/r05/walk_six_groups_tests.txt; exit 1; }
tail -2 gpurun_out/r05/walk_six_groups_tests.txt
timeout -k 10 600 python tools/gen_probe.py c5 c4 c3 > gpurun_out/r05/gen_probe_six_groups.txt 2>&1 || { tail -30 gpurun_out/r05/gen_probe_six_groups.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/gen_probe_six_groups.txt
