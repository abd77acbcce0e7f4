#!/bin/bash
# round 4, session k: the L1 matrix kernel on 16-byte segments (l1_matrix16_kernel): parity, rate
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
( timeout -k 10 400 python -m pytest tests/test_similarity_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r04/l1_tests.txt 2>&1 &&
timeout -k 10 300 python tools/l1_probe.py 8192 20000 40000 > gpurun_out/r04/l1_probe.txt 2>&1
echo "rc=$?"; cat gpurun_out/r04/l1_tests.txt; cat gpurun_out/r04/l1_probe.txt
