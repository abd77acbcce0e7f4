#!/bin/bash
# round 4, session l: row select in registers: parity, rate of the three consumer kernels
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
( timeout -k 10 400 python -m pytest tests/test_similarity_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r04/sim_tests.txt 2>&1 &&
timeout -k 10 300 python tools/sim_probe.py > gpurun_out/r04/sim_probe_after.txt 2>&1
echo "rc=$?"; cat gpurun_out/r04/sim_tests.txt; cat gpurun_out/r04/sim_probe_after.txt
