#!/bin/bash
# round 5, session g: the domain cutter's recursion on the GPU: goldens + fuzz against the host library
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 600 python -m pytest tests/test_reccut.py -m gpu -x -q -k "reccut_goldens or reccut_fuzz" 2>&1 | tail -30 ) > gpurun_out/r05/reccut_gpu_tests.txt 2>&1
rc=$?
cat gpurun_out/r05/reccut_gpu_tests.txt
exit $rc
