#!/bin/bash
# round 4, session m: one-launch stitch: parity, next-rows numbers
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
( timeout -k 10 500 python -m pytest tests/test_stitch.py tests/test_similarity_gpu.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r04/stitch_tests.txt 2>&1 &&
timeout -k 10 400 python tools/next_rows_bench.py > gpurun_out/r04/next_rows_after.json 2> gpurun_out/r04/next_rows_after.err
echo "rc=$?"; cat gpurun_out/r04/stitch_tests.txt; cat gpurun_out/r04/next_rows_after.json; tail -3 gpurun_out/r04/next_rows_after.err
