#!/bin/bash
# round 5, session x: the two chains of the contact selection side by side: reccut / make_db tests, the flush timeline
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_reccut.py tests/test_make_db_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05/topk_beside_tests.txt 2>&1 || { cat gpurun_out/r05/topk_beside_tests.txt; exit 1; }
tail -2 gpurun_out/r05/topk_beside_tests.txt
bash tools/r05_v.sh
