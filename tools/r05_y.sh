#!/bin/bash
# round 5, session y: band 1's start vector in `pos` (LDS back to 3 / 2 workgroups per CU), four bands in every class
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_reccut.py tests/test_make_db_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05/reccut_band_tests2.txt 2>&1 || { cat gpurun_out/r05/reccut_band_tests2.txt; exit 1; }
tail -2 gpurun_out/r05/reccut_band_tests2.txt
bash tools/r05_v.sh || exit 1
bash tools/r05_w.sh
