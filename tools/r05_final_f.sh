#!/bin/bash
# round 5, closing pass f: the flush's quantize on its own stream: make_db / bench tests, then the 100 000- and the 1 000 000-protein build
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_make_db_gpu.py tests/test_bench_launch.py tests/test_reccut.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r05/flush_stream_tests.txt 2>&1 || { cat gpurun_out/r05/flush_stream_tests.txt; exit 1; }
tail -2 gpurun_out/r05/flush_stream_tests.txt
timeout -k 10 900 bash tools/db_build_scale.sh 100000 one > gpurun_out/r05/db_build_100k_final.txt 2>&1 || { tail -30 gpurun_out/r05/db_build_100k_final.txt; exit 1; }
grep -E "stage|sha256|wall" gpurun_out/r05/db_build_100k_final.txt
timeout -k 10 1100 bash tools/db_build_scale.sh 1000000 one > gpurun_out/r05/db_build_1M_final.txt 2>&1 || { tail -30 gpurun_out/r05/db_build_1M_final.txt; exit 1; }
grep -E "stage|sha256|wall|resident" gpurun_out/r05/db_build_1M_final.txt
