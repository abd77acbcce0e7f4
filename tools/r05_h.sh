#!/bin/bash
# round 5, session h: the flush with the cutter on the GPU: make_db / reccut / windows tests, then the flush profile
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_reccut.py tests/test_make_db_gpu.py tests/test_stitch.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05/flush_tests.txt 2>&1 || { cat gpurun_out/r05/flush_tests.txt; exit 1; }
cat gpurun_out/r05/flush_tests.txt
timeout -k 10 600 python tools/flush_profile.py > gpurun_out/r05/flush_profile_gpu_cutter.txt 2>&1 || { tail -20 gpurun_out/r05/flush_profile_gpu_cutter.txt; exit 1; }
head -40 gpurun_out/r05/flush_profile_gpu_cutter.txt
