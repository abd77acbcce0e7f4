#!/bin/bash
# round 5, closing pass j: the cutter's scratch guarded across streams + the test that keeps the context busy from two streams; profile
# passes on these sources (the stamp)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_make_db_gpu.py tests/test_reccut.py tests/test_stitch.py tests/test_windows.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r05/two_streams_tests.txt 2>&1 || { cat gpurun_out/r05/two_streams_tests.txt; exit 1; }
tail -2 gpurun_out/r05/two_streams_tests.txt
bash tools/r05_final_b.sh c2 c3 c4 c5 > gpurun_out/r05/profile_pass_j.txt 2>&1 || { tail -20 gpurun_out/r05/profile_pass_j.txt; exit 1; }
grep -h "frac" gpurun_out/prof_r05_c*/summary.md
