#!/bin/bash
# round 5, session p: the two-phase flush: its tests, the make_db / reccut suites around it, the flush timeline
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_make_db_gpu.py tests/test_reccut.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r05/flush2_tests.txt 2>&1 || { cat gpurun_out/r05/flush2_tests.txt; exit 1; }
tail -3 gpurun_out/r05/flush2_tests.txt
timeout -k 10 400 python tools/flush_timeline.py 2048 > gpurun_out/r05/flush_timeline_two_phase.txt 2>&1 || { tail -30 gpurun_out/r05/flush_timeline_two_phase.txt; exit 1; }
timeout -k 10 400 python tools/flush_timeline.py 2048 tiefree >> gpurun_out/r05/flush_timeline_two_phase.txt 2>&1 || { tail -30 gpurun_out/r05/flush_timeline_two_phase.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/flush_timeline_two_phase.txt
