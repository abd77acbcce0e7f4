#!/bin/bash
# round 5, session r: whole GPU suite on the two-phase flush + fused general kernel, then the 100 000-protein build (hashes of round 3)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r05/gpu_tests_r.txt 2>&1
rc=$?
tail -4 gpurun_out/r05/gpu_tests_r.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 bash tools/db_build_scale.sh 100000 one > gpurun_out/r05/db_build_100k_two_phase.txt 2>&1 || { tail -30 gpurun_out/r05/db_build_100k_two_phase.txt; exit 1; }
tail -22 gpurun_out/r05/db_build_100k_two_phase.txt
