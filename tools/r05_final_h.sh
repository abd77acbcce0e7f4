#!/bin/bash
# round 5, closing pass h: every table upload now waits for the readers of another stream (stitch_impl did not: a flush's cutter on its
# side stream read a job table the caller's stream had overwritten -- the memory fault of session g at 92 180 proteins): the suites
# around it once, then the 100 000-protein build that faulted
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_make_db_gpu.py tests/test_reccut.py tests/test_stitch.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r05/table_guard_tests.txt 2>&1 || { cat gpurun_out/r05/table_guard_tests.txt; exit 1; }
tail -2 gpurun_out/r05/table_guard_tests.txt
timeout -k 10 400 bash tools/db_build_scale.sh 100000 one > gpurun_out/r05/db_build_100k_h.txt 2>&1
rc=$?
cp /tmp/dbs_one/time1.txt gpurun_out/r05/db_build_100k_h_stderr.txt 2>/dev/null
grep -E "stage fingerprint|sha256|wall|Memory|fault" gpurun_out/r05/db_build_100k_h.txt gpurun_out/r05/db_build_100k_h_stderr.txt | tail -8
echo "rc $rc"
exit $rc
