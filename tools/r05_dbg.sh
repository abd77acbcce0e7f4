#!/bin/bash
# round 5, debugging session: the 100 000-protein build that stopped at 92 180 proteins -- the whole stderr this time
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
export AMD_LOG_LEVEL=1
timeout -k 10 400 bash tools/db_build_scale.sh 100000 one > gpurun_out/r05/db_build_100k_dbg_$1.txt 2>&1
rc=$?
cp /tmp/dbs_one/time1.txt gpurun_out/r05/db_build_100k_dbg_$1_stderr.txt 2>/dev/null
tail -5 /tmp/dbs_one/log1.txt > gpurun_out/r05/db_build_100k_dbg_$1_log_tail.txt 2>/dev/null
grep -E "stage fingerprint|sha256|wall|committed" gpurun_out/r05/db_build_100k_dbg_$1.txt | tail -8
echo "rc $rc"
exit $rc
