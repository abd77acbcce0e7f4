#!/bin/bash
# round 4, session n: contact top-k in two reads: parity, kernel times (two builds), the call as a flush makes it
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
( timeout -k 10 600 python -m pytest tests/test_reccut.py tests/test_make_db_gpu.py -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r04/topk_tests.txt 2>&1 &&
timeout -k 10 400 python tools/next_rows_bench.py > gpurun_out/r04/next_rows_topk2.json 2> gpurun_out/r04/next_rows_topk2.err
echo "rc=$?"; cat gpurun_out/r04/topk_tests.txt; grep -A10 contact_topk gpurun_out/r04/next_rows_topk2.json
for v in "" build_variants/topk4.so; do
( export TMPDIR=/tmp DCTFP_LIBRARY=$v; [ -z "$v" ] && unset DCTFP_LIBRARY; R=$PWD; rm -rf gpurun_out/prof_r04_topk2; cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_topk2 -- python3 $R/tools/topk_host_profile.py > $R/gpurun_out/r04/topk_host_profile_$(basename "$v" .so).txt 2>&1 ); find gpurun_out/prof_r04_topk2 -name '*.db' -delete; find gpurun_out/prof_r04_topk2 -name '*kernel_trace.csv' -delete
echo "== ${v:-default}"; grep -h "contact_topk\|contact_sort" gpurun_out/prof_r04_topk2/*/*_kernel_stats.csv | cut -c1-150; head -3 gpurun_out/r04/topk_host_profile_$(basename "$v" .so).txt | tail -1
done
