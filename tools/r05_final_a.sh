#!/bin/bash
# round 5, closing pass a: the whole GPU suite, smoke(), on the sources the profiles are taken on
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r05/gpu_tests_final.txt 2>&1
rc=$?
tail -6 gpurun_out/r05/gpu_tests_final.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r05/smoke_final.txt 2>&1 || { tail -20 gpurun_out/r05/smoke_final.txt; exit 1; }
tail -2 gpurun_out/r05/smoke_final.txt
sha256sum dctdomain_amd/*.so > gpurun_out/r05/final_sha256_on_gpu_box.txt
