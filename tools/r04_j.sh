#!/bin/bash
# round 4, session j: rows of the next job requested into LDS before the epilogue (10-wave shape): parity on the variant, A/B
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
( DCTFP_LIBRARY=build_variants/pf6.so timeout -k 10 500 python -m pytest tests/test_walk_kernel.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r04/pf6_tests.txt 2>&1 &&
AB_ROUNDS=5 timeout -k 10 400 python tools/ab_many.py build_variants/base3.so build_variants/pf6.so build_variants/pf4.so -- c4 > gpurun_out/r04/ab_pf.txt 2>&1
echo "rc=$?"; cat gpurun_out/r04/pf6_tests.txt; cat gpurun_out/r04/ab_pf.txt
