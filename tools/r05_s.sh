#!/bin/bash
# round 5, session s: small_call_kernel: the parity suite (goldens call by call, kernel variants, its own test), then the per-call rate
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fences.py tests/test_integration_stub.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r05/small_call_tests.txt 2>&1 || { cat gpurun_out/r05/small_call_tests.txt; exit 1; }
tail -3 gpurun_out/r05/small_call_tests.txt
timeout -k 10 600 python tools/pcie_rate.py profile > gpurun_out/r05/pcie_inclusive_rate_one_launch.txt 2>&1 || { tail -30 gpurun_out/r05/pcie_inclusive_rate_one_launch.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/pcie_inclusive_rate_one_launch.txt | head -12
