#!/bin/bash
# round 5, session t: small_call_kernel again: its own test, the per-call rate with one launch and with three
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_calls or golden_bit_exact" 2>&1 | tail -5 ) > gpurun_out/r05/small_call_tests2.txt 2>&1 || { cat gpurun_out/r05/small_call_tests2.txt; exit 1; }
tail -2 gpurun_out/r05/small_call_tests2.txt
timeout -k 10 300 python tools/pcie_rate.py > gpurun_out/r05/pcie_rate_one_launch.txt 2>&1 || { tail -30 gpurun_out/r05/pcie_rate_one_launch.txt; exit 1; }
DCTFP_SMALL_ONE=0 timeout -k 10 300 python tools/pcie_rate.py > gpurun_out/r05/pcie_rate_three_launches.txt 2>&1 || { tail -30 gpurun_out/r05/pcie_rate_three_launches.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/pcie_rate_one_launch.txt | head -4; grep -v amdgpu.ids gpurun_out/r05/pcie_rate_three_launches.txt | head -4
