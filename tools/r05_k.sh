#!/bin/bash
# round 5, session k: one protein per call through dctfp_quantize_one: goldens, the per-call rate
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fences.py tests/test_integration_stub.py tests/test_reccut.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05/quantize_one_tests.txt 2>&1 || { cat gpurun_out/r05/quantize_one_tests.txt; exit 1; }
tail -3 gpurun_out/r05/quantize_one_tests.txt
timeout -k 10 300 python tools/pcie_rate.py profile > gpurun_out/r05/pcie_inclusive_rate.txt 2>&1 || { tail gpurun_out/r05/pcie_inclusive_rate.txt; exit 1; }
head -24 gpurun_out/r05/pcie_inclusive_rate.txt | cut -c1-160
