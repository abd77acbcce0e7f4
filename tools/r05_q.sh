#!/bin/bash
# round 5, session q: fused walks in the general walk kernel: parity (test_walk_general), then the c5 / c4 / c3 mixes at PROST's kept sizes
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_walk_general.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r05/gen_fused_tests.txt 2>&1 || { cat gpurun_out/r05/gen_fused_tests.txt; exit 1; }
tail -3 gpurun_out/r05/gen_fused_tests.txt
timeout -k 10 600 python tools/gen_probe.py c5 c4 c3 > gpurun_out/r05/gen_probe_fused.txt 2>&1 || { tail -30 gpurun_out/r05/gen_probe_fused.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/gen_probe_fused.txt
