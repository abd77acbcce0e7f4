#!/bin/bash
# round 5, session o: where the GPU cutter's time goes by length class (instrumented build, finer marks inside the scan)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
export DCTFP_LIBRARY=build_variants/cut_timing.so
{
timeout -k 10 300 python tools/cut_timing_probe.py 512 81 1330 || exit 1
timeout -k 10 300 python tools/cut_timing_probe.py 64 1025 1330 || exit 1
timeout -k 10 300 python tools/cut_timing_probe.py 128 513 1024 || exit 1
timeout -k 10 300 python tools/cut_timing_probe.py 256 150 512 || exit 1
} > gpurun_out/r05/cut_timing_by_class.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05/cut_timing_by_class.txt
