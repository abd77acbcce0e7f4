#!/bin/bash
# round 5, session u: the cutter's scan in row bands: goldens + fuzz, the kernel's time by class (instrumented build), the flush
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_reccut.py tests/test_make_db_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05/reccut_band_tests.txt 2>&1 || { cat gpurun_out/r05/reccut_band_tests.txt; exit 1; }
tail -3 gpurun_out/r05/reccut_band_tests.txt
{
DCTFP_LIBRARY=build_variants/cut_timing.so timeout -k 10 300 python tools/cut_timing_probe.py 512 81 1330 || exit 1
DCTFP_LIBRARY=build_variants/cut_timing.so timeout -k 10 300 python tools/cut_timing_probe.py 64 1025 1330 || exit 1
DCTFP_LIBRARY=build_variants/cut_timing.so timeout -k 10 300 python tools/cut_timing_probe.py 128 513 1024 || exit 1
DCTFP_LIBRARY=build_variants/cut_timing.so timeout -k 10 300 python tools/cut_timing_probe.py 256 150 512 || exit 1
} > gpurun_out/r05/cut_timing_by_class_bands.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05/cut_timing_by_class_bands.txt | grep -E "proteins|scan|pre/post|single"
timeout -k 10 400 python tools/flush_timeline.py 2048 tiefree > gpurun_out/r05/flush_timeline_bands.txt 2>&1 || { tail -30 gpurun_out/r05/flush_timeline_bands.txt; exit 1; }
grep -E "best of|GPU:|cutter waited" gpurun_out/r05/flush_timeline_bands.txt
