#!/bin/bash
# round 5, session m: where the flush's 10.8 us per protein go (host marks + GPU events), synthetic and tie-free maps
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
timeout -k 10 400 python tools/flush_timeline.py 2048 > gpurun_out/r05/flush_timeline.txt 2>&1 || { tail -20 gpurun_out/r05/flush_timeline.txt; exit 1; }
timeout -k 10 400 python tools/flush_timeline.py 2048 tiefree >> gpurun_out/r05/flush_timeline.txt 2>&1 || { tail -20 gpurun_out/r05/flush_timeline.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/flush_timeline.txt
