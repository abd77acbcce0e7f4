#!/bin/bash
# round 5, session v: the flush timeline with the GPU's own times for top-k / cutter / copy
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
timeout -k 10 400 python tools/flush_timeline.py 2048 tiefree > gpurun_out/r05/flush_timeline_v.txt 2>&1 || { tail -30 gpurun_out/r05/flush_timeline_v.txt; exit 1; }
timeout -k 10 400 python tools/flush_timeline.py 2048 >> gpurun_out/r05/flush_timeline_v.txt 2>&1 || { tail -30 gpurun_out/r05/flush_timeline_v.txt; exit 1; }
grep -E "best of|on the GPU|cutter waited|enqueued top-k" gpurun_out/r05/flush_timeline_v.txt
