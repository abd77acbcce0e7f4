#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of
# bench.py, outputs under gpurun_out/prof_<tag>/.  Usage: tools/profile_gpu.sh <tag> [bench args]
set -o pipefail
TAG=${1:-r01}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"   # (results of an earlier run would be averaged into the summary)
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $REPO/bench.py --cpu-seconds 0 --parity-sample 0 --workloads none $*"
echo "== kernel trace + stats" | tee "$OUT/log.txt"
# (the bench's own step count and warm-up: what `--stats` and the steady-state summary see is then the line's timed region)
export PROF_WARMUP=3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 20 --warmup 3 >> "$OUT/log.txt" 2>&1 || exit 1
echo "== pmc FETCH_SIZE" | tee -a "$OUT/log.txt"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH --steps 2 --warmup 1 >> "$OUT/log.txt" 2>&1 || exit 1
echo "== pmc WRITE_SIZE" | tee -a "$OUT/log.txt"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH --steps 2 --warmup 1 >> "$OUT/log.txt" 2>&1 || exit 1
echo "== pmc SQ busy / valu / mfma" | tee -a "$OUT/log.txt"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $BENCH --steps 2 --warmup 1 >> "$OUT/log.txt" 2>&1 || echo "sq pass failed (non fatal)" | tee -a "$OUT/log.txt"
cd "$REPO"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.md" 2>> "$OUT/log.txt"
# keep only the small files
find "$OUT" -name '*.db' -delete
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
find "$OUT" -path '*pmc_*' -name '*kernel_trace.csv' -delete
tail -40 "$OUT/summary.md"
