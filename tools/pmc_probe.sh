#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Runs on the GPU box (via gpurun): PMC passes of tools/path_probe.py with ONE configuration, to see what the waves of a
# kernel spend their cycles on.  usage: tools/pmc_probe.sh <tag> <workload> <cfg>      e.g.  r02_c5_walk c5 path=2
set -o pipefail
TAG=$1; W=$2; CFG=$3
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
P="python3 $REPO/tools/path_probe.py $W $CFG"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $P > "$OUT/log.txt" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- $P >> "$OUT/log.txt" 2>&1 || echo "second SQ pass failed" >> "$OUT/log.txt"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $P >> "$OUT/log.txt" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $P >> "$OUT/log.txt" 2>&1 || exit 1
cd "$REPO"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.md" 2>> "$OUT/log.txt"
find "$OUT" -name '*.db' -delete
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
grep -E "walk_ab|stage_a|stage_b" "$OUT/summary.md"
