#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# usage: tools/workloads_run.sh  -- default bench line + the other workloads (no CPU leg), one JSON line each
python bench.py 2>/dev/null
python bench.py --workload c3 --cpu-seconds 0 2>/dev/null
python bench.py --workload c4 --n-seq 12000 --cpu-seconds 0 2>/dev/null
python bench.py --workload c5 --n-seq 40000 --cpu-seconds 0 2>/dev/null
python bench.py --storage float16 --cpu-seconds 0 2>/dev/null
