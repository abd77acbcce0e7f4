#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# usage: tools/ab_build_run.sh "<extra hipcc flags for B>" [workload] [n_seq]   (A = no extra flags)
set -e
python3 -c "import build_ext; build_ext.build_library(lib_path='/tmp/libA.so', extra_flags=('-DDCTFP_AB_BASE',))"   # (all units of the library: build_ext.UNITS)
python3 -c "import sys, build_ext; build_ext.build_library(lib_path='/tmp/libB.so', extra_flags=tuple(sys.argv[1].split()))" "$1"
python tools/ab_libs.py /tmp/libA.so /tmp/libB.so ${2:-c2} ${3:-10000} 2>/dev/null
