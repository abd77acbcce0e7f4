#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# usage: tools/probe_with_flags.sh "<hipcc flags>" <probe.py> [args]  -- run a probe against a library built with extra flags
set -e
flags="$1"; shift
python3 -c "import sys, build_ext; build_ext.build_library(lib_path='/tmp/libprobe.so', extra_flags=tuple(sys.argv[1].split()))" "$flags"   # (all units: build_ext.UNITS)
DCTFP_LIBRARY=/tmp/libprobe.so python "$@" 2>/dev/null
