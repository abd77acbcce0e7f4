#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# usage: tools/probe_with_flags.sh "<hipcc flags>" <probe.py> [args]  -- run a probe against a library built with extra flags
set -e
flags="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $flags -o /tmp/libprobe.so dctdomain_amd/csrc/dctfp.hip
DCTFP_LIBRARY=/tmp/libprobe.so python "$@" 2>/dev/null
