#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# usage: tools/ab_build_run.sh "<extra hipcc flags for B>" [workload] [n_seq]   (A = no extra flags)
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -o /tmp/libA.so dctdomain_amd/csrc/dctfp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $1 -o /tmp/libB.so dctdomain_amd/csrc/dctfp.hip
python tools/ab_libs.py /tmp/libA.so /tmp/libB.so ${2:-c2} ${3:-10000} 2>/dev/null
