#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# A/B builds of libdctfp.so with other build-time knobs: tools/build_variant.sh NAME -DDCTFP_WALK_MIN_WAVES=3 ...
# -> build_variants/NAME.so (git-ignored; travels with gpurun).  Use with DCTFP_LIBRARY=build_variants/NAME.so.
# Same units, same parallel build as the product (build_ext.build_library); the resource usage of every kernel goes to
# build_variants/NAME.log (-Rpass-analysis=kernel-resource-usage), the walk kernels' lines are printed.
# VARIANT_UNITS="k_walk.hip dctfp.hip" (default: these two and k_gen.hip): the units the flags are for; the rest comes from
# the default build's objects.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_variants
python3 - "$name" "$@" <<'PY' 2> build_variants/$name.log
import sys
import build_ext
name, flags = sys.argv[1], sys.argv[2:]
import os
units = os.environ.get('VARIANT_UNITS', 'k_walk.hip dctfp.hip k_gen.hip').split()
build_ext.build_library(extra_flags=tuple(flags) + ('-Rpass-analysis=kernel-resource-usage',), lib_path=f'build_variants/{name}.so',
                        flag_units=None if units == ['all'] else units)
PY
python3 - "$name" <<'PY'
import re, sys
txt = open(f'build_variants/{sys.argv[1]}.log').read()
seen = set()
for m in re.finditer(r'Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)', txt, re.S):
    n = m.group(1)
    k = re.search(r'walk_ab_kernelI(\w+?)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)', n)
    if k and n not in seen and (int(m.group(3)) > 0 or (k.group(1) == 'f' and k.group(5) == '8' and k.group(3) == '4' and k.group(7) == '0')):
        seen.add(n)
        print(f'  walk {k.group(1)} S={k.group(2)} G={k.group(3)} U={k.group(5)} fused={k.group(6)} ma={k.group(7)}: {m.group(2)} VGPRs, scratch {m.group(3)} B/lane, '
              f'{m.group(4)} waves/SIMD, LDS {m.group(5)}')
PY
