#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# A/B builds of libdctfp.so with other build-time knobs: tools/build_variant.sh NAME -DDCTFP_WALK_MIN_WAVES=3 ...
# -> build_variants/NAME.so (git-ignored; travels with gpurun).  Use with DCTFP_LIBRARY=build_variants/NAME.so.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include "$@" -Rpass-analysis=kernel-resource-usage \
    -o build_variants/$name.so dctdomain_amd/csrc/dctfp.hip 2> build_variants/$name.log
python3 - "$name" <<'PY'
import re, sys
txt = open(f'build_variants/{sys.argv[1]}.log').read()
for m in re.finditer(r'Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)', txt, re.S):
    n = m.group(1)
    k = re.search(r'walk_ab_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)', n)
    if k and k.group(4) == '8' and (k.group(1), k.group(2)) in (('3', '4'), ('5', '4'), ('10', '3')):
        print(f'  walk S={k.group(1)} G={k.group(2)} U=8 fused={k.group(5)}: {m.group(2)} VGPRs, scratch {m.group(3)} B/lane, {m.group(4)} waves/SIMD')
PY
