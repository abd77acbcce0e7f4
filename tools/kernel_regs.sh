#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# usage: tools/kernel_regs.sh [extra hipcc flags] -- VGPRs / occupancy / spills of the float32 n=3 stage-A variants and stage B
for u in dctfp k_walk k_gen k_stage_b k_stage_a_f32 k_stage_a_f64 k_stage_a_f16 k_stage_a_bf16; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idctdomain_amd/csrc "$@" -c dctdomain_amd/csrc/$u.hip -o /tmp/regs_$u.o -Rpass-analysis=kernel-resource-usage 2>&1; done | python3 -c "
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = m.group(1); vals = {}
    for key in ('VGPRs', 'Occupancy \[waves/SIMD\]', 'SGPRs Spill', 'VGPRs Spill', 'LDS Size \[bytes/block\]'):
        m = re.search(r' ' + key + r': (\d+)', line)
        if m and cur: vals[key[:6]] = m.group(1)
    if 'LDS Size' in line and cur:
        if 'stage_a_kernelIfLi3ELi4E' in cur or 'stage_b_mfma' in cur:
            short = re.sub(r'.*(stage_[ab]_\w*?kernelI\w*?)EEv.*', r'\1', cur)[:60]
            print(short, vals)
        cur = None
"
