#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Runs tools/microbench/power_pipes.hip variant by variant and samples clocks + package power beside each (GPU box).
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/power_pipes tools/microbench/power_pipes.hip || exit 1
for mode in read valu16 valu24 mfma; do
    /tmp/power_pipes $mode 7 > /tmp/pp_$mode.txt &
    PID=$!
    sleep 3
    for i in 1 2 3; do
        rocm-smi --showclocks --showpower --json 2>/dev/null | python3 -c "
import json, sys
d = json.load(sys.stdin)['card0']
print('    sclk', d.get('sclk clock speed:'), ' power', d.get('Current Socket Graphics Package Power (W)'), 'W')"
        sleep 1
    done
    wait $PID
    cat /tmp/pp_$mode.txt
done
