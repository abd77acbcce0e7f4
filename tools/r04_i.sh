#!/bin/bash
# round 4, session i: fuzz of the general walk kernel (three dispatches per call, oracle sample on the host cores)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
timeout -k 10 800 python tools/fuzz_gen.py 36 16 1 > gpurun_out/r04/fuzz_gen_seed1.txt 2>&1
echo "rc=$?" >> gpurun_out/r04/fuzz_gen_seed1.txt
tail -5 gpurun_out/r04/fuzz_gen_seed1.txt
