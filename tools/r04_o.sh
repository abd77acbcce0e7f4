#!/bin/bash
# round 4, session o: soaks on the closing library: general-kernel fuzz seeds 2 and 3 (150 calls x 3 dispatches each), whole-dispatch fuzz seed 8
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
sha256sum dctdomain_amd/libdctfp.so > gpurun_out/r04/soak_sha256.txt
timeout -k 10 420 python tools/fuzz_gen.py 150 16 2 > gpurun_out/r04/fuzz_gen_seed2.txt 2>&1; tail -1 gpurun_out/r04/fuzz_gen_seed2.txt
timeout -k 10 420 python tools/fuzz_gen.py 150 16 3 > gpurun_out/r04/fuzz_gen_seed3.txt 2>&1; tail -1 gpurun_out/r04/fuzz_gen_seed3.txt
timeout -k 10 300 python tools/fuzz_soak.py 25 16 8 > gpurun_out/r04/fuzz_soak_seed8.txt 2>&1; tail -1 gpurun_out/r04/fuzz_soak_seed8.txt
