#!/bin/bash
# round 4, session p: more soaks on the closing library (spare GPU minutes): general-kernel fuzz seeds 4, 5; multi-domain soaks at D = 2560 and 640 with other seeds
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r04
sha256sum dctdomain_amd/libdctfp.so > gpurun_out/r04/soak2_sha256.txt
timeout -k 10 330 python tools/fuzz_gen.py 120 16 4 > gpurun_out/r04/fuzz_gen_seed4.txt 2>&1; tail -1 gpurun_out/r04/fuzz_gen_seed4.txt
timeout -k 10 330 python tools/fuzz_gen.py 120 16 5 > gpurun_out/r04/fuzz_gen_seed5.txt 2>&1; tail -1 gpurun_out/r04/fuzz_gen_seed5.txt
timeout -k 10 250 python tools/fuzz_soak.py 20 16 9 > gpurun_out/r04/fuzz_soak_seed9.txt 2>&1; tail -1 gpurun_out/r04/fuzz_soak_seed9.txt
