set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export LIBC_FATAL_STDERR_=1
timeout -k 10 300 python tools/parity_soak_mixed.py 6144 16 640 2>&1 | tee gpurun_out/r03/parity_soak_mixed_D640.txt | tail -2
timeout -k 10 300 python tools/parity_soak_mixed.py 4096 16 1280 2>&1 | tee gpurun_out/r03/parity_soak_mixed_D1280.txt | tail -2
timeout -k 10 300 python tools/parity_soak_mixed.py 2048 16 2560 2>&1 | tee gpurun_out/r03/parity_soak_mixed_D2560.txt | tail -2
timeout -k 10 400 python tools/parity_soak.py 16384 16 2>&1 | tee gpurun_out/r03/parity_soak_c2_16384.txt | tail -2
timeout -k 10 500 python tools/fuzz_soak.py 40 16 3 2>&1 | tee gpurun_out/r03/fuzz_soak_seed3.txt | tail -4
