cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export LIBC_FATAL_STDERR_=1
timeout -k 10 1100 python tools/fuzz_soak.py 36 16 4 2>&1 | tee gpurun_out/r03/fuzz_soak_seed4.txt | tail -3
