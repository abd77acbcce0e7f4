set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export LIBC_FATAL_STDERR_=1
timeout -k 10 1150 bash tools/db_build_scale.sh 1000000 one 2>&1 | tee gpurun_out/r03/db_build_1M_one.txt
