set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export LIBC_FATAL_STDERR_=1
timeout -k 10 900 bash tools/db_build_scale.sh 100000 two_resume > gpurun_out/r03/db_build_100k_two_resume.txt 2>&1
cat gpurun_out/r03/db_build_100k_two_resume.txt
timeout -k 10 900 bash tools/db_build_scale.sh 100000 one > gpurun_out/r03/db_build_100k_one.txt 2>&1
cat gpurun_out/r03/db_build_100k_one.txt
