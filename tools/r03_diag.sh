set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export LIBC_FATAL_STDERR_=1 AMD_LOG_LEVEL=1
timeout -k 10 120 python -X faulthandler -m pytest "tests/test_walk_kernel.py::test_short_jobs_through_the_walk_kernel" -x -q > gpurun_out/r03/diag1.txt 2>&1; echo "rc=$?" >> gpurun_out/r03/diag1.txt
grep -v "^  File\|Extension modules" gpurun_out/r03/diag1.txt | head -60
