set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export LIBC_FATAL_STDERR_=1
timeout -k 10 400 python tools/ab_many.py dctdomain_amd/libdctfp.so build_variants/aux3.so build_variants/aux18.so build_variants/aux19.so -- c5 c4 c2 > gpurun_out/r03/ab10_aux.txt 2>&1
cat gpurun_out/r03/ab10_aux.txt
timeout -k 10 1000 bash tools/db_build_scale.sh 1000000 two_resume > gpurun_out/r03/db_build_1M_two_resume.txt 2>&1
cat gpurun_out/r03/db_build_1M_two_resume.txt
