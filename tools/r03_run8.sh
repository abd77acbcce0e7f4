set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export LIBC_FATAL_STDERR_=1
timeout -k 10 600 python -m pytest tests/test_walk_kernel.py "tests/test_gpu_parity.py::test_kernel_variants_agree_with_golden" tests/test_gpu_parity.py::test_half_precision_storage tests/test_gpu_parity.py::test_fused_groups_large_batch tests/test_fences.py -x -q > gpurun_out/r03/tests_run8.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/tests_run8.txt
tail -5 gpurun_out/r03/tests_run8.txt
grep -q "pytest rc=0" gpurun_out/r03/tests_run8.txt || exit 1
timeout -k 10 600 python tools/ab_many.py build_variants/base.so dctdomain_amd/libdctfp.so -- c5 c4 c2 > gpurun_out/r03/ab8.txt 2>&1
cat gpurun_out/r03/ab8.txt
timeout -k 10 200 python tools/clock_probe.py c5 > gpurun_out/r03/clock_probe8.txt 2>&1
cat gpurun_out/r03/clock_probe8.txt | cut -c1-600
timeout -k 10 900 bash tools/db_build_scale.sh 100000 two_resume > gpurun_out/r03/db_build_100k_two_resume.txt 2>&1
cat gpurun_out/r03/db_build_100k_two_resume.txt
