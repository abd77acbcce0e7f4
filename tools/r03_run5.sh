set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_walk_kernel.py "tests/test_gpu_parity.py::test_kernel_variants_agree_with_golden" tests/test_gpu_parity.py::test_half_precision_storage tests/test_gpu_parity.py::test_fused_groups_large_batch -x -q > gpurun_out/r03/tests_run5.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/tests_run5.txt
tail -5 gpurun_out/r03/tests_run5.txt
grep -q "pytest rc=0" gpurun_out/r03/tests_run5.txt || exit 1
timeout -k 10 600 python tools/ab_many.py build_variants/base.so dctdomain_amd/libdctfp.so build_variants/async.so -- c4 c5 c2 c3 > gpurun_out/r03/ab5.txt 2>&1
cat gpurun_out/r03/ab5.txt
DCTFP_LIBRARY=build_variants/timeline.so timeout -k 10 300 python tools/walk_timeline.py c4 c5 > gpurun_out/r03/timeline5.txt 2>&1
cat gpurun_out/r03/timeline5.txt
DCTFP_LIBRARY=build_variants/tl_async.so timeout -k 10 300 python tools/walk_timeline.py c4 c5 > gpurun_out/r03/timeline5_async.txt 2>&1
cat gpurun_out/r03/timeline5_async.txt
