set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_walk_kernel.py "tests/test_gpu_parity.py::test_kernel_variants_agree_with_golden" -x -q > gpurun_out/r03/tests_run4.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/tests_run4.txt
tail -5 gpurun_out/r03/tests_run4.txt
grep -q "pytest rc=0" gpurun_out/r03/tests_run4.txt || exit 1
timeout -k 10 600 python tools/ab_many.py build_variants/base.so dctdomain_amd/libdctfp.so@ab_narrow=2 build_variants/d2.so@ab_narrow=2 build_variants/exp.so@ab_narrow=2,ab_unroll=12 build_variants/exp.so@ab_narrow=2,ab_unroll=16 -- c4 c5 c2 > gpurun_out/r03/ab4.txt 2>&1
cat gpurun_out/r03/ab4.txt
DCTFP_LIBRARY=build_variants/timeline.so timeout -k 10 300 python tools/walk_timeline.py c4 c5 path=2,ab_narrow=2 > gpurun_out/r03/timeline4.txt 2>&1
cat gpurun_out/r03/timeline4.txt
DCTFP_LIBRARY=build_variants/tl_d2.so timeout -k 10 300 python tools/walk_timeline.py c4 c5 path=2,ab_narrow=2 > gpurun_out/r03/timeline4_d2.txt 2>&1
cat gpurun_out/r03/timeline4_d2.txt
timeout -k 10 300 python tools/qdim_probe.py > gpurun_out/r03/qdim_probe1.txt 2>&1
cat gpurun_out/r03/qdim_probe1.txt
