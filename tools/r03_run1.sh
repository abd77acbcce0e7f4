set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_walk_kernel.py "tests/test_gpu_parity.py::test_kernel_variants_agree_with_golden" -x -q > gpurun_out/r03/tests_walk1.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/tests_walk1.txt
tail -5 gpurun_out/r03/tests_walk1.txt
DCTFP_LIBRARY=build_variants/timeline.so timeout -k 10 300 python tools/walk_timeline.py c2 c4 c5 > gpurun_out/r03/timeline1.txt 2>&1
cat gpurun_out/r03/timeline1.txt
DCTFP_LIBRARY=build_variants/exp.so timeout -k 10 300 python tools/path_probe.py c4 path=2 path=2,ab_unroll=12 path=2,ab_unroll=16 > gpurun_out/r03/probe_c4_unroll.txt 2>&1
cat gpurun_out/r03/probe_c4_unroll.txt
