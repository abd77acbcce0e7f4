set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests/test_walk_kernel.py tests/test_context_cache.py tests/test_fences.py tests/test_similarity_gpu.py "tests/test_make_db_gpu.py::test_constant_channel_is_reported_by_the_drop_in" tests/test_gpu_parity.py -x -q > gpurun_out/r03/tests_run2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/tests_run2.txt
tail -15 gpurun_out/r03/tests_run2.txt
grep -q "pytest rc=0" gpurun_out/r03/tests_run2.txt || exit 1
DCTFP_LIBRARY=build_variants/timeline.so timeout -k 10 300 python tools/walk_timeline.py c2 c4 c5 > gpurun_out/r03/timeline2.txt 2>&1
cat gpurun_out/r03/timeline2.txt
timeout -k 10 300 python tools/path_probe.py c2 c3 c4 c5 path=2 path=1 > gpurun_out/r03/probe2_main.txt 2>&1
cat gpurun_out/r03/probe2_main.txt
DCTFP_LIBRARY=build_variants/exp.so timeout -k 10 300 python tools/path_probe.py c4 path=2 path=2,ab_unroll=12 path=2,ab_unroll=16 > gpurun_out/r03/probe2_c4_unroll.txt 2>&1
cat gpurun_out/r03/probe2_c4_unroll.txt
DCTFP_LIBRARY=build_variants/d2.so timeout -k 10 300 python tools/path_probe.py c4 c5 c2 path=2 > gpurun_out/r03/probe2_d2.txt 2>&1
cat gpurun_out/r03/probe2_d2.txt
DCTFP_LIBRARY=build_variants/w3d2.so timeout -k 10 300 python tools/path_probe.py c4 c5 c2 path=2 > gpurun_out/r03/probe2_w3d2.txt 2>&1
cat gpurun_out/r03/probe2_w3d2.txt
