. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round 4: fused builds of the general walk kernel + 2 channels per lane; at most S - 1 finishers in walk_ab_kernel.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_walk_general.py -q -m gpu -x 2>&1 | tee $O/g_walk_general_tests.txt | tail -25
grep -q " passed" $O/g_walk_general_tests.txt && ! grep -q "failed\|error" $O/g_walk_general_tests.txt || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_walk_kernel.py tests/test_fences.py tests/test_context_cache.py -q -m gpu -x 2>&1 | tail -4 || exit 1
python tools/qdim_probe.py 2>&1 | grep -v amdgpu | tee $O/g_qdim_probe.txt
python tools/gen_probe.py c5 c4 2>&1 | grep -v amdgpu | tee $O/g_gen_probe.txt
AB_ROUNDS=3 python tools/ab_many.py build_variants/base2.so dctdomain_amd/libdctfp.so -- c5 c4 c2 2>&1 | grep -v amdgpu | tee $O/ab_finishers.txt
