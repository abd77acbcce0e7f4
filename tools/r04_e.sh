. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round 4: the general walk kernel.  Its own tests first (wrapped in a short timeout: a new kernel), then the parity suites,
# then the rates of the general shapes (tools/qdim_probe.py).
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_walk_general.py -q -m gpu -x 2>&1 | tee $O/e_walk_general_tests.txt | tail -30
grep -q " passed" $O/e_walk_general_tests.txt && ! grep -q "failed\|error" $O/e_walk_general_tests.txt || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_walk_kernel.py tests/test_fences.py tests/test_context_cache.py -q -m gpu -x 2>&1 | tail -5 || exit 1
timeout -k 10 300 python tools/qdim_probe.py > $O/qdim_probe.txt 2>&1; cat $O/qdim_probe.txt
