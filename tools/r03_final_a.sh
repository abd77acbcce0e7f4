. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round-3 closing run, part A: the whole GPU suite, smoke, the bench lines of every workload.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export LIBC_FATAL_STDERR_=1
sha256sum dctdomain_amd/libdctfp.so dctdomain_amd/libreccut.so | tee gpurun_out/r03/final_sha256_on_box.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tee gpurun_out/r03/final_gpu_tests.txt | tail -6
grep -q " passed" gpurun_out/r03/final_gpu_tests.txt && ! grep -q "failed" gpurun_out/r03/final_gpu_tests.txt || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/r03/final_smoke.txt | tail -2
bash tools/workloads_run.sh > gpurun_out/r03/workloads.jsonl 2> gpurun_out/r03/workloads.err
cut -c1-400 gpurun_out/r03/workloads.jsonl
