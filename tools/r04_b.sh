. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round 4, host-path session: GPU suite on the rewritten host side (C piece builder, device contact ordering, C stitch
# geometry, vectorised flush), then the host-time measurements VERDICT r3 #5 / "do this" #2 names.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
lscpu | grep -E "Model name|^CPU\(s\)|Flags" | cut -c1-300 > $O/b_host_cpu.txt; nproc >> $O/b_host_cpu.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tee $O/b_gpu_tests.txt | tail -15
grep -q " passed" $O/b_gpu_tests.txt && ! grep -q "failed\|error" $O/b_gpu_tests.txt || exit 1
python tools/flush_profile.py 2048 > $O/flush_profile_after.txt 2>&1; head -8 $O/flush_profile_after.txt
python tools/next_rows_bench.py > $O/next_rows_kernels.json 2> $O/next_rows.err; cat $O/next_rows_kernels.json
for w in "c2 10000" "c5 40000" "c4 12000"; do set -- $w; echo "== $1 $2"; python tools/host_time.py $1 $2; done > $O/host_time.txt 2>&1; cat $O/host_time.txt
python tools/reccut_bench.py 200 > $O/reccut_bench_on_box.txt 2>&1; tail -2 $O/reccut_bench_on_box.txt
