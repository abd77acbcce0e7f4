. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round 4, first GPU session: the whole GPU suite ONCE on the instrumented library (exception barrier, crash handler, runtime
# report in the pytest header), smoke, the default bench line, and the same bench under rocprofv3 (which runtime files does the
# profiler map?).
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
sha256sum dctdomain_amd/*.so | tee $O/a_sha256_on_box.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tee $O/a_gpu_tests.txt | tail -6
grep -q " passed" $O/a_gpu_tests.txt && ! grep -q "failed" $O/a_gpu_tests.txt || exit 1
head -30 $O/a_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $O/a_smoke.txt | tail -12
python bench.py > $O/a_bench_c2.json 2> $O/a_bench_c2.err && cut -c1-600 $O/a_bench_c2.json && tail -12 $O/a_bench_c2.err
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/a_prof -- python3 $R/bench.py --cpu-seconds 0 --parity-sample 0 --steps 5 > $R/$O/a_bench_prof.json 2> $R/$O/a_bench_prof.err)
echo "rocprof rc=$?"; tail -15 $O/a_bench_prof.err; cut -c1-300 $O/a_bench_prof.json
find $O/a_prof -name '*.db' -delete; find $O/a_prof -name '*kernel_trace.csv' -size +2M -delete
