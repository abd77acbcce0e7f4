. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round 4: stitch front end with lazy views, then BASELINE config 5 at its stated scale (1 M proteins, one worker).
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_stitch.py tests/test_make_db_gpu.py tests/test_reccut.py -q -m gpu -x 2>&1 | tail -4 || exit 1
python tools/next_rows_bench.py > $O/next_rows_kernels.json 2> $O/next_rows.err; head -12 $O/next_rows_kernels.json
bash tools/db_build_scale.sh ${1:-1000000} one > $O/db_build_1M_one.txt 2>&1; tail -25 $O/db_build_1M_one.txt
