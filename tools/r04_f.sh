. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round 4: where does the float16 step lose 0.87 ms that the kernel does not take?  Kernel + copy timeline of the timed loop.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --storage float16 --cpu-seconds 0 --parity-sample 0 --steps 20 2>/dev/null | cut -c1-900 > $O/f_fp16_line.txt; cat $O/f_fp16_line.txt
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/f_prof -- python3 $R/bench.py --storage float16 --cpu-seconds 0 --parity-sample 0 --steps 8 --warmup 2 > /dev/null 2>&1)
python tools/timeline.py $O/f_prof 24 | tee $O/f_fp16_timeline.txt
f=$(find $O/f_prof -name '*memory_copy_trace.csv' | head -1); tail -12 $f | cut -c1-200
find $O/f_prof -name '*.db' -delete
