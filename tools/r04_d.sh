. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round 4, walk-kernel experiments: A/B of build variants in one process on one allocation (tools/ab_many.py), then the parity
# suites on the candidate build.  Usage: tools/r04_d.sh OUTNAME CANDIDATE "spec spec ..." [workloads]
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
NAME=$1; CAND=$2; SPECS=$3; WL=${4:-c4 c5 c2}
AB_ROUNDS=3 python tools/ab_many.py $SPECS -- $WL > $O/ab_$NAME.txt 2>&1; cat $O/ab_$NAME.txt | grep -v amdgpu.ids
if [ -n "$CAND" ]; then
  DCTFP_LIBRARY=$GRAFT_REPO_ROOT/build_variants/$CAND.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_walk_kernel.py tests/test_fences.py -q -m gpu -x 2>&1 | tail -4
fi
