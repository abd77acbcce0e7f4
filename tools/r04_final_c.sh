. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round-4 closing run, part C (after the L1 / row-select / stitch kernels): rocprofv3 kernel stats + PMC passes of bench.py for
# C2, c4, c5 and the general kernel at [5, 44] (the stamp of profiles/traffic.json covers every kernel source), kernel-level
# stats of the similarity and stitch kernels (tools/next_rows_bench.py under rocprofv3).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
sha256sum dctdomain_amd/libdctfp.so | tee gpurun_out/r04/final_sha256_on_box_c.txt
bash tools/profile_gpu.sh r04_c2 2>&1 | tail -25 &&
bash tools/profile_gpu.sh r04_c4 --workload c4 --n-seq 12000 2>&1 | tail -12 &&
bash tools/profile_gpu.sh r04_c5 --workload c5 --n-seq 40000 2>&1 | tail -12 &&
bash tools/profile_gpu.sh r04_gen_5x44 --qdim 5,44 2>&1 | tail -12 &&
( export TMPDIR=/tmp; R=$PWD; cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_next_rows -- python3 $R/tools/next_rows_bench.py > $R/gpurun_out/r04/next_rows_under_rocprof.json 2> $R/gpurun_out/r04/next_rows_under_rocprof.err )
find gpurun_out/prof_r04_next_rows -name '*.db' -delete; find gpurun_out/prof_r04_next_rows -name '*kernel_trace.csv' -size +2M -delete
find gpurun_out/prof_r04_next_rows -name '*kernel_stats.csv' | head -3
