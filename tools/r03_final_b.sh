. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round-3 closing run, part B: rocprofv3 kernel stats + PMC passes of bench.py for C2, c4, c5; phase timeline; clocks.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
sha256sum dctdomain_amd/libdctfp.so | tee gpurun_out/r03/final_sha256_on_box_b.txt
bash tools/profile_gpu.sh r03_c2 2>&1 | tail -25
bash tools/profile_gpu.sh r03_c4 --workload c4 --n-seq 12000 2>&1 | tail -12
bash tools/profile_gpu.sh r03_c5 --workload c5 --n-seq 40000 2>&1 | tail -12
DCTFP_LIBRARY=build_variants/timeline.so timeout -k 10 300 python tools/walk_timeline.py c2 c4 c5 2>&1 | tee gpurun_out/r03/timeline_final.txt
