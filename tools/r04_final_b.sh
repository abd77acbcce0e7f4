. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round-4 closing run, part B: rocprofv3 kernel stats + PMC passes of bench.py for C2, c4, c5; the phase timeline and the
# per-wave waits of the ticket flush (instrumented build).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
sha256sum dctdomain_amd/libdctfp.so | tee gpurun_out/r04/final_sha256_on_box_b.txt
bash tools/profile_gpu.sh r04_c2 2>&1 | tail -25
bash tools/profile_gpu.sh r04_c4 --workload c4 --n-seq 12000 2>&1 | tail -12
bash tools/profile_gpu.sh r04_c5 --workload c5 --n-seq 40000 2>&1 | tail -12
DCTFP_LIBRARY=build_variants/timeline.so timeout -k 10 300 python tools/walk_timeline.py c2 c4 c5 2>&1 | tee gpurun_out/r04/timeline_final.txt
DCTFP_LIBRARY=build_variants/timeline.so timeout -k 10 300 python tools/wave_wait_probe.py 2>&1 | tee gpurun_out/r04/wave_wait_probe_final.txt | tail -30
