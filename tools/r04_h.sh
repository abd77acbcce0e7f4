. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round 4: roofline evidence for the general walk kernel (rocprofv3 stats + PMC of bench.py --qdim 5,44), the host profile of
# the one-protein-per-call path.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/profile_gpu.sh r04_gen_5x44 --qdim 5,44 2>&1 | tail -22
python bench.py --qdim 5,44 --cpu-seconds 0 2>/dev/null | cut -c1-1200
python tools/pcie_rate.py profile > gpurun_out/r04/one_protein_profile.txt 2>&1; head -45 gpurun_out/r04/one_protein_profile.txt
