#!/bin/bash
# round 5, closing pass g: one-protein calls without the redundant event records (parity + rate), the flush parity soak against the
# reference's RecCut binary, then the profile passes on these sources (re-stamp)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fences.py tests/test_integration_stub.py tests/test_context_cache.py tests/test_make_db_gpu.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r05/quantize_one_tests2.txt 2>&1 || { cat gpurun_out/r05/quantize_one_tests2.txt; exit 1; }
tail -2 gpurun_out/r05/quantize_one_tests2.txt
timeout -k 10 300 python tools/pcie_rate.py profile > gpurun_out/r05/pcie_inclusive_rate.txt 2>&1 || { tail gpurun_out/r05/pcie_inclusive_rate.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/pcie_inclusive_rate.txt | head -3
{
timeout -k 10 900 python tools/parity_soak_flush.py 1536 16 640 || exit 1
timeout -k 10 900 python tools/parity_soak_flush.py 512 16 2560 || exit 1
} > gpurun_out/r05/parity_soak_flush.txt 2>&1 || { tail -20 gpurun_out/r05/parity_soak_flush.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/parity_soak_flush.txt
bash tools/r05_final_b.sh c2 c3 c4 c5 > gpurun_out/r05/profile_pass_g.txt 2>&1 || { tail -20 gpurun_out/r05/profile_pass_g.txt; exit 1; }
grep -h "frac" gpurun_out/prof_r05_c*/summary.md
timeout -k 10 900 bash tools/db_build_scale.sh 100000 one > gpurun_out/r05/db_build_100k_final2.txt 2>&1 || { tail -30 gpurun_out/r05/db_build_100k_final2.txt; exit 1; }
grep -E "stage fingerprint|sha256|wall" gpurun_out/r05/db_build_100k_final2.txt
