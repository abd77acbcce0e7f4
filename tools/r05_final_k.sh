#!/bin/bash
# round 5, closing pass k: the stage-A launch ladder in one include (four units of 15 lines): whole GPU suite + smoke once more, then the
# profile passes on these sources (the stamp)
cd "$(dirname "$0")/.." && . tools/env.sh
bash tools/r05_final_a.sh || exit 1
bash tools/r05_final_b.sh c2 c3 c4 c5 > gpurun_out/r05/profile_pass_k.txt 2>&1 || { tail -20 gpurun_out/r05/profile_pass_k.txt; exit 1; }
grep -h "frac" gpurun_out/prof_r05_c*/summary.md
