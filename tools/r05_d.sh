#!/bin/bash
# round 5, session d: where contact_topk1_kernel's time goes (truncated builds)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
for v in ${VARIANTS:-stop0 stop1 stop2 full}; do
  DCTFP_LIBRARY=$PWD/build_variants/topk1_$v.so timeout -k 10 200 python tools/topk_phase_probe.py 2>&1 | tail -1 | tee -a gpurun_out/r05/topk1_phase_probe.txt
done
