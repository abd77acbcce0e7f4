#!/bin/bash
# round 5, closing pass p: the flush soak again, with discontinuous domains among the reference binary's answers
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
{
  timeout -k 10 600 python tools/parity_soak_flush.py 3072 16 640 | tail -1
  timeout -k 10 600 python tools/parity_soak_flush.py 768 16 1280 | tail -1
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05/parity_soak_flush_discontinuous.txt
