#!/bin/bash
# round 5, closing pass d: parity soaks on the shipped library (host cores = the checker)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
{
  timeout -k 10 500 python tools/parity_soak_windows.py 2048 16 1280 | tail -2
  timeout -k 10 300 python tools/parity_soak_windows.py 768 16 640 | tail -1
  timeout -k 10 400 python tools/parity_soak_windows.py 512 16 2560 | tail -1
  timeout -k 10 400 python tools/parity_soak.py 8192 16 | tail -1
  timeout -k 10 400 python tools/parity_soak_mixed.py 4000 16 640 | tail -1
  timeout -k 10 400 python tools/parity_soak_mixed.py 4000 16 1280 | tail -1
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05/parity_soak_final_library.txt
