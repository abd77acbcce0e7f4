#!/bin/bash
# round 5, closing pass o: parity soaks on the final library (host cores = the checker): config 3 from windows, headline shape,
# multi-domain lists at [3, 80] and at [3, 85] (walk_ab_kernel with six column groups), the flush against the reference's RecCut binary
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
{
  timeout -k 10 500 python tools/parity_soak_windows.py 2048 16 1280 | tail -2
  timeout -k 10 300 python tools/parity_soak_windows.py 768 16 640 | tail -1
  timeout -k 10 400 python tools/parity_soak_windows.py 512 16 2560 | tail -1
  timeout -k 10 400 python tools/parity_soak.py 8192 16 | tail -1
  timeout -k 10 400 python tools/parity_soak_mixed.py 4000 16 640 | tail -1
  timeout -k 10 400 python tools/parity_soak_mixed.py 4000 16 1280 | tail -1
  timeout -k 10 400 python tools/parity_soak_mixed.py 3000 16 640 85 | tail -1
  timeout -k 10 400 python tools/parity_soak_mixed.py 3000 16 1280 85 | tail -1
  timeout -k 10 400 python tools/parity_soak_mixed.py 1200 16 2560 85 | tail -1
  timeout -k 10 600 python tools/parity_soak_flush.py 2048 16 640 | tail -1
  timeout -k 10 600 python tools/parity_soak_flush.py 768 16 2560 | tail -1
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05/parity_soak_final_library.txt
