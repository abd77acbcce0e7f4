#!/bin/bash
# round 5, session a: dctfp_quantize_windows (shared window rows averaged in the walk kernel's row load) -- parity, then
# BASELINE config 3 from windows in its three forms
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests/test_windows.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r05/windows_tests.txt 2>&1 || { cat gpurun_out/r05/windows_tests.txt; exit 1; }
cat gpurun_out/r05/windows_tests.txt
for form in fused stitch stitched; do
  n=10000; [ $form = stitch ] && n=5000
  timeout -k 10 400 python bench.py --workload c3 --c3-form $form --n-seq $n --steps 10 --warmup 2 --cpu-seconds 0 --parity-sample 24 2> gpurun_out/r05/c3_$form.err | tee -a gpurun_out/r05/c3_forms.jsonl || { tail -20 gpurun_out/r05/c3_$form.err; exit 1; }
done
