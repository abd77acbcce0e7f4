#!/bin/bash
# stage-A launch-shape sweep on the GPU box: prints fingerprints/s and stage-A GB/s per variant
for w in 4 8 16; do for u in 4 8; do
  echo -n "waves=$w unroll=$u: "
  python bench.py --cpu-seconds 0 --parity-sample 0 --steps 10 --warmup 2 --opt a_waves=$w --opt a_unroll=$u | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'fp/s  stageA', round(d['roofline']['achieved']), 'GB/s  ms/step', round(d['ms_per_step'],3), ' stageB ms', round(d['roofline']['stage_b_avg_launch_ms'],3))"
done; done
