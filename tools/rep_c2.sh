#!/bin/bash
for rep in 1 2 3; do for cfg in "4 8" "8 4" "16 4"; do set -- $cfg
  python bench.py --cpu-seconds 0 --parity-sample 0 --steps 20 --warmup 5 --opt a_waves=$1 --opt a_unroll=$2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rep $rep waves=$1 unroll=$2', round(d['value']), 'fp/s stageA', round(d['roofline']['achieved']), 'GB/s ms/step', round(d['ms_per_step'],3))"
done; done
