#!/bin/bash
# usage: tools/shape_sweep.sh  -- c5 / c4 whole-path rate against stage-A launch shape and LDS pad (bench.py, no CPU leg)
for w in c5 c4; do
  n=40000; [ $w = c4 ] && n=4000
  for cfg in "a_waves=0" "a_waves=2" "a_waves=4" "a_waves=8" "a_waves=4 a_lds_pad=11264" "a_waves=4 a_lds_pad=18432" "a_waves=8 a_lds_pad=8192"; do
    opts=""; for o in $cfg; do opts="$opts --opt $o"; done
    python bench.py --workload $w --n-seq $n --cpu-seconds 0 --parity-sample 4 --steps 10 --warmup 3 $opts 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$w', '$cfg'.ljust(28), 'whole', round(r['whole_path_GBps']), 'stage A', round(r['achieved']), 'GB/s  ms/step', round(d['ms_per_step'],2), d['parity'])"
  done
done
