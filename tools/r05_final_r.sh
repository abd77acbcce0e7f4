#!/bin/bash
# round 5, session r2: where the two-kernel path spends its time on the c5 / c4 mixes at [5, 44] against [3, 80] (stage A / stage B events)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/two_kernel_split.jsonl
for q in 5,44 4,80 3,80; do for w in c5 c4; do
  timeout -k 10 300 python bench.py --workload $w --qdim $q --opt path=1 --cpu-seconds 0 --parity-sample 0 --steps 6 --warmup 2 $( [ $w = c4 ] && echo --n-seq 12000 ) $( [ $w = c5 ] && echo --n-seq 40000 ) 2>/dev/null | tail -1 >> gpurun_out/r05/two_kernel_split.jsonl || exit 1
done; done
python3 - <<'PY'
import json
for line in open('gpurun_out/r05/two_kernel_split.jsonl'):
    l = json.loads(line); r = l['roofline']
    print(l['config']['workload'][:14], l['config']['workload'].split('qdim')[-1][:12] if 'qdim' in l['config']['workload'] else '', 'ms/step', round(l['ms_per_step'], 2), 'stage A launch ms', round(r['avg_launch_ms'], 3), 'stage B launch ms', round(r['stage_b_avg_launch_ms'], 3), 'kernel', r['kernel'], 'whole GB/s', round(r['whole_path_GBps']))
PY
