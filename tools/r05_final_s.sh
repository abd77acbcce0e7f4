#!/bin/bash
# round 5, session s2: the two-kernel path's scratch cap (4 GB by default) against 16 / 32 GB on the c5 / c4 mixes at [5, 44]
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/ws_cap.jsonl
for ws in 4096 16384 32768; do for w in c5 c4; do
  timeout -k 10 300 python bench.py --workload $w --qdim 5,44 --opt path=1 --opt workspace_mb=$ws --cpu-seconds 0 --parity-sample 16 --steps 6 --warmup 2 $( [ $w = c4 ] && echo --n-seq 12000 ) $( [ $w = c5 ] && echo --n-seq 40000 ) 2>/dev/null | tail -1 >> gpurun_out/r05/ws_cap.jsonl || exit 1
done; done
python3 - <<'PY'
import json
for i, line in enumerate(open('gpurun_out/r05/ws_cap.jsonl')):
    l = json.loads(line); r = l['roofline']
    print([4096, 4096, 16384, 16384, 32768, 32768][i], l['config']['workload'][:8], 'ms/step', round(l['ms_per_step'], 2), 'whole GB/s', round(r['whole_path_GBps']), l['parity'])
PY
