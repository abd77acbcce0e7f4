#!/bin/bash
# round 5, session z: the default bench line with the c4_pipeline workload (config 4 as stated: contact map -> RecCut -> fingerprints)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( time timeout -k 10 580 python bench.py > gpurun_out/r05/bench_default_line.json 2> gpurun_out/r05/bench_default_stderr.txt ) 2>&1 | tail -3
tail -3 gpurun_out/r05/bench_default_stderr.txt
python3 - <<'PY'
import json
l = json.loads(open('gpurun_out/r05/bench_default_line.json').read().strip().splitlines()[-1])
print('C2', round(l['value']), round(l['roofline']['frac'], 4), l['roofline']['traffic'], l['parity'])
for k, v in l['workloads'].items():
    print(k, round(v['value']), v['ms_per_step'], (v.get('roofline') or {}).get('frac'), v['parity'], v.get('gpu_ms'), v.get('host_ms'), v.get('us_per_protein'))
PY
