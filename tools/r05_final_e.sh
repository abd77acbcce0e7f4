#!/bin/bash
# round 5, closing pass (second half of the round): whole GPU suite + smoke + sha256, the default bench line, the profile passes
cd "$(dirname "$0")/.." && . tools/env.sh
bash tools/r05_final_a.sh || exit 1
timeout -k 10 580 python bench.py > gpurun_out/r05/bench_default_line.json 2> gpurun_out/r05/bench_default_stderr.txt || { tail -20 gpurun_out/r05/bench_default_stderr.txt; exit 1; }
python3 - <<'PY'
import json
l = json.loads(open('gpurun_out/r05/bench_default_line.json').read().strip().splitlines()[-1])
print('C2', round(l['value']), round(l['roofline']['frac'], 4), l['roofline']['traffic'], l['parity'])
for k, v in l['workloads'].items():
    print(k, round(v['value']), round(v['ms_per_step'], 3), (v.get('roofline') or {}).get('frac'), v['parity'], v.get('us_per_protein'))
PY
bash tools/r05_final_b.sh c2 c3 c4 c5
