#!/bin/bash
# round 5, closing pass c: the default bench line (C2 + the workloads object) on the stamped sources; the 100 000-protein build
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( time timeout -k 10 600 python bench.py > gpurun_out/r05/bench_default_line.json 2> gpurun_out/r05/bench_default_line.err ) 2>&1 | tail -4
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r05/bench_default_line.json').read().strip().splitlines()[-1])
print('C2', round(d['value']), round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4), 'traffic', d['roofline']['traffic'], d['parity'], {k: d['cpu_baseline'][k] for k in ('value', 'cores', 'os_cpu_count', 'usable_cores')})
for k, v in d.get('workloads', {}).items():
    print(k, round(v['value']), round(v['ms_per_step'], 3), round(v['roofline']['frac'], 4), 'traffic', v['roofline']['traffic'], v['parity'])
PY
timeout -k 10 900 bash tools/db_build_scale.sh 100000 one > gpurun_out/r05/db_build_100k.txt 2>&1 || { tail -20 gpurun_out/r05/db_build_100k.txt; exit 1; }
tail -22 gpurun_out/r05/db_build_100k.txt
