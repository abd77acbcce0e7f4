#!/bin/bash
# round 5, session b: the whole GPU suite on the library with two-source pieces, then the default bench line (C2 + the
# "workloads" object: c3 from windows, c4, c5)
cd "$(dirname "$0")/.." && . tools/env.sh
mkdir -p gpurun_out/r05
( timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r05/gpu_suite_b.txt 2>&1 || { cat gpurun_out/r05/gpu_suite_b.txt; exit 1; }
tail -5 gpurun_out/r05/gpu_suite_b.txt
( time timeout -k 10 600 python bench.py > gpurun_out/r05/bench_default_b.json 2> gpurun_out/r05/bench_default_b.err ) 2>&1 | tail -4
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r05/bench_default_b.json').read().strip().splitlines()[-1])
print('C2', round(d['value']), round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4), d['parity'], d['cpu_baseline'])
for k, v in d.get('workloads', {}).items():
    print(k, round(v['value']), round(v['ms_per_step'], 3), round(v['roofline']['frac'], 4), v['parity'])
PY
