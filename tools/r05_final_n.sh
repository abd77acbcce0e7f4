#!/bin/bash
# round 5, closing pass n (walk_ab_kernel with six column groups in the library): whole GPU suite + smoke + sha256, the default bench
# line, the profile passes (the stamp), [3, 85] through bench.py
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
bash tools/r05_final_b.sh c2 c3 c4 c5 > gpurun_out/r05/profile_pass_n.txt 2>&1 || { tail -20 gpurun_out/r05/profile_pass_n.txt; exit 1; }
grep -h "frac" gpurun_out/prof_r05_c*/summary.md
for w in c2 c4 c5; do
  timeout -k 10 300 python bench.py --workload $w --qdim 3,85 --cpu-seconds 0 --steps 10 --warmup 2 $( [ $w = c4 ] && echo --n-seq 12000 ) $( [ $w = c5 ] && echo --n-seq 40000 ) 2>/dev/null | tail -1 >> gpurun_out/r05/bench_qdim_3x85.jsonl || exit 1
done
python3 - <<'PY'
import json
for line in open('gpurun_out/r05/bench_qdim_3x85.jsonl'):
    l = json.loads(line)
    print(l['config']['workload'][:40], round(l['value']), round(l['roofline']['frac'], 4), l['roofline']['kernel'], l['parity'])
PY
