#!/bin/bash
# round 5, last closing pass: whole GPU suite + smoke + sha256, default bench line, profile passes (re-stamp), per-call rate, then the
# 1 000 000-protein build -- all on the final library
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
bash tools/r05_final_b.sh c2 c3 c4 c5 > gpurun_out/r05/profile_pass_i.txt 2>&1 || { tail -20 gpurun_out/r05/profile_pass_i.txt; exit 1; }
grep -h "frac" gpurun_out/prof_r05_c*/summary.md
timeout -k 10 200 python tools/pcie_rate.py profile > gpurun_out/r05/pcie_inclusive_rate.txt 2>&1 || { tail gpurun_out/r05/pcie_inclusive_rate.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/pcie_inclusive_rate.txt | head -2
timeout -k 10 900 bash tools/db_build_scale.sh 1000000 one > gpurun_out/r05/db_build_1M_final.txt 2>&1 || { tail -30 gpurun_out/r05/db_build_1M_final.txt; exit 1; }
grep -E "stage fingerprint|sha256|wall|resident" gpurun_out/r05/db_build_1M_final.txt
