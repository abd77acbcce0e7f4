#!/usr/bin/env python3
"""Collects the per-workload traffic.json files that tools/profile_gpu.sh left under gpurun_out/prof_<tag>/ into
profiles/traffic.json, stamped with the sha256 of the kernel sources they were measured on (bench.py prints
roofline.traffic only while that stamp matches its own sources).
usage: tools/stamp_traffic.py c2=gpurun_out/prof_r02_c2 c4=gpurun_out/prof_r02_c4 ..."""
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_sha256 only; nothing touches the GPU)

out = {'source_sha256': bench.source_sha256(), 'measured': datetime.date.today().isoformat(),
       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py; bytes = 2 x FETCH_SIZE KiB + '
                 'WRITE_SIZE KiB (gfx950: FETCH_SIZE counts 128-B requests as 64 B; MI355X_MICROARCH.md, HBM)',
       'workloads': {}}
for arg in sys.argv[1:]:
    w, d = arg.split('=')
    with open(os.path.join(d, 'traffic.json')) as fh:
        t = json.load(fh)
    out['workloads'][w] = {'kernel': t.get('kernel', 'stage_a_kernel'), 'hbm_bytes_per_launch': t['stage_a_hbm_bytes_per_launch'],
                           'fetch_bytes': t['fetch_bytes'], 'write_bytes': t['write_bytes'], 'raw': t.get('raw', {})}
with open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w') as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({w: v['hbm_bytes_per_launch'] for w, v in out['workloads'].items()}))
