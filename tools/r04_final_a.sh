. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Round-4 closing run, part A: the whole GPU suite (header: which runtime files), smoke, the bench line of every workload,
# parity soaks of the final library against the scipy oracle on the host cores.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
sha256sum dctdomain_amd/*.so | tee $O/final_sha256_on_box.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tee $O/final_gpu_tests.txt | tail -8
grep -q " passed" $O/final_gpu_tests.txt && ! grep -q "failed" $O/final_gpu_tests.txt || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $O/final_smoke.txt | tail -3
bash tools/workloads_run.sh > $O/workloads.jsonl 2> $O/workloads.err
python - <<'PY'
import json
for l in open('gpurun_out/r04/workloads.jsonl'):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['config']['workload'][:40], round(d['value']), 'fp/s', round(d['ms_per_step'], 3), 'ms  kernel', round(d['roofline']['achieved']), 'GB/s  whole path', round(d['roofline']['whole_path_GBps']), d['parity'], d['host_table_ms'])
PY
timeout -k 10 400 python tools/parity_soak.py 8192 16 > $O/parity_soak_c2_8192.txt 2>&1; tail -2 $O/parity_soak_c2_8192.txt
for D in 640 1280 2560; do timeout -k 10 500 python tools/parity_soak_mixed.py $([ $D = 2560 ] && echo 2000 || echo 6000) 16 $D > $O/parity_soak_mixed_D$D.txt 2>&1; tail -1 $O/parity_soak_mixed_D$D.txt; done
timeout -k 10 200 python tools/pcie_rate.py > $O/pcie_inclusive_rate.txt 2>&1; tail -4 $O/pcie_inclusive_rate.txt
