set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
rocm-smi --showclocks --showpower 2>&1 | head -30 > gpurun_out/r03/smi_idle.txt
timeout -k 10 200 python tools/clock_probe.py c2 c4 c5 > gpurun_out/r03/clock_probe.txt 2>&1
cat gpurun_out/r03/clock_probe.txt | cut -c1-1500
export TMPDIR=/tmp; cd /tmp
for w in c2 c4 c5; do
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/pmc_clk_$w -- python3 $GRAFT_REPO_ROOT/tools/path_probe.py $w path=2 > $GRAFT_REPO_ROOT/gpurun_out/r03/pmc_clk_$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for w in ('c2', 'c4', 'c5'):
    cnt = {}
    for f in glob.glob(f'gpurun_out/r03/pmc_clk_{w}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'walk_ab' in r['Kernel_Name']:
                cnt.setdefault(r['Counter_Name'], []).append((float(r['Counter_Value']), int(r.get('Start_Timestamp', 0) or 0), int(r.get('End_Timestamp', 0) or 0)))
    for k, v in cnt.items():
        vals = [x[0] for x in v]
        durs = [(x[2] - x[1]) for x in v if x[2] > x[1]]
        print(w, k, 'launches', len(vals), 'mean', sum(vals) / len(vals), 'mean dur ns', (sum(durs) / len(durs)) if durs else None,
              'cycles per ns', (sum(vals) / len(vals)) / (sum(durs) / len(durs)) if durs else None)
PY
find gpurun_out/r03 -name '*.db' -delete; find gpurun_out/r03 -name '*kernel_trace.csv' -size +1M -delete
