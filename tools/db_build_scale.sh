#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# BASELINE config 5 on one GPU box: N synthetic proteins with a pfam-like length mix through the make_db drop-in
# (synthetic language model), with the evidence the judge asked for: wall time per stage, peak RSS, file sizes, content
# hashes -- and, in mode `two_resume`, a deliberate kill at ~50 % followed by a resumed run.
#   tools/db_build_scale.sh N one          one worker, uninterrupted
#   tools/db_build_scale.sh N two_resume   two workers (both on the one GPU here), killed at ~half, resumed
# The hashes of `one` and `two_resume` must agree (profiles/r03/db_build_1M.txt).
N=${1:-200000}; MODE=${2:-one}
OUT=/tmp/dbs_$MODE
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s.%N)
python - <<PY
import numpy as np
rng = np.random.default_rng(7)
lens = np.clip(rng.gamma(2.2, 170.0, size=$N).astype(int), 81, 1330)
aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
with open('$OUT/x.fasta', 'wb') as f:
    for i, L in enumerate(lens):
        f.write(b'>sp%07d\n' % i + aa[rng.integers(0, 20, size=L)].tobytes() + b'\n')
print('proteins', $N, 'residues', int(lens.sum()))
PY
T1=$(date +%s.%N)
echo "fasta written in $(python -c "print(round($T1 - $T0, 1))") s, $(du -m $OUT/x.fasta | cut -f1) MB"
GPUFLAG=""; [ "$MODE" = "two_resume" ] && GPUFLAG="--gpu 2"
CMD="python -m dctdomain_amd.make_db --fafile $OUT/x.fasta --dbfile $OUT/x --model synthetic --cpu 16 --flush 2048 --noindex $GPUFLAG"
count() { python - <<PY
import sqlite3
try:
    c = sqlite3.connect('file:$OUT/x.db?mode=ro', uri=True, timeout=30)
    print(c.execute('SELECT COUNT(*) FROM sequences WHERE fpcount > 0').fetchone()[0])
except Exception as e:
    print(0)
PY
}
if [ "$MODE" = "two_resume" ]; then
    setsid python tools/run_with_rss.py $CMD --out $OUT/log1.txt > $OUT/stdout1.txt 2> $OUT/time1.txt &
    PID=$!
    while kill -0 $PID 2>/dev/null; do
        sleep 5
        DONE=$(count)
        echo "  ... $DONE proteins committed after $(python -c "import time; print(round(time.time() - $T1, 0))") s"
        if [ "$DONE" -ge $((N * 48 / 100)) ]; then
            echo "KILLING the build (SIGKILL to the whole process group) at $DONE of $N committed proteins"
            kill -KILL -- -$PID; break
        fi
    done
    wait $PID 2>/dev/null
    T2=$(date +%s.%N)
    echo "first run ended after $(python -c "print(round($T2 - $T1, 1))") s with $(count) proteins committed; resuming"
    python tools/run_with_rss.py $CMD --out $OUT/log2.txt > $OUT/stdout2.txt 2> $OUT/time2.txt &
    PID=$!
    while kill -0 $PID 2>/dev/null; do
        sleep 30
        echo "  ... $(count) proteins committed (resumed run)"
    done
    wait $PID || { tail -20 $OUT/time2.txt; exit 1; }
    T3=$(date +%s.%N)
    echo "resumed run: $(python -c "print(round($T3 - $T2, 1))") s wall"
    grep -E "^stage " $OUT/log2.txt; grep -E "Maximum resident|Elapsed" $OUT/time2.txt
else
    python tools/run_with_rss.py $CMD --out $OUT/log1.txt > $OUT/stdout1.txt 2> $OUT/time1.txt &
    PID=$!
    while kill -0 $PID 2>/dev/null; do       # (progress lines: a silent run of minutes is taken to be hung by the GPU pool)
        sleep 30
        echo "  ... $(count) proteins committed after $(python -c "import time; print(round(time.time() - $T1, 0))") s"
    done
    wait $PID || { tail -20 $OUT/time1.txt; exit 1; }
    T3=$(date +%s.%N)
    echo "uninterrupted run: $(python -c "print(round($T3 - $T1, 1))") s wall = $(python -c "print(round($N / ($T3 - $T1)))") proteins/s"
    grep -E "^stage " $OUT/log1.txt; grep -E "Maximum resident|Elapsed" $OUT/time1.txt
fi
grep -c "constant channel" $OUT/log*.txt
python - <<PY
import hashlib, os, sqlite3
import numpy as np
z = np.load('$OUT/x-dct.npz')
print('proteins', len(z['sid']), 'fingerprints', z['dct'].shape, ' .db MB', round(os.path.getsize('$OUT/x.db') / 1e6, 1),
      ' -dct.npz MB', round(os.path.getsize('$OUT/x-dct.npz') / 1e6, 1), ' .dom MB', round(os.path.getsize('$OUT/x.dom') / 1e6, 1))
rows = z['dct'].reshape(-1, 6, 80)
print('every row has exactly one 127 and at least one 0:', bool(((rows == 127).sum(axis=2) == 1).all() and ((rows == 0).sum(axis=2) >= 1).all()))
for name in ('x-dct.npz', 'x.dom'):
    h = hashlib.sha256()
    with open(os.path.join('$OUT', name), 'rb') as f:
        for blk in iter(lambda: f.read(1 << 24), b''):
            h.update(blk)
    print('sha256', name, h.hexdigest())
c = sqlite3.connect('$OUT/x.db')
h = hashlib.sha256()
n = 0
for vid, dom, blob, pid in c.execute('SELECT vid, domain, fingerprint, pid FROM fingerprints ORDER BY vid'):
    h.update(f'{vid} {dom} {pid} '.encode()); h.update(blob); n += 1
print('sha256 fingerprints table (vid, domain, pid, blob in vid order)', h.hexdigest(), 'rows', n)
print('pending after the build:', c.execute('SELECT COUNT(*) FROM sequences WHERE fpcount = 0').fetchone()[0],
      ' metadata:', c.execute('SELECT seq_num, fp_num, seqs_fp FROM metadata ORDER BY datetime DESC LIMIT 1').fetchone())
PY
