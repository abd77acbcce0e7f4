#!/bin/bash
. "$(dirname "${BASH_SOURCE[0]}")/env.sh"   # LIBC_FATAL_STDERR_, PYTHONFAULTHANDLER, DCTFP_CRASH_BACKTRACE
# Scaled-down rehearsal of BASELINE config 5 (database build through the make_db drop-in) on one GPU:
# N synthetic proteins with a pfam-like length mix -> .db / -dct.npz / .dom, synthetic language model.
N=${1:-20000}
OUT=/tmp/dbr
rm -rf $OUT; mkdir -p $OUT
python - <<PY
import numpy as np
rng = np.random.default_rng(7)
lens = np.clip(rng.gamma(2.2, 170.0, size=$N).astype(int), 81, 1330)
with open('$OUT/x.fasta', 'w') as f:
    for i, L in enumerate(lens):
        f.write(f'>sp{i:07d}\n' + ''.join('ACDEFGHIKLMNPQRSTVWY'[v] for v in rng.integers(0, 20, size=L)) + '\n')
print('residues', int(lens.sum()))
PY
START=$(python -c "import time; print(time.time())")
python -m dctdomain_amd.make_db --fafile $OUT/x.fasta --dbfile $OUT/x --model synthetic --cpu 16 --flush 2048 --noindex > $OUT/log.txt 2> $OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
END=$(python -c "import time; print(time.time())")
tail -4 $OUT/log.txt
python -c "print('make_db wall seconds: %.1f  (%.0f proteins/s)' % ($END - $START, $N / ($END - $START)))"
python - <<PY
import numpy as np, os
z = np.load('$OUT/x-dct.npz')
print('proteins', len(z['sid']), 'fingerprints', z['dct'].shape, 'db MB', round(os.path.getsize('$OUT/x.db') / 1e6, 1), 'npz MB', round(os.path.getsize('$OUT/x-dct.npz') / 1e6, 1))
rows = z['dct'].reshape(-1, 6, 80)
print('invariants ok:', bool(((rows == 127).sum(axis=2) == 1).all() and ((rows == 0).sum(axis=2) >= 1).all()))
PY
