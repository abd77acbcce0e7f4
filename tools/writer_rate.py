"""How fast is the single writer of a database build (at 8 GPUs the Python writer is the bound and
nothing measured it)?  Synthetic records (pfam-like lengths, 1-6 domains + whole protein, random int8 fingerprints) go
through make_db.OrderedWriter -> SQLite exactly as worker processes deliver them (flushes of 512, interleaved from 8
shards), then rename_vid / update_metadata / the .npz and .dom exports.  CPU only; nothing here touches the GPU.
usage: python tools/writer_rate.py [n_proteins]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dctdomain_amd.database import Database
from dctdomain_amd.dist import balanced_shards
from dctdomain_amd.make_db import OrderedWriter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rng = np.random.default_rng(5)
lengths = np.clip(rng.gamma(2.2, 170.0, size=n).astype(np.int64), 81, 1330)
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, 'w.fasta')
    t0 = time.perf_counter()
    with open(fa, 'w') as fh:
        for i, L in enumerate(lengths):
            fh.write(f'>P{i:07d} synthetic\n{"A" * int(L)}\n')
    db = Database(os.path.join(tmp, 'w.db'), fa)
    pending = db.pending()
    t_init = time.perf_counter() - t0
    # what the workers would send: per protein (pid, domain strings, int8 (k, 480))
    recs = {}
    n_fp = 0
    for pid, seq in pending:
        L = len(seq)
        k = max(1, min(int(round(L / 110)), 6))
        if k == 1:
            doms = [f'1-{L}']
        else:
            e = [round(i * L / k) for i in range(k + 1)]
            doms = [f'{a + 1}-{b}' for a, b in zip(e[:-1], e[1:])] + [f'1-{L}']
        recs[pid] = (pid, doms, rng.integers(0, 128, size=(len(doms), 480), dtype=np.int8))
        n_fp += len(doms)
    shards = balanced_shards([len(s) for _, s in pending], 8)
    flushes = [[recs[pending[i][0]] for i in ix[a:a + 512]] for ix in shards for a in range(0, len(ix), 512)]
    # interleave the 8 workers' flushes round-robin, as their queues would deliver them
    by_worker, pos = [], 0
    for ix in shards:
        cnt = (len(ix) + 511) // 512
        by_worker.append(flushes[pos:pos + cnt])
        pos += cnt
    order = [w[i] for i in range(max(len(w) for w in by_worker)) for w in by_worker if i < len(w)]
    writer = OrderedWriter(db, pending)
    t0 = time.perf_counter()
    held_max = 0
    for f in order:
        writer.add(f)
        held_max = max(held_max, len(writer.held))
    writer.finish()
    t_write = time.perf_counter() - t0
    t0 = time.perf_counter()
    db.rename_vid()
    db.update_metadata()
    t_meta = time.perf_counter() - t0
    t0 = time.perf_counter()
    db.save_fprints(os.path.join(tmp, 'w-dct.npz'))
    db.save_doms(os.path.join(tmp, 'w.dom'))
    t_export = time.perf_counter() - t0
    size = os.path.getsize(os.path.join(tmp, 'w.db'))
    db.close()
print(f'{n} proteins, {n_fp} fingerprints: fasta -> .db + pending {t_init:.1f} s; writer {t_write:.1f} s = '
      f'{n / t_write:,.0f} proteins/s = {n_fp / t_write:,.0f} fingerprints/s (8 interleaved shards, flush 512, at most '
      f'{held_max} records held); rename_vid + metadata {t_meta:.1f} s; .npz + .dom export {t_export:.1f} s; .db {size / 1e6:.0f} MB')
