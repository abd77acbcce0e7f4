"""The single writer of a database build (dctdomain_amd.make_db.OrderedWriter) and the worker supervision
(_run_workers), without a GPU: commit per flush in pending order (the .db is the checkpoint during a run, as in the
reference: src/database.py:211-224 commits per protein and :150 resumes on fpcount = 0), and a worker that fails or
dies ends the build instead of hanging it."""

import os
import sqlite3
import time

import numpy as np
import pytest

from dctdomain_amd import make_db
from dctdomain_amd.database import Database


def _fasta(path, n):
    rng = np.random.default_rng(11)
    with open(path, 'w') as fh:
        for i in range(n):
            L = int(rng.integers(30, 90))
            fh.write(f'>p{i:03d} x\n' + ''.join('ACDEFGHIKLMNPQRSTVWY'[int(v)] for v in rng.integers(0, 20, L)) + '\n')


def _record(pid, seq):
    rng = np.random.default_rng(abs(hash(pid)) % (1 << 31))
    doms = [f'1-{len(seq) // 2}', f'{len(seq) // 2 + 1}-{len(seq)}', f'1-{len(seq)}'] if len(seq) % 2 else [f'1-{len(seq)}']
    return pid, doms, rng.integers(0, 128, size=(len(doms), 480)).astype(np.int8)


def _files(db, base):
    db.rename_vid()
    db.save_fprints(base + '-dct.npz')
    db.save_doms(base + '.dom')
    z = np.load(base + '-dct.npz')
    return {k: z[k] for k in z.files}, open(base + '.dom').read()


def test_writer_commits_each_flush_in_pending_order(tmp_path):
    fa = str(tmp_path / 'x.fasta')
    _fasta(fa, 23)
    # reference build: everything in one add_fprints call
    ref = Database(str(tmp_path / 'ref'), fa)
    pend = ref.pending()
    ref.add_fprints([make_db._Rec(*_record(p, s)) for p, s in pend])
    ref_npz, ref_dom = _files(ref, str(tmp_path / 'ref'))
    ref.close()

    db = Database(str(tmp_path / 'inc'), fa)
    pend = db.pending()
    assert [p for p, _ in pend] == [p for p, _ in sorted(pend, key=lambda x: len(x[1]))]      # ascending length
    w = make_db.OrderedWriter(db, pend)
    other = sqlite3.connect(str(tmp_path / 'inc.db'))        # what a second process (or a restart) would see

    def n_pending():
        return other.execute('SELECT COUNT(*) FROM sequences WHERE fpcount = 0').fetchone()[0]

    recs = [_record(p, s) for p, s in pend]
    # two interleaved shards (what two GPU workers send), flushes of 3
    shard = [recs[0::2], recs[1::2]]
    w.add(shard[0][:3])                       # pending indices 0, 2, 4: only index 0 is contiguous
    assert w.written == 1 and n_pending() == 22 and len(w.held) == 2
    w.add(shard[1][:3])                       # 1, 3, 5 -> 0..5 complete
    assert w.written == 6 and n_pending() == 17 and not w.held
    # a crash here loses nothing that was committed: a new Database on the file resumes after protein 5
    again = Database(str(tmp_path / 'inc.db'))
    assert [p for p, _ in again.pending()] == [p for p, _ in pend[6:]]
    again.close()
    w.add(shard[1][3:])
    w.add(shard[0][3:])
    w.finish()
    assert w.written == 23 and n_pending() == 0
    got_npz, got_dom = _files(db, str(tmp_path / 'inc'))
    for k in ref_npz:
        np.testing.assert_array_equal(ref_npz[k], got_npz[k])
    assert got_dom == ref_dom
    vids = [v for v, in other.execute('SELECT vid FROM fingerprints')]
    assert vids == list(range(1, len(vids) + 1))
    other.close()
    db.close()


def test_interrupted_build_resumes_to_identical_files(tmp_path):
    fa = str(tmp_path / 'x.fasta')
    _fasta(fa, 17)
    full = Database(str(tmp_path / 'full'), fa)
    pend = full.pending()
    w = make_db.OrderedWriter(full, pend)
    for i in range(0, len(pend), 4):
        w.add([_record(p, s) for p, s in pend[i:i + 4]])
    w.finish()
    full_npz, full_dom = _files(full, str(tmp_path / 'full'))
    full.close()

    part = Database(str(tmp_path / 'part'), fa)
    w = make_db.OrderedWriter(part, part.pending())
    w.add([_record(p, s) for p, s in pend[:4]])
    w.add([_record(p, s) for p, s in pend[4:8]])
    part.conn.close()                                   # "killed" after two flushes
    part = Database(str(tmp_path / 'part.db'))          # restart: same command on the existing database
    rest = part.pending()
    assert [p for p, _ in rest] == [p for p, _ in pend[8:]]
    w = make_db.OrderedWriter(part, rest)
    for i in range(0, len(rest), 4):
        w.add([_record(p, s) for p, s in rest[i:i + 4]])
    w.finish()
    got_npz, got_dom = _files(part, str(tmp_path / 'part'))
    for k in full_npz:
        np.testing.assert_array_equal(full_npz[k], got_npz[k])
    assert got_dom == full_dom
    part.close()


# -- worker supervision ------------------------------------------------------------------------------------
def _worker_ok(rank, n_gpu, payload, out_q):
    out_q.put(('recs', rank, [(f'r{rank}', ['1-5'], None)] * payload))
    out_q.put(('done', rank))


def _worker_raises_before_anything(rank, n_gpu, payload, out_q):
    # what `--model esm` without fair-esm did in round 1: fails in load_model, before any result
    make_db._gpu_worker(rank, n_gpu, [[], []], 'esm', 500, 1, 8, out_q)


def _worker_dies_silently(rank, n_gpu, payload, out_q):
    if rank == 1:
        os._exit(7)                     # what a GPU fault / abort / OOM kill looks like: no message at all
    time.sleep(60)                      # the healthy worker would run for a long time
    out_q.put(('done', rank))


def test_workers_all_fine():
    got = []
    make_db._run_workers(2, (3,), got.extend, target=_worker_ok, poll_s=0.5)
    assert len(got) == 6


def test_worker_error_is_reported_not_hung():
    t0 = time.time()
    with pytest.raises(RuntimeError, match='GPU worker [01] failed'):
        make_db._run_workers(2, (0,), lambda r: None, target=_worker_raises_before_anything, poll_s=0.5)
    assert time.time() - t0 < 120


def test_dead_worker_is_noticed_and_the_others_are_stopped():
    t0 = time.time()
    with pytest.raises(RuntimeError, match=r'GPU worker 1 died \(exit code 7\)'):
        make_db._run_workers(2, (0,), lambda r: None, target=_worker_dies_silently, poll_s=0.5)
    assert time.time() - t0 < 45          # did not wait for the sleeping worker
