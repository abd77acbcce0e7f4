"""Output writers: .db / -dct.npz / .dom layout against the reference's own example
outputs (tests/golden/ref_fixtures = reference test/test/example-dct.npz, example.dom and
test/example.fasta, data files only)."""

import os
import sqlite3
from io import BytesIO
from types import SimpleNamespace

import numpy as np
import pytest

import golden_util as gu

FIX = os.path.join(gu.GOLD, 'ref_fixtures')


def ref_fingerprints():
    z = np.load(os.path.join(FIX, 'example-dct.npz'))
    out = []
    for i, pid in enumerate(z['sid']):
        s, e = z['idx'][i], z['idx'][i + 1]
        doms = [str(d) for d in z['dom'][s:e]]
        out.append(SimpleNamespace(pid=str(pid), domains=doms,
                                   quants={d: z['dct'][s + k].astype(np.int64) for k, d in enumerate(doms)}))
    return z, out


def test_layout_matches_reference_example(tmp_path, capsys):
    from dctdomain_amd.database import Database
    z, fps = ref_fingerprints()
    db = Database(str(tmp_path / 'ex'), os.path.join(FIX, 'example.fasta'))
    rows = db.cur.execute('SELECT pid, length, fpcount FROM sequences').fetchall()
    assert [r[0] for r in rows] == [str(s) for s in z['sid']]          # ascending length = reference order
    assert [r[1] for r in rows] == sorted(r[1] for r in rows) and all(r[2] == 0 for r in rows)
    pend = db.pending()
    assert [p[0] for p in pend] == [str(s) for s in z['sid']]
    # feed the reference's fingerprints through the writer in two batches
    db.add_fprints(fps[:3])
    assert [p[0] for p in db.pending()] == [f.pid for f in fps[3:]]    # resume: done proteins are skipped
    for f in fps[3:]:
        db.add_fprint(f)
    db.rename_vid()
    vids = [v[0] for v in db.cur.execute('SELECT vid FROM fingerprints').fetchall()]
    assert vids == list(range(1, 44))
    blob = db.cur.execute('SELECT fingerprint FROM fingerprints WHERE vid = 1').fetchone()[0]
    assert len(blob) == 608 and blob[:6] == b'\x93NUMPY'
    assert b"'descr': '|i1', 'fortran_order': False, 'shape': (480,), }" in blob[:128]
    db.update_metadata()
    meta = db.cur.execute('SELECT seq_num, fp_num, seqs_fp FROM metadata').fetchone()
    assert meta == (8, 43, '8/8')
    npz = str(tmp_path / 'ex-dct.npz')
    db.save_fprints(npz)
    got = np.load(npz)
    assert sorted(got.files) == sorted(z.files)
    for k in z.files:
        assert got[k].dtype == z[k].dtype and got[k].shape == z[k].shape, k
        np.testing.assert_array_equal(got[k], z[k])
    dom = str(tmp_path / 'ex.dom')
    db.save_doms(dom)
    assert open(dom).read() == open(os.path.join(FIX, 'example.dom')).read()
    # flat index writer round trip (faiss itself is not installed here: parity unpinned)
    db.create_index()
    from dctdomain_amd.database import read_flat_index
    idx = read_flat_index(str(tmp_path / 'ex.index'))
    np.testing.assert_array_equal(idx, z['dct'].astype(np.float32))
    assert db.load_fprints('P53875')[0][0] == 1
    db.seq_info('P53875')
    db.close()
    # reopening without a fasta works on the .db
    db2 = Database(str(tmp_path / 'ex.db'))
    assert db2.get_last_vid() == 44
    db2.close()


def test_yield_seqs_batches(tmp_path):
    from dctdomain_amd.database import Database
    fa = tmp_path / 's.fasta'
    lens = [5, 100, 120, 150, 200, 300, 900]
    with open(fa, 'w') as f:
        for i, L in enumerate(lens):
            f.write(f'>p{i} desc\n' + 'A' * L + '\n')
    db = Database(str(tmp_path / 's'), str(fa))
    single = list(db.yield_seqs(1, 8))
    assert [len(b) for b in single] == [1] * 7 and [b[0][0] for b in single] == [f'p{i}' for i in range(7)]
    packed = list(db.yield_seqs(500, 8))
    assert [[p for p, _ in b] for b in packed] == [['p0', 'p1', 'p2', 'p3'], ['p4', 'p5'], ['p6']]
    assert sum(len(b) for b in packed) == 7            # nothing is lost (reference bug not copied)
    capped = list(db.yield_seqs(10000, 1))
    assert max(len(b) for b in capped) == 2
    # too-short proteins never reach the path: (length - 2) * 80 < 240
    fa2 = tmp_path / 't.fasta'
    with open(fa2, 'w') as f:
        f.write('>a\nAAAA\n>b\nAAAAA\n')
    db2 = Database(str(tmp_path / 't'), str(fa2))
    assert [p for p, _ in db2.pending()] == ['b']
    db.close()
    db2.close()


def test_flat_index_bytes_follow_the_published_faiss_layout(tmp_path):
    """`.index` (src/database.py:240-243: faiss.IndexFlatL2(480) + faiss.write_index).  faiss is not installed here, so
    the file cannot be read back by it -- parity stays UNPINNED -- but the serializer is held to the layout that faiss
    1.7.4 publishes in faiss/impl/index_write.cpp, field by field, on a hand-built example:
        write_index():         fourcc "IxF2" for an IndexFlat with METRIC_L2
        write_index_header():  d (int32), ntotal (int64), two dummies 1 << 20 (int64 each), is_trained (1 byte),
                               metric_type (int32, METRIC_L2 = 1; metric_arg only follows for metric_type > 1)
        WRITEXBVECTOR(codes):  codes.size() / 4 (uint64) = ntotal * d, then the float32 values row by row."""
    import struct
    from dctdomain_amd.database import read_flat_index, write_flat_index
    vec = np.array([[0, 127, 5], [1, 2, 3]], dtype=np.float32)
    path = str(tmp_path / 't.index')
    write_flat_index(path, vec)
    expected = b''.join([
        b'IxF2',
        struct.pack('<i', 3),                      # d
        struct.pack('<q', 2),                      # ntotal
        struct.pack('<q', 1 << 20),                # dummy
        struct.pack('<q', 1 << 20),                # dummy
        b'\x01',                                   # is_trained
        struct.pack('<i', 1),                      # METRIC_L2
        struct.pack('<Q', 6),                      # codes.size() / 4
        struct.pack('<6f', 0, 127, 5, 1, 2, 3),    # codes
    ])
    assert open(path, 'rb').read() == expected
    assert len(expected) == 4 + 4 + 8 + 16 + 1 + 4 + 8 + 24
    np.testing.assert_array_equal(read_flat_index(path), vec)
    # the production shape: int8 fingerprints cast to float32, 480 wide
    fps = np.arange(3 * 480, dtype=np.int64).reshape(3, 480) % 128
    write_flat_index(path, fps.astype(np.int8).astype(np.float32))
    raw = open(path, 'rb').read()
    assert raw[:4] == b'IxF2' and struct.unpack('<i', raw[4:8])[0] == 480 and struct.unpack('<q', raw[8:16])[0] == 3
    assert len(raw) == 45 + 3 * 480 * 4
    np.testing.assert_array_equal(np.frombuffer(raw[45:], dtype=np.float32).reshape(3, 480), fps.astype(np.float32))
