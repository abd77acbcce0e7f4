"""CPU-side checks: domain-string rules, piece tables, and that the C-ABI library loads
and exports every symbol include/dctfp.h declares (no GPU compute here)."""

import ctypes
import json
import os
import re

import numpy as np
import pytest

import golden_util as gu
from oracle import dct_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from dctdomain_amd import _lib
    with open(os.path.join(ROOT, 'include', 'dctfp.h')) as fh:
        header = fh.read()
    declared = sorted(set(re.findall(r'\b(dctfp_[a-z0-9_]+)\s*\(', header)))
    assert len(declared) >= 11
    for path in (_lib.LIB_PATH, _lib.EXPERIMENTS_LIB_PATH):      # the product and its twin with the engineering knobs
        lib = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(lib, name), f'{name} declared in include/dctfp.h but not exported by {path}'
        assert lib.dctfp_version() == int(re.search(r'#define DCTFP_VERSION (\d+)', header).group(1))
    assert sorted(_lib.EXPORTS) == declared


def test_struct_layouts_match_header():
    from dctdomain_amd import _lib
    assert ctypes.sizeof(_lib.Layer) == 40
    assert _lib.PIECE_DTYPE.itemsize == 24
    assert _lib.PIECE_DTYPE.fields['row_start'][1] == 0
    assert _lib.PIECE_DTYPE.fields['n_rows'][1] == 8
    assert _lib.PIECE_DTYPE.fields['domain'][1] == 12
    assert _lib.PIECE_DTYPE.fields['seq'][1] == 16


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import dctdomain_amd as dd
    with pytest.raises(dd.DctfpError):
        dd.Context(0)
    fp = dd.Fingerprint(pid='a', seq='AAAA', embed={0: np.zeros((4, 96), np.float32)}, domains=['1-4'])
    with pytest.raises(RuntimeError):
        fp.quantize([3, 80])


def test_product_does_not_import_the_oracle():
    """The product path must never route through the CPU checker (or scipy)."""
    pkg = os.path.join(ROOT, 'dctdomain_amd')
    pat = re.compile(r'^\s*(from|import)\s+(oracle|scipy)\b', re.M)
    n = 0
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                with open(os.path.join(dirpath, f)) as fh:
                    src = fh.read()
                assert not pat.search(src), f'{f} imports the oracle or scipy'
                assert 'liboracle' not in src and 'oracle/_ref' not in src, f
                n += 1
    assert n >= 5


def test_split_domain_matches_reference_table():
    from dctdomain_amd.domains import split_domain
    with open(os.path.join(gu.GOLD, 'getdoms_golden.json')) as fh:
        table = json.load(fh)
    for row in table:
        pieces, key = split_domain(row['dom'], row['L'])
        rows = [r for (s, n) in pieces for r in range(s, s + n)]
        assert key == row['key'], row
        assert rows == row['rows'], row
        # and the oracle's own restatement agrees
        op, ok = orc.split_domain(row['dom'], row['L'])
        assert ok == key


def test_piece_table_layout():
    from dctdomain_amd import PieceTable
    t = PieceTable([100, 50], [['1-30,61-100', '101-120', '1-100'], ['1-50', '0-10']])
    assert t.n_domains == 3
    assert t.keys == ['1-30,61-100', '1-100', '1-50']
    assert list(t.owner) == [0, 0, 1]
    assert list(t.lengths) == [70, 100, 50]
    p = t.pieces
    assert p['row_start'].tolist() == [0, 60, 0, 0]
    assert p['n_rows'].tolist() == [30, 40, 100, 50]
    assert p['domain'].tolist() == [0, 0, 1, 2]
    assert p['seq'].tolist() == [0, 0, 0, 1]
    w = PieceTable.whole_sequences([5, 7])
    assert w.pieces['n_rows'].tolist() == [5, 7] and w.keys == ['1-5', '1-7']


def test_fingerprint_is_picklable_and_keeps_the_reference_fields():
    """make_db passes Fingerprint objects through multiprocessing.Pool.starmap (src/make_db.py:48-49)."""
    import pickle
    import dataclasses
    import dctdomain_amd as dd
    x = np.arange(12, dtype=np.float32).reshape(4, 3)
    fp = dd.Fingerprint(pid='P1', seq='ACDE', embed={15: x, 21: x + 1}, contacts=np.eye(4, dtype=np.float32), domains=['1-4'])
    assert [f.name for f in dataclasses.fields(fp)] == ['pid', 'seq', 'embed', 'contacts', 'domains', 'quants']
    fp.quants['1-4'] = np.arange(480)
    back = pickle.loads(pickle.dumps(fp))
    assert back.pid == 'P1' and back.domains == ['1-4'] and list(back.embed) == [15, 21]
    np.testing.assert_array_equal(back.embed[21], x + 1)
    np.testing.assert_array_equal(back.quants['1-4'], np.arange(480))
    empty = dd.Fingerprint()
    assert empty.pid == '' and empty.embed == {} and empty.domains == [] and empty.quants == {}
    assert isinstance(empty.contacts, np.ndarray)
    for name in ('writece', 'reccut', 'scale', 'idct_quant', 'get_doms', 'quantize'):
        assert callable(getattr(fp, name))


def _table_python(seq_rows, domains):
    """PieceTable through the string-by-string Python path (domains.split_domain)."""
    from dctdomain_amd import PieceTable
    t = PieceTable.__new__(PieceTable)
    t.seq_rows = np.ascontiguousarray(np.asarray(seq_rows, dtype=np.int64))
    t._init_python([d for doms in domains for d in doms], np.array([len(d) for d in domains], dtype=np.int32))
    return t


def _same_table(a, b):
    assert a.n_domains == b.n_domains
    assert a.keys == b.keys
    for name in ('owner', 'source', 'lengths'):
        assert list(getattr(a, name)) == list(getattr(b, name)), name
    assert a.pieces.tolist() == b.pieces.tolist()


def test_piece_table_c_builder_matches_the_reference_table_and_the_python_rules():
    """dctfp_build_pieces (the C builder behind PieceTable) against the get_doms table produced by the imported reference
    (tests/golden/getdoms_golden.json) and against domains.split_domain on seeded random strings that hit every quirk:
    pieces beginning beyond the sequence (dropped, the next one skipped, the first equal string removed), beg == 0,
    clipped ends, empty domains, repeated pieces."""
    from dctdomain_amd import PieceTable
    with open(os.path.join(gu.GOLD, 'getdoms_golden.json')) as fh:
        gold = json.load(fh)
    for row in gold:
        t = PieceTable([row['L']], [[row['dom']]])
        rows = [r for p in t.pieces for r in range(int(p['row_start']), int(p['row_start']) + int(p['n_rows']))]
        assert rows == row['rows'], row
        if rows:
            assert t.keys == [row['key']], row
        else:
            assert t.n_domains == 0
    # all golden strings as ONE batch
    _same_table(PieceTable([r['L'] for r in gold], [[r['dom']] for r in gold]),
                _table_python([r['L'] for r in gold], [[r['dom']] for r in gold]))
    rng = np.random.default_rng(5)
    for _ in range(300):
        n_seq = int(rng.integers(1, 6))
        lens = [int(rng.integers(1, 120)) for _ in range(n_seq)]
        doms = []
        for L in lens:
            dl = []
            for _ in range(int(rng.integers(0, 5))):
                parts = []
                for _ in range(int(rng.integers(1, 5))):
                    kind = rng.integers(0, 10)
                    if kind == 0:
                        b, e = 0, int(rng.integers(0, L + 20))
                    elif kind <= 2:
                        b = int(rng.integers(L, L + 30))
                        e = b + int(rng.integers(0, 20))
                    else:
                        b = int(rng.integers(1, L + 1))
                        e = int(rng.integers(b - 3 if b > 3 else b, L + 15))
                    parts.append(f'{b}-{e}')
                    if rng.random() < 0.25:
                        parts.append(parts[int(rng.integers(0, len(parts)))])      # a repeated piece
                dl.append(','.join(parts))
            doms.append(dl)
        _same_table(PieceTable(lens, doms), _table_python(lens, doms))


def test_piece_table_leaves_odd_strings_to_python():
    """Whatever is not digits-digits[,...] keeps Python's own semantics: int() accepts blanks, '+', '_'; a malformed piece
    raises ValueError like the reference's `beg, end = reg.split('-')`."""
    from dctdomain_amd import PieceTable
    t = PieceTable([100, 100], [[' 1 - 30', '+5-10', '1_0-2_0'], ['1-100']])
    assert t.keys == [' 1 - 30', '+5-10', '1_0-2_0', '1-100'] and list(t.lengths) == [30, 6, 11, 100]
    for bad in ('', '5', '1-2-3', '1-5,', 'a-b'):
        with pytest.raises(ValueError):
            PieceTable([100], [[bad]])
    with pytest.raises(ValueError):
        PieceTable([100, 50], [['1-5']])           # one domain list per sequence
    e = PieceTable([10, 10], [[], []])
    assert e.n_domains == 0 and len(e.pieces) == 0 and e.keys == []


def test_stitch_sizes_in_c_match_the_reference_rules():
    """dctfp_stitch_sizes (host only): rows of the stitched embedding = first window + sum(window - overlap)
    (src/embedding.py:185-187), side of the combined contact map = inc * i + rows_i of the last window (:123-150); the
    reference's torch expressions fail to broadcast where this returns DCTFP_ERR_SHAPE."""
    import ctypes as C
    from dctdomain_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)

    def sizes(rows_per_seq, step, square):
        rows = np.array([r for seq in rows_per_seq for r in seq], dtype=np.int32)
        seq_win = np.zeros(len(rows_per_seq) + 1, dtype=np.int64)
        np.cumsum([len(s) for s in rows_per_seq], out=seq_win[1:])
        out = np.zeros(len(rows_per_seq), dtype=np.int64)
        rc = lib.dctfp_stitch_sizes(rows.ctypes.data, seq_win.ctypes.data, len(rows_per_seq), step, square, out.ctypes.data)
        return rc, out

    for _ in range(50):
        seqs = []
        for _s in range(int(rng.integers(1, 9))):
            n_win = int(rng.integers(1, 6))
            seqs.append([500] * (n_win - 1) + [int(rng.integers(201, 501))])
        rc, out = sizes(seqs, 200, 0)
        assert rc == 0 and out.tolist() == [s[0] + sum(r - 200 for r in s[1:]) for s in seqs]
        rc, out = sizes(seqs, 300, 1)
        assert rc == 0 and out.tolist() == [300 * (len(s) - 1) + s[-1] for s in seqs]
    assert sizes([[500, 150]], 200, 0)[0] == _lib.DCTFP_ERR_SHAPE          # a window not longer than the overlap
    assert b'not longer than the overlap' in lib.dctfp_last_error()
    assert sizes([[100, 500]], 200, 0)[0] == _lib.DCTFP_ERR_SHAPE          # the running embedding shorter than the overlap
    assert sizes([[200, 300]], 300, 1)[0] == _lib.DCTFP_ERR_SHAPE          # window offset beyond the running contact map
    assert sizes([[300, 300]], 300, 1) [0] == 0


def test_runtime_report_and_crash_handler_exports():
    """dctfp_runtime_info works without a GPU (the runtime answers 'no device', the maps are still read);
    dctfp_crash_handler installs and removes itself."""
    import ctypes as C
    from dctdomain_amd import _lib
    lib = _lib.load()
    buf = C.create_string_buffer(4096)
    n = lib.dctfp_runtime_info(buf, len(buf))
    text = buf.value.decode()
    assert n >= 1 and 'compiled against HIP' in text and 'libamdhip64' in text
    assert lib.dctfp_runtime_info(buf, 8) == _lib.DCTFP_ERR_INVALID
    assert len(_lib.mapped_runtimes()['libamdhip64']) == 1
    assert lib.dctfp_crash_handler(1) == 0 and lib.dctfp_crash_handler(0) == 0
    if os.environ.get('DCTFP_CRASH_BACKTRACE') == '1':
        assert lib.dctfp_crash_handler(1) == 0      # (leave it as conftest asked for it)


def test_order_pairs_is_lexsort_by_value_then_index():
    """similarity.order_pairs (one sort of packed 64-bit keys; what orders more than 1024 survivors per row and the candidates of
    several database blocks) against np.lexsort, the whole int32 range of values, ties included."""
    from dctdomain_amd.similarity import order_pairs
    rng = np.random.default_rng(12)
    for span in (3, 1000, 2**31 - 1):
        v = rng.integers(-span, span, size=(50, 300)).astype(np.int64)
        v[:, :4] = np.array([-2**31, 2**31 - 1, -2**31, 2**31 - 1])
        i = np.stack([rng.permutation(2**20)[:300] for _ in range(50)]).astype(np.int64)
        order = np.lexsort((i, v), axis=1)
        a, b = order_pairs(v, i)
        np.testing.assert_array_equal(a, np.take_along_axis(v, order, axis=1))
        np.testing.assert_array_equal(b, np.take_along_axis(i, order, axis=1))


def test_ranked_hits_is_the_reference_list_sort():
    """query_db.ranked_hits (one stable argsort, scores rounded at once) against the reference's own bookkeeping: a dict of
    (fingerprint, hit) -> distance filled in that order, sorted by distance (Python's stable sort), the first khits, each score
    ``round(1 - (d / 17000), 4)`` of a numpy scalar -- ties in every position, distances beyond 17 000 (negative scores)."""
    from dctdomain_amd.query_db import ranked_hits
    rng = np.random.default_rng(31)
    for n_fp, k, khits, span in ((1, 1, 1, 5), (3, 7, 10, 4), (5, 100, 100, 30000), (4, 50, 500, 3), (2, 100, 50, 122400)):
        d = np.sort(rng.integers(0, span, size=(n_fp, k)), axis=1).astype(np.int64)      # rows ascending, as a flat index returns them
        top_hits = {}
        for i, row in enumerate(d):
            for j, dist in enumerate(row):
                top_hits[i, j] = dist
        ref = list(dict(sorted(top_hits.items(), key=lambda x: x[1])).keys())[:khits]
        ii, jj, scores = ranked_hits(d, khits)
        assert list(zip(ii, jj)) == ref
        assert [str(s) for s in scores] == [str(round(1 - (top_hits[key] / 17000), 4)) for key in ref]

