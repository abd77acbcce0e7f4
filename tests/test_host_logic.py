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



def _random_encoding(rng, n_prot):
    """Records as dctfp_reccut writes them, with room to spare: {D, then per domain n_segs, (first, last) ...}."""
    from dctdomain_amd import reccut
    lens = rng.integers(22, 900, size=n_prot)
    room = reccut.reccut_room(lens)
    enc_off = np.zeros(n_prot + 1, dtype=np.int64)
    np.cumsum(room, out=enc_off[1:])
    enc = np.full(int(enc_off[-1]), -7, dtype=np.int32)
    expect = []
    for p, L in enumerate(lens.tolist()):
        n_dom = int(rng.integers(1, max(2, min(6, L // 22 + 1))))
        cuts = np.sort(rng.choice(np.arange(1, L), size=2 * n_dom - 1, replace=False)) if L > 2 * n_dom else np.arange(1, 2 * n_dom)
        bounds = [0] + cuts.tolist() + [L]
        segs = [(bounds[i], bounds[i + 1] - 1) for i in range(len(bounds) - 1)]        # 2 n_dom segments tiling the protein
        rng.shuffle(segs)
        rec, doms = [n_dom], []
        for d in range(n_dom):
            mine = segs[2 * d:2 * d + 2] if d < n_dom - 1 else segs[2 * d:]
            rec.append(len(mine))
            for a, b in mine:
                rec += [a, b]
            doms.append(','.join(f'{a + 1}-{b + 1}' for a, b in mine))
        if n_dom > 1:
            doms.append(f'1-{L}')
        assert len(rec) <= room[p]
        enc[enc_off[p]:enc_off[p] + len(rec)] = rec
        expect.append(doms)
    return lens.astype(np.int64), enc, enc_off, expect


def test_reccut_pieces_equals_format_then_parse():
    """dctfp_reccut_pieces (strings + piece table of a flush straight from the cutter's integers) against the two-step way it
    replaces: reccut_format_packed -> '1-L' appended -> PieceTable (dctfp_build_pieces)."""
    from dctdomain_amd import _lib
    import dctdomain_amd as dd
    lib = _lib.load()
    rng = np.random.default_rng(11)
    lens, enc, enc_off, expect = _random_encoding(rng, 300)
    n = len(lens)
    piece_cap, text_cap = len(enc) // 2 + n + 1, 12 * len(enc) + 32 * n + 64
    pieces = np.empty(piece_cap, dtype=_lib.PIECE_DTYPE)
    text = np.empty(text_cap, dtype=np.uint8)
    counts = np.empty(n, dtype=np.int32)
    tl, npc, nd, nu = (ctypes.c_int64() for _ in range(4))
    _lib.check(lib.dctfp_reccut_pieces(n, enc.ctypes.data, enc_off.ctypes.data, lens.ctypes.data, text.ctypes.data, text_cap,
                                       ctypes.byref(tl), counts.ctypes.data, pieces.ctypes.data, piece_cap, ctypes.byref(npc),
                                       ctypes.byref(nd), ctypes.byref(nu)), lib)
    assert nu.value == 0
    flat = text[:tl.value].tobytes().decode('ascii').split(';')
    assert flat.pop() == ''
    assert flat == [d for doms in expect for d in doms] and counts.tolist() == [len(d) for d in expect]
    ref = dd.PieceTable(lens, expect)
    assert ref.n_domains == nd.value and ref.keys == flat
    got = pieces[:npc.value]
    for f in ('row_start', 'n_rows', 'domain', 'seq'):
        np.testing.assert_array_equal(got[f], ref.pieces[f], err_msg=f)
    t = dd.PieceTable.from_pieces(lens, got, nd.value, flat, counts)
    np.testing.assert_array_equal(t.owner, ref.owner)
    np.testing.assert_array_equal(t.source, ref.source)
    np.testing.assert_array_equal(t.lengths, ref.lengths)
    # status -1, a truncated record and a segment past the end are left to the caller -- and only those
    bad = enc.copy()
    bad[enc_off[3]] = -1
    bad[enc_off[10] + 3] = lens[10]              # last residue of protein 10's first segment: outside
    bad[enc_off[20] + 1] = 10 ** 6               # n_segs beyond the room
    _lib.check(lib.dctfp_reccut_pieces(n, bad.ctypes.data, enc_off.ctypes.data, lens.ctypes.data, text.ctypes.data, text_cap,
                                       ctypes.byref(tl), counts.ctypes.data, pieces.ctypes.data, piece_cap, ctypes.byref(npc),
                                       ctypes.byref(nd), ctypes.byref(nu)), lib)
    assert nu.value == 3 and [p for p in range(n) if counts[p] == 0] == [3, 10, 20]
    assert nd.value == sum(len(d) for p, d in enumerate(expect) if p not in (3, 10, 20))
    assert lib.dctfp_reccut_pieces(n, enc.ctypes.data, enc_off.ctypes.data, lens.ctypes.data, text.ctypes.data, 10,
                                   ctypes.byref(tl), counts.ctypes.data, pieces.ctypes.data, piece_cap, ctypes.byref(npc),
                                   ctypes.byref(nd), ctypes.byref(nu)) != 0


def test_reccut_room_and_domain_encoding():
    from dctdomain_amd import _lib, reccut
    lib = _lib.load()
    ns = np.arange(0, 6000)
    np.testing.assert_array_equal(reccut.reccut_room(ns), [lib.dctfp_reccut_room(int(v)) for v in ns])
    assert reccut._encode_domains(['1-30,61-90', '31-60']).tolist() == [2, 2, 0, 29, 60, 89, 1, 30, 59]
    assert reccut._encode_domains(['1-30', 'x-3']) is None and reccut._encode_domains([]) is None
    assert reccut._encode_domains(['+1-30']) is None


def test_tensor_table_helper_equals_the_attribute_reads(monkeypatch):
    """_tensor_table.so (one C++ pass over a list of tensors) against torch's own attributes and against the Python passes it
    replaces; entries that are no tensors are marked."""
    import torch
    from dctdomain_amd import _geom
    big = torch.arange(4000 * 24, dtype=torch.float32).reshape(4000, 24)
    ts = [big[i * 7:i * 7 + (i % 5) + 1] for i in range(300)] + [big[:10, ::2], big.t()[:3], torch.zeros(5, dtype=torch.float64),
                                                                  torch.zeros((2, 3, 4), dtype=torch.float16), torch.zeros((0, 24))]
    assert _geom.helper() is not None, 'dctdomain_amd/_tensor_table.so was not built (build_ext.build_tensor_table)'
    ptrs, meta = _geom.tensor_table(ts)
    for i, t in enumerate(ts):
        d = t.dim()
        assert ptrs[i] == t.data_ptr() and meta[i, 0] == d
        assert meta[i, 1] == t.size(0) and meta[i, 3] == t.stride(0)
        if d >= 2:
            assert meta[i, 2] == t.size(1) and meta[i, 4] == t.stride(1)
        assert meta[i, 5] == _geom.code_of(t)
    assert len({int(c) for c in meta[:300, 5]}) == 1 and meta[302, 5] != meta[0, 5]
    mixed = ts[:4] + ['x', None, np.zeros(3)]
    p2, m2 = _geom.tensor_table(mixed)
    assert m2[4:, 0].tolist() == [-1, -1, -1] and p2[4:].tolist() == [0, 0, 0]
    # the Python passes fill the same table
    monkeypatch.setattr(_geom, '_helper', None)
    monkeypatch.setattr(_geom, '_helper_tried', True)
    p3, m3 = _geom.tensor_table(ts)
    np.testing.assert_array_equal(p3, ptrs)
    np.testing.assert_array_equal(m3, meta)
    p4, m4 = _geom.tensor_table(mixed)
    np.testing.assert_array_equal(m4, m2)
    np.testing.assert_array_equal(p4, p2)
