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
    assert t.owner == [0, 0, 1]
    assert t.lengths == [70, 100, 50]
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
