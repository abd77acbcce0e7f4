"""Similarity consumers against the reference's committed outputs:
bench/G6PD/G6PD-dctsim.txt (351 pairs from G6PD-dct.npz + G6PD.pair) and
test/test/example-search.txt (400 lines of query_db output over example-dct.npz)."""

import os
from types import SimpleNamespace

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
FIX = os.path.join(gu.GOLD, 'ref_fixtures')


def test_l1_matrix_exact():
    from dctdomain_amd.similarity import l1_matrix
    rng = np.random.default_rng(0)
    # (the kernel works on 128 x 128 tiles, 32 dwords of the fingerprints at a time: shapes around those edges, columns that
    #  end inside a thread's eight -> the narrow store path, a width past four chunks)
    for na, nb, d in ((1, 1, 480), (43, 43, 480), (130, 77, 460), (5, 200, 255), (64, 65, 7), (3, 3, 1027),
                      (128, 128, 480), (129, 257, 480), (300, 391, 128), (257, 130, 129), (1000, 517, 480)):
        a = rng.integers(0, 128, size=(na, d)).astype(np.int8)
        b = rng.integers(-128, 128, size=(nb, d)).astype(np.int8)
        exp = np.abs(a.astype(np.int64)[:, None, :] - b.astype(np.int64)[None, :, :]).sum(-1)
        got = l1_matrix(a, b).cpu().numpy()
        np.testing.assert_array_equal(got, exp)


def test_l1_matrix_16_byte_kernel_edges():
    """Rows on 16-byte boundaries take l1_matrix16_kernel (round 4): widths that end inside a 16-byte segment / inside a
    128-byte chunk (slices of wider matrices: the row stride stays a multiple of 16), row offsets, tiles past the matrix
    edges, every byte value -- against numpy and against the 4-byte kernel (experiments library, option l1_kernel = 1)."""
    import torch
    from dctdomain_amd import _lib
    from dctdomain_amd.similarity import l1_matrix
    rng = np.random.default_rng(5)
    wide_a = torch.from_numpy(rng.integers(-128, 128, size=(391, 528)).astype(np.int8)).cuda()
    wide_b = torch.from_numpy(rng.integers(-128, 128, size=(300, 528)).astype(np.int8)).cuda()
    for d in (1, 3, 15, 16, 17, 127, 128, 129, 130, 255, 470, 480, 496, 512, 528):
        for ra, rb in ((0, 0), (1, 3)):
            a, b = wide_a[ra:, :d], wide_b[rb:, :d]
            assert a.data_ptr() % 16 == 0 and a.stride(0) % 16 == 0
            exp = (a.cpu().numpy().astype(np.int64)[:, None, :] - b.cpu().numpy().astype(np.int64)[None, :, :])
            np.testing.assert_array_equal(l1_matrix(a, b).cpu().numpy(), np.abs(exp).sum(-1), err_msg=f'd={d} rows from {ra}/{rb}')
    # extremes: -128 against 127 in every byte, 480 wide = 255 * 480
    lo = torch.full((130, 480), -128, dtype=torch.int8, device='cuda')
    hi = torch.full((129, 480), 127, dtype=torch.int8, device='cuda')
    assert (l1_matrix(lo, hi) == 255 * 480).all() and (l1_matrix(hi, hi) == 0).all()
    # the two kernels on the same aligned input
    ectx = _lib.experiments_context(torch.cuda.current_device())
    a, b = wide_a[:, :480].contiguous(), wide_b[:, :480].contiguous()
    outs = []
    for which in (0, 1):
        ectx.set_option('l1_kernel', which)
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.int32, device='cuda')
        _lib.check(ectx._lib.dctfp_l1_matrix(ectx.handle, a.data_ptr(), a.shape[0], 480, b.data_ptr(), b.shape[0], 480, 480,
                                             out.data_ptr(), b.shape[0], None), ectx._lib)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    ectx.set_option('l1_kernel', 0)
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], l1_matrix(a, b).cpu().numpy())


def test_pair_sim_matches_G6PD_golden(tmp_path):
    from dctdomain_amd import dct_sim
    out = str(tmp_path / 'sim.txt')
    dct_sim.main(['--dct', os.path.join(FIX, 'G6PD-dct.npz'), '--pair', os.path.join(FIX, 'G6PD.pair'), '--output', out])
    got = open(out).read()
    exp = open(os.path.join(FIX, 'G6PD-dctsim.txt')).read()
    assert got == exp
    assert got.count('\n') == 352


def test_domain_sim_and_all_sim(tmp_path, capsys):
    from dctdomain_amd import dct_sim
    z = np.load(os.path.join(FIX, 'example-dct.npz'))
    i0, i1, i2 = z['idx'][0], z['idx'][1], z['idx'][2]
    a, b = z['dct'][i0:i1], z['dct'][i1:i2]
    maxs, s = dct_sim.domain_sim(a, b)
    ref = [1 - min(np.abs(x.astype(np.int64) - y.astype(np.int64)).sum() / 17000, 1) for x in a for y in b]
    assert maxs == max(ref) and s == ref[-1]
    assert dct_sim.prostSimilarity(a[0], a[0]) == 1
    out = str(tmp_path / 'all.txt')
    dct_sim.main(['--dct', os.path.join(FIX, 'example-dct.npz'), '--output', out])
    lines = open(out).read().strip().split('\n')
    assert len(lines) == 1 + 8 * 7 // 2
    dbout = str(tmp_path / 'db.txt')
    dct_sim.main(['--dct', os.path.join(FIX, 'example-dct.npz'), '--db', os.path.join(FIX, 'example-dct.npz'),
                  '--output', dbout, '--top', '3'])
    first = open(dbout).read().strip().split('\n')[1].split()
    assert first[0] == first[1] and first[2] == '1.0'


def test_query_search_matches_example_golden(tmp_path):
    from dctdomain_amd import query_db
    from dctdomain_amd.database import Database
    z = np.load(os.path.join(FIX, 'example-dct.npz'))
    db = Database(str(tmp_path / 'ex'), os.path.join(FIX, 'example.fasta'))
    fps = []
    for i, pid in enumerate(z['sid']):
        s, e = z['idx'][i], z['idx'][i + 1]
        doms = [str(d) for d in z['dom'][s:e]]
        fps.append(SimpleNamespace(pid=str(pid), domains=doms, quants={d: z['dct'][s + k] for k, d in enumerate(doms)}))
    db.add_fprints(fps)
    db.rename_vid()
    db.close()
    out = str(tmp_path / 'search.txt')
    query_db.main(['--query', str(tmp_path / 'ex.db'), '--db', str(tmp_path / 'ex.db'), '--out', out, '--khits', '50'])
    assert open(out).read() == open(os.path.join(FIX, 'example-search.txt')).read()


def test_row_select_matches_stable_argsort():
    import torch
    from dctdomain_amd.similarity import row_select
    rng = np.random.default_rng(4)
    for n_rows, n_cols, k, hi in ((3, 43, 43, 5000), (7, 1000, 50, 40), (1, 5, 1, 3), (5, 70000, 100, 20000), (4, 300, 300, 2)):
        d = rng.integers(0, hi, size=(n_rows, n_cols)).astype(np.int32)
        v, i = row_select(torch.from_numpy(d).cuda(), k)
        order = np.argsort(d, axis=1, kind='stable')[:, :k]
        np.testing.assert_array_equal(i, order)
        np.testing.assert_array_equal(v, np.take_along_axis(d, order, axis=1))


def test_row_select_in_registers_segments_ties_and_extremes():
    """row_select_reg_kernel (round 4: the row in the registers of one workgroup, bisection on value and column): rows of one
    and of several segments (40 960 columns each), a last segment shorter than k, rows that are one long tie, negative and
    extreme int32 values, k = 1 / 1024 (its limit) / 1025 (the radix select) -- against numpy's stable argsort and against
    the radix select on the same matrix (experiments option row_select = 1), on a strided matrix."""
    import ctypes as C
    import torch
    from dctdomain_amd import _lib
    from dctdomain_amd.similarity import row_select
    rng = np.random.default_rng(11)
    cases = [  # rows, columns, k, values
        (3, 40960, 100, lambda s: rng.integers(0, 120000, size=s)),
        (2, 40965, 100, lambda s: rng.integers(0, 3, size=s)),              # second segment of 5 entries, heavy ties
        (2, 100000, 1024, lambda s: rng.integers(-50000, 50000, size=s)),   # three segments, negative values
        (2, 100000, 1025, lambda s: rng.integers(-50000, 50000, size=s)),   # radix select
        (3, 90000, 7, lambda s: np.full(s, 17)),                              # one long tie: the first k columns
        (2, 50000, 1, lambda s: rng.integers(0, 2, size=s)),
        (2, 3000, 3000, lambda s: rng.integers(0, 10, size=s)),              # k = n_cols > 1024: radix select
        (2, 1000, 1000, lambda s: rng.integers(0, 10, size=s)),              # k = n_cols
        (2, 45000, 50, lambda s: rng.choice(np.array([-2**31, 2**31 - 1, 0, -1, 5]), size=s)),
    ]
    ectx = _lib.experiments_context(torch.cuda.current_device())
    for n_rows, n_cols, k, gen in cases:
        d = gen((n_rows, n_cols)).astype(np.int32)
        wide = torch.zeros((n_rows, n_cols + 3), dtype=torch.int32, device='cuda')
        wide[:, :n_cols] = torch.from_numpy(d).cuda()
        dist = wide[:, :n_cols]                                              # row stride > columns
        v, i = row_select(dist, k)
        order = np.argsort(d, axis=1, kind='stable')[:, :k]
        np.testing.assert_array_equal(i, order, err_msg=f'{n_rows} x {n_cols}, k = {k}')
        np.testing.assert_array_equal(v, np.take_along_axis(d, order, axis=1))
        outs = []
        for which in (0, 1):
            ectx.set_option('row_select', which)
            val = torch.empty((n_rows, k), dtype=torch.int32, device='cuda')
            idx = torch.empty((n_rows, k), dtype=torch.int32, device='cuda')
            _lib.check(ectx._lib.dctfp_row_select(ectx.handle, dist.data_ptr(), n_rows, n_cols, dist.stride(0), k, val.data_ptr(),
                                                  idx.data_ptr(), None), ectx._lib)
            torch.cuda.synchronize()
            vi = np.stack([val.cpu().numpy().astype(np.int64), idx.cpu().numpy().astype(np.int64)], axis=-1)
            outs.append(np.take_along_axis(vi, np.lexsort((vi[..., 1], vi[..., 0]), axis=1)[..., None], axis=1))
        ectx.set_option('row_select', 0)
        np.testing.assert_array_equal(outs[0], outs[1], err_msg=f'{n_rows} x {n_cols}, k = {k}: registers vs radix select')


def test_row_select_random_shapes():
    """60 random (columns, k, value range) draws around the register kernel's limits: 1 .. 3 segments, k up to 1024, value
    ranges from 2 (everything ties) to the whole int32 range -- against numpy's stable argsort."""
    import torch
    from dctdomain_amd.similarity import row_select
    rng = np.random.default_rng(23)
    for _ in range(60):
        n_cols = int(rng.choice([1, 2, 63, 64, 65, 1023, 1024, 1025, 4097, 40959, 40960, 40961, 81920, 81921, int(rng.integers(1, 120000))]))
        k = int(min(n_cols, rng.choice([1, 2, 10, 100, 128, 129, 200, 256, 257, 1000, 1024, int(rng.integers(1, 1025))])))   # (ordering networks of 128 / 256 / 1024)
        span = int(rng.choice([2, 3, 17, 1000, 122400, 2**31 - 1]))
        lo = int(rng.integers(-2**31, 2**31 - span))
        d = (rng.integers(0, span, size=(2, n_cols)) + lo).astype(np.int32)
        v, i = row_select(torch.from_numpy(d).cuda(), k)
        order = np.argsort(d, axis=1, kind='stable')[:, :k]
        np.testing.assert_array_equal(i, order, err_msg=f'{n_cols} columns, k = {k}, span {span}')
        np.testing.assert_array_equal(v, np.take_along_axis(d, order, axis=1))


def test_database_axis_tiling_and_proteins_without_fingerprints(tmp_path, monkeypatch):
    """Both consumers tile the database side as well (query_db.COL_ROWS / dct_sim.Blocks.COL_ROWS): tiny tiles must give
    the same answers as one tile; a protein with zero fingerprints in the npz (an empty stripe, data_ptr() == 0) must not
    reach the kernels."""
    from dctdomain_amd import dct_sim, query_db
    rng = np.random.default_rng(8)
    counts = [3, 0, 2, 5, 0, 0, 4, 1]                               # proteins 1, 4, 5: no fingerprint at all
    idx = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    dct = rng.integers(0, 128, size=(int(idx[-1]), 480)).astype(np.int8)
    sid = np.array([f'p{i}' for i in range(len(counts))])
    f = str(tmp_path / 'gaps-dct.npz')
    np.savez(f, sid=sid, idx=idx, dom=np.array(['1-9'] * int(idx[-1])), dct=dct)
    whole = dct_sim.Blocks(f)
    d = np.abs(dct.astype(np.int64)[:, None, :] - dct.astype(np.int64)[None, :, :]).sum(-1)
    for i in range(len(counts)):
        for j in range(len(counts)):
            blk = d[idx[i]:idx[i + 1], idx[j]:idx[j + 1]]
            if blk.size:
                assert whole.mn[i, j] == blk.min() and whole.last[i, j] == blk[-1, -1]
            else:
                assert whole.mn[i, j] == 0x7fffffff and whole.last[i, j] == 0x7fffffff
    monkeypatch.setattr(dct_sim.Blocks, 'COL_ROWS', 4)
    monkeypatch.setattr(dct_sim.Blocks, 'TILE_INTS', 12)
    tiled = dct_sim.Blocks(f)
    np.testing.assert_array_equal(tiled.mn, whole.mn)
    np.testing.assert_array_equal(tiled.last, whole.last)
    # query side: k nearest over database blocks of 7 rows == over the whole database
    q = rng.integers(0, 128, size=(9, 480)).astype(np.int8)
    db = rng.integers(0, 4, size=(40, 480)).astype(np.int8)         # few distinct values: many distance ties
    qrows = [(i, f'q{i % 3}', '1-9') for i in range(len(q))]
    drows = [(i, f'd{i}', '1-9') for i in range(len(db))]
    one = list(query_db.search(qrows, q, drows, db, 12))
    monkeypatch.setattr(query_db, 'COL_ROWS', 7)
    monkeypatch.setattr(query_db, 'TILE_INTS', 21)
    assert list(query_db.search(qrows, q, drows, db, 12)) == one
    dist = np.abs(q.astype(np.int64)[:, None, :] - db.astype(np.int64)[None, :, :]).sum(-1)
    top = round(1 - dist[[0, 3, 6]].min() / 17000, 4)                # protein q0 = query rows 0, 3, 6
    assert [ln for ln in one if ln.startswith('Query: q0')][0].endswith(f'Similarity: {top}')
