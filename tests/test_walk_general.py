"""GPU parity of the GENERAL walk kernel (`walk_gen_kernel`, round 4): every shape `walk_ab_kernel` does not take -- kept
sizes other than 3 x 65..80 (PROST's [5, 44] / [3, 85], anything n = 2..8, m <= 128), widths below 512 or not a multiple of 4
(an L x L contact map as a layer), float64 rows -- in one launch, Y' never in HBM.

* the reference's golden cases `qdim_*` and `contact_layer_*` (and every other golden the kernel is eligible for) with the
  walk path forced on a one-protein call, `last_path == 2` asserted;
* batches large enough for the default dispatch (>= 256 jobs), `last_path == 2` asserted, bit-exact against the oracle:
  PROST's shapes, n = 2 .. 8, unaligned widths (one channel per lane), float64 rows (through `walk_ab_kernel<double>` at the
  reference's [3, 80], through the general kernel otherwise), multi-domain lists (parts + whole protein, discontinuous).
"""

import numpy as np
import pytest

import golden_util as gu
from oracle import dct_oracle as orc
from recipes import make_input

pytestmark = pytest.mark.gpu

ALL_OK = gu.cases(expect='ok')


@pytest.fixture(scope='module')
def dd():
    import torch
    assert torch.cuda.is_available()
    import dctdomain_amd
    return dctdomain_amd


def _lds_fits(n, m, D, esz):
    """The host's rule (dctfp.hip, gen_slot_bytes): one slot of Y'[n][CH] + partial [S][n][cp] within 150 KB, <= 16 waves."""
    for vec in ((16 // esz), 1):
        if vec > 1 and D % vec:
            continue
        waves = -(-D // (64 * vec))
        cp = 16 * -(-m // 16)
        if waves <= 16 and (n * waves * 64 * vec + (0 if cp <= 64 * vec else waves * n * cp)) * 8 + 64 <= 150 * 1024:
            return True
        if vec == 1:
            return False
    return False


def _general_last_group(case, layers):
    """Is the LAST layer group of the case one the general kernel takes (what `last_path` reports)?"""
    x, n, m = layers[-1], case['qdim'][-2], case['qdim'][-1]
    tuned = n == 3 and 64 < m <= (96 if x.dtype == np.float32 else 80) and 512 <= x.shape[1] <= 2560 and x.shape[1] % 4 == 0
    return (not tuned) and n >= 2 and m >= 2 and x.dtype in (np.float32, np.float64) and x.shape[0] <= 8192 \
        and len(case['keys']) > 0 and _lds_fits(n, m, x.shape[1], x.dtype.itemsize)


def test_goldens_through_the_general_walk_kernel(dd):
    import torch
    ctx = dd.get_context(torch.cuda.current_device())
    ctx.set_option('path', 2)
    seen = set()
    six_groups = {'qdim_3x85_L64_D640', 'qdim_3x85_L300_D1280'}
    try:
        for case in ALL_OK:
            layers = gu.build_layers(case)
            if len({x.shape[0] for x in layers}) != 1:
                continue
            qd = case['qdim']
            table = dd.PieceTable([layers[0].shape[0]], [case['domains']])
            if table.n_domains == 0:
                continue
            lbs = [dd.LayerBatch([torch.from_numpy(x).cuda()], qd[2 * i], qd[2 * i + 1]) for i, x in enumerate(layers)]
            out = dd.quantize_batch(lbs, table, ctx=ctx).cpu().numpy()
            if _general_last_group(case, layers):
                assert ctx.get_option('last_path') == 2, f"{case['id']} did not run a walk kernel"
                assert ctx.get_option('last_walk_groups') == 0, case['id']
                seen.add(case['id'])
            elif case['id'] in six_groups:      # [3, 85] at D = 640 / 1280: walk_ab_kernel's builds with six column groups (round 5)
                assert ctx.get_option('last_path') == 2 and ctx.get_option('last_walk_groups') == 6, case['id']
                seen.add(case['id'])
            got, off = {}, 0
            for i in range(len(layers)):
                nm = qd[2 * i] * qd[2 * i + 1]
                for row, key in enumerate(table.keys):
                    got.setdefault(key, []).append(out[row, off:off + nm])
                off += nm
            exp = gu.expected(case)
            assert list(got) == case['keys']
            for key in exp:
                np.testing.assert_array_equal(np.concatenate(got[key]).astype(np.int64), exp[key].astype(np.int64),
                                              err_msg=f"{case['id']} {key}")
    finally:
        ctx.set_option('path', 0)
    # (qdim_n1: n = 1 -> the zero-fill kernel, nothing to contract; qdim_mixed: its last layer group is [3, 80] at D = 640,
    #  walk_ab_kernel's own shape -- its first group, [5, 44], went through the general kernel all the same)
    want = {c['id'] for c in ALL_OK if c['id'].startswith(('qdim_', 'contact_layer_')) and c['id'] not in ('qdim_n1', 'qdim_mixed')}
    assert want <= seen, sorted(want - seen)


def _batch(n_seq, D, seed, lo=12, hi=160, multi=True, recipe='esm', dtype=np.float32):
    rng = np.random.default_rng(seed)
    lens, doms = [], []
    for s in range(n_seq):
        L = int(rng.integers(lo, hi))
        lens.append(L)
        if multi and s % 3 == 0 and L >= 3 * lo:
            k = int(rng.integers(2, 4))
            cuts = np.sort(rng.choice(np.arange(lo, L - lo + 1), size=k - 1, replace=False)) if L - 2 * lo + 1 >= k - 1 else []
            edges = [0] + [int(c) for c in cuts] + [L]
            dl = [f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:]) if b - a >= lo]
            if len(dl) >= 3 and s % 2 == 0:
                dl = [dl[0] + ',' + dl[-1]] + dl[1:-1]          # discontinuous first + last part
            doms.append(dl + [f'1-{L}'])
        else:
            doms.append([f'1-{L}'])
    layers = [[make_input(recipe if k == 0 else 'gauss', L, D, seed * 7919 + 2 * s + k).astype(dtype) for s, L in enumerate(lens)]
              for k in range(2)]
    return lens, doms, layers


SHAPES = [  # (id, qdim of the two layers, D, storage, min rows)
    ('prost_5x44_D1280', [5, 44, 5, 44], 1280, np.float32, 12),
    ('prost_3x85_D1280', [3, 85, 3, 85], 1280, np.float32, 12),
    ('4x80_D640', [4, 80, 4, 80], 640, np.float32, 12),
    ('8x128_D640', [8, 128, 8, 128], 640, np.float32, 12),
    ('2x16_and_7x33_D640', [2, 16, 7, 33], 640, np.float32, 12),
    ('6x100_D1280', [6, 100, 6, 100], 1280, np.float32, 12),               # one slot of LDS only
    ('prost_5x44_D2560', [5, 44, 5, 44], 2560, np.float32, 12),            # ten waves, one slot
    ('contact_like_5x44_D257', [5, 44, 5, 44], 257, np.float32, 12),       # odd width: one channel per lane
    ('3x80_D200', [3, 80, 3, 80], 200, np.float32, 12),                    # the reference's qdim below the tuned kernel's widths
    ('f64_rows_3x80_D1280', [3, 80, 3, 80], 1280, np.float64, 12),         # walk_ab_kernel<double>
    ('f64_rows_5x44_D640', [5, 44, 5, 44], 640, np.float64, 12),
    ('mixed_5x44_then_3x80_D640', [5, 44, 3, 80], 640, np.float32, 12),    # two layer groups: general, then tuned
    # round 5: 80 < m <= 96 through walk_ab_kernel's builds with six column groups ([E | O] halves of 48 slots)
    ('prost_3x85_D640', [3, 85, 3, 85], 640, np.float32, 12),
    ('prost_3x85_D2560', [3, 85, 3, 85], 2560, np.float32, 12),
    ('3x96_D1280', [3, 96, 3, 96], 1280, np.float32, 12),
    ('3x81_then_3x80_D1284', [3, 81, 3, 80], 1284, np.float32, 12),        # six groups, then five; a width that ends inside a pair group
    ('3x85_f64_rows_D640', [3, 85, 3, 85], 640, np.float64, 12),           # float64 rows stay with the general kernel
]


@pytest.mark.parametrize('name,qdim,D,dtype,lo', SHAPES, ids=[s[0] for s in SHAPES])
def test_general_shapes_in_batches_default_dispatch(dd, name, qdim, D, dtype, lo):
    import torch
    one_group = qdim[:2] == qdim[2:]            # (the dispatch judges every group of equal layers on its own job count)
    lens, doms, layers = _batch(150 if one_group else 280, D, 101 + D + qdim[0], lo=lo, hi=90 if D >= 1280 else 140, dtype=dtype)
    table = dd.PieceTable(lens, doms)
    assert (2 if one_group else 1) * table.n_domains >= 256
    dev = [[torch.from_numpy(x).cuda() for x in layers[k]] for k in range(2)]
    lbs = [dd.LayerBatch(dev[k], qdim[2 * k], qdim[2 * k + 1]) for k in range(2)]
    ctx = dd.get_context(torch.cuda.current_device())
    # (a) whole proteins only: the default dispatch sends the batch to the walk kernels
    whole = dd.PieceTable(lens, [[f'1-{L}'] for L in lens])
    assert (2 if one_group else 1) * whole.n_domains >= 256
    launches = ctx.get_option('walk_launches')
    out_whole = dd.quantize_batch(lbs, whole, ctx=ctx).cpu().numpy()
    assert ctx.get_option('last_path') == 2, name
    assert ctx.get_option('walk_launches') - launches == (1 if one_group else 2), name
    # (b) parts + whole protein, discontinuous parts: the general kernel when asked for (by default the two kernels take such a
    # batch: their stage A reads the rows of a protein once for all of its domains)
    # Round 5: for n <= 5 it streams such proteins as FUSED walks (rows read once, the whole protein collected beside the parts).
    n_last, m_last = qdim[2], qdim[3]
    tuned_last = n_last == 3 and 64 < m_last <= 96 and 512 <= D <= 2560 and dtype == np.float32     # walk_ab_kernel's own shapes (m > 80: six column groups)
    expect_fused = 0 if tuned_last else int(n_last <= 5)
    ctx.set_option('path', 2)
    try:
        out = dd.quantize_batch(lbs, table, ctx=ctx).cpu().numpy()
        assert ctx.get_option('last_path') == 2, name
        assert ctx.get_option('last_gen_fused') == expect_fused, name
        assert ctx.get_option('last_walk_groups') == ((6 if m_last > 80 else 5) if tuned_last else 0), name
        if expect_fused:                      # ... and every job on its own when asked to: the same bytes
            ctx.set_option('gen_fuse', 0)
            try:
                unfused = dd.quantize_batch(lbs, table, ctx=ctx).cpu().numpy()
                assert ctx.get_option('last_path') == 2 and ctx.get_option('last_gen_fused') == 0
            finally:
                ctx.set_option('gen_fuse', 1)
            np.testing.assert_array_equal(out, unfused, err_msg=f'{name}: fused vs unfused general kernel')
    finally:
        ctx.set_option('path', 0)
    for s in range(0, len(lens), 7):
        q = orc.quantize([layers[0][s], layers[1][s]], [f'1-{lens[s]}'], qdim)
        np.testing.assert_array_equal(out_whole[s].astype(np.int64), q[f'1-{lens[s]}'], err_msg=f'{name}: whole protein {s}')
    ctx.set_option('path', 1)
    try:
        two = dd.quantize_batch(lbs, table, ctx=ctx).cpu().numpy()
        assert ctx.get_option('last_path') == 1
    finally:
        ctx.set_option('path', 0)
    np.testing.assert_array_equal(out, two, err_msg=f'{name}: walk kernel vs two-kernel path')
    row = 0
    for s, L in enumerate(lens):
        q = orc.quantize([layers[0][s], layers[1][s]], doms[s], qdim)
        for key, exp in q.items():
            assert table.keys[row] == key
            np.testing.assert_array_equal(out[row].astype(np.int64), exp, err_msg=f'{name}: seq {s} (L={L}) domain {key}')
            row += 1
    assert row == table.n_domains


def test_general_kernel_nan_inf_and_constant_channels(dd):
    """Degenerate inputs through the general kernel: a NaN, an inf, an exactly constant channel -> the (layer, domain) block
    is all 0, as everywhere (src/fingerprint.py:110-123, :194-195), and the constant channel is counted."""
    import torch
    lens, doms, layers = _batch(140, 640, 77, multi=False)
    layers[0][3][5, 17] = np.nan
    layers[0][4][2, 600] = np.inf
    layers[1][5][:, 33] = 1.25
    qdim = [5, 44, 5, 44]
    table = dd.PieceTable(lens, doms)
    dev = [[torch.from_numpy(x).cuda() for x in layers[k]] for k in range(2)]
    lbs = [dd.LayerBatch(dev[k], 5, 44) for k in range(2)]
    ctx = dd.get_context(torch.cuda.current_device())
    ctx.set_option('degenerate_channels', 0)
    out = dd.quantize_batch(lbs, table, ctx=ctx).cpu().numpy()
    assert ctx.get_option('last_path') == 2
    assert (out[3, :220] == 0).all() and (out[4, :220] == 0).all() and (out[5, 220:] == 0).all()
    assert ctx.get_option('degenerate_channels') == 1
    assert ctx.get_option('degenerate_seen') == 1
    for s in (0, 1, 2, 6, 139):
        q = orc.quantize([layers[0][s], layers[1][s]], doms[s], qdim)
        np.testing.assert_array_equal(out[s].astype(np.int64), q[f'1-{lens[s]}'])


@pytest.mark.parametrize('qdim,D', [([5, 44, 5, 44], 2560), ([3, 100, 3, 100], 1280), ([4, 80, 4, 80], 640), ([2, 30, 5, 44], 1280)],
                         ids=['5x44_D2560', '3x100_D1280', '4x80_D640', '2x30_5x44_D1280'])
def test_fused_general_kernel_on_reccut_shaped_lists(dd, qdim, D):
    """PROST-shaped kept sizes on RecCut-shaped domain lists (1-6 parts tiling the protein, some discontinuous, + the whole
    protein; single-domain proteins between them) in one call: the general kernel's fused walks ("path" = 2) against the
    two-kernel path the default dispatch still picks for such lists (it is the faster one there: profiles/r05/gen_probe_fused.txt)
    byte for byte, and against the oracle on a sample."""
    import torch
    rng = np.random.default_rng(77 + D + qdim[0])
    lens, doms = [], []
    for s in range(420):
        L = int(rng.integers(40, 420))
        lens.append(L)
        k = int(rng.integers(1, 7)) if L >= 150 else 1
        if k == 1:
            doms.append([f'1-{L}'])
            continue
        extra = rng.multinomial(L - 22 * k, np.ones(k) / k)          # parts of at least 22 rows (RecCut's Min_Size)
        edges = [0] + np.cumsum(22 + extra).tolist()
        assert edges[-1] == L
        parts = [f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:])]
        if k >= 3 and s % 2 == 0:
            parts = [parts[0] + ',' + parts[-1]] + parts[1:-1]       # a discontinuous domain: head + tail
        if s % 5 == 0:
            parts = parts[::-1]                                       # parts in another order than along the chain
        doms.append(parts + [f'1-{L}'])
    layers = [[make_input('esm' if k == 0 else 'gauss', L, D, 31 * D + 2 * s + k).astype(np.float32) for s, L in enumerate(lens)]
              for k in range(2)]
    table = dd.PieceTable(lens, doms)
    dev = [[torch.from_numpy(x).cuda() for x in layers[k]] for k in range(2)]
    lbs = [dd.LayerBatch(dev[k], qdim[2 * k], qdim[2 * k + 1]) for k in range(2)]
    ctx = dd.get_context(torch.cuda.current_device())
    ctx.set_option('path', 2)
    try:
        out = dd.quantize_batch(lbs, table, ctx=ctx).cpu().numpy()
        assert ctx.get_option('last_path') == 2 and ctx.get_option('last_gen_fused') == 1
    finally:
        ctx.set_option('path', 0)
    two = dd.quantize_batch(lbs, table, ctx=ctx).cpu().numpy()
    assert ctx.get_option('last_path') == 1 and ctx.get_option('last_gen_fused') == 0
    np.testing.assert_array_equal(out, two)
    row_of = np.concatenate([[0], np.cumsum([len(d) for d in doms])])
    for s in range(0, len(lens), 9):
        q = orc.quantize([layers[0][s], layers[1][s]], doms[s], qdim)
        for i, (key, exp) in enumerate(q.items()):
            assert table.keys[row_of[s] + i] == key
            np.testing.assert_array_equal(out[row_of[s] + i].astype(np.int64), exp, err_msg=f'seq {s} (L={lens[s]}) domain {key}')
