"""GPU parity of the walk kernel (K3, `walk_ab_kernel`) -- the kernel every bench number comes from -- under the
dispatch a user gets: calls of >= 256 jobs.  Three things the one-protein golden tests cannot show:

* short jobs (3 .. 7 rows: the shortest domain the reference accepts is '1-3', src/fingerprint.py:174-201) through the
  first-row load, the tail groups and a partially filled flush, plain and fused (parts of 3 .. 7 rows + whole protein);
* the BASELINE config 2 shape (L = 500, D = 1280, 2 layers) and the config 3 shape (ragged L in [50, 2000]) as batches
  large enough for the default dispatch, with the reference's golden cases placed inside the batch;
* that the kernel that ran IS the walk kernel (`last_path == 2`), so a dispatch regression cannot silently move these
  tests to the two-kernel path.
"""

import numpy as np
import pytest

import golden_util as gu
from oracle import dct_oracle as orc
from recipes import make_input

pytestmark = pytest.mark.gpu

CASES = {c['id']: c for c in gu.cases(expect='ok')}


@pytest.fixture(scope='module')
def dd():
    import torch
    assert torch.cuda.is_available()
    import dctdomain_amd
    return dctdomain_amd


class _Options:
    """dctfp_set_option for the duration of a block."""

    def __init__(self, ctx, **opts):
        self.ctx, self.opts = ctx, opts

    def __enter__(self):
        self.saved = {k: self.ctx.get_option(k) for k in self.opts}
        for k, v in self.opts.items():
            self.ctx.set_option(k, v)
        return self.ctx

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            self.ctx.set_option(k, v)


def _short_job_batch(D, seed, recipe):
    """Sequences whose every domain has 3 .. 7 rows (+ the whole protein of the fused ones):
    lens, domain strings, float32 layers[k][s]; goldens = {sequence index: golden case id} (one-layer cases: layer 0)."""
    rng = np.random.default_rng(seed)
    lens, doms, golden_at = [], [], {}
    # (a) the reference's own shortest cases at this width
    for L in (3, 4, 5):
        for fam in ('gauss', 'esm'):
            golden_at[len(lens)] = f'grid_{fam}_L{L}_D{D}'
            lens.append(L)
            doms.append([f'1-{L}'])
    # (b) proteins tiled by parts of 3 .. 7 rows + the whole protein (fused walks), some parts discontinuous
    for _ in range(48):
        parts = [int(v) for v in rng.integers(3, 8, size=int(rng.integers(2, 7)))]
        edges = np.concatenate([[0], np.cumsum(parts)])
        L = int(edges[-1])
        dl = [f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:])]
        if len(dl) >= 3 and rng.random() < 0.3:                 # first + last part as one discontinuous domain
            dl = [dl[0] + ',' + dl[-1]] + dl[1:-1]
        lens.append(L)
        doms.append(dl + [f'1-{L}'])
    # (c) single short domains: whole sequences of 3 .. 7 rows, inner windows, two-piece domains (plain walks)
    for i in range(150):
        L = int(rng.integers(3, 8))
        kind = i % 3
        if kind == 0:
            lens.append(L)
            doms.append([f'1-{L}'])
        elif kind == 1:
            lens.append(L + 4)
            doms.append([f'3-{L + 2}'])
        else:
            lens.append(12)
            doms.append([f'1-3,{16 - L}-12' if L > 3 else '2-4'])      # two pieces: 3 + (L - 3) rows
    layers = []
    for k in range(2):
        per_seq = []
        for s, L in enumerate(lens):
            if k == 0 and s in golden_at and recipe == 'golden':
                per_seq.append(gu.build_layers(CASES[golden_at[s]])[0])
            else:
                per_seq.append(make_input('esm' if recipe == 'golden' else recipe, L, D, seed * 1000 + 2 * s + k))
        layers.append(per_seq)
    return lens, doms, layers, golden_at


@pytest.mark.parametrize('D', [640, 1280, 2560])
@pytest.mark.parametrize('storage', ['float32', 'float16'])
def test_short_jobs_through_the_walk_kernel(dd, D, storage):
    """Jobs of 3, 4, 5, 6, 7 rows, plain and fused, under the default dispatch (>= 256 jobs) and with path = 2:
    bit-exact against the oracle; the reference goldens grid_*_L3/L4/L5 inside the float32 batch."""
    import torch
    # float16 storage: gaussian values (an ESM-like channel offset of 200 makes every short float16 channel constant:
    # the fenced 0/0 class, tests/test_fences.py)
    lens, doms, layers, golden_at = _short_job_batch(D, 31 + D, 'golden' if storage == 'float32' else 'gauss')
    dt = getattr(torch, storage)
    dev = [[torch.from_numpy(x).to(dt).cuda() for x in layers[k]] for k in range(2)]
    # what the kernels read, as float32 (exact) for the oracle
    seen = [[t.float().cpu().numpy() for t in dev[k]] for k in range(2)]
    for k in range(2):      # no exactly constant channel in any job (that class is fenced, not matched)
        for s, x in enumerate(seen[k]):
            for dom in doms[s]:
                rows, _ = orc.get_doms(x, dom)
                assert (rows.max(axis=0) > rows.min(axis=0)).all(), (s, dom)
    table = dd.PieceTable(lens, doms)
    assert 2 * table.n_domains >= 256 and {n for n in table.lengths if n <= 7} == {3, 4, 5, 6, 7}
    lbs = [dd.LayerBatch(dev[k], 3, 80) for k in range(2)]
    ctx = dd.get_context(torch.cuda.current_device())
    outs = {}
    for name, opts in (('default', {}), ('forced', dict(path=2)), ('forced_unfused', dict(path=2, fuse=0)),
                       ('two_kernels', dict(path=1))):
        with _Options(ctx, **opts):
            outs[name] = dd.quantize_batch(lbs, table).cpu().numpy()
            assert ctx.get_option('last_path') == (1 if name == 'two_kernels' else 2), name
    row = 0
    for s, L in enumerate(lens):
        q = orc.quantize([seen[0][s], seen[1][s]], doms[s], [3, 80, 3, 80])
        for key, exp in q.items():
            assert table.keys[row] == key
            for name, out in outs.items():
                np.testing.assert_array_equal(out[row].astype(np.int64), exp, err_msg=f'{name}: seq {s} (L={L}) domain {key}')
            if storage == 'float32' and s in golden_at:
                gold = gu.expected(CASES[golden_at[s]])[key]
                np.testing.assert_array_equal(outs['default'][row, :240].astype(np.int64), gold.astype(np.int64),
                                              err_msg=golden_at[s])
            row += 1
    assert row == table.n_domains


def _device_batch(torch, lens, D, n_layers, seed):
    """ESM-like synthetic rows generated on the device (bench.py's recipe), one tensor per layer + row offsets."""
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)
    total = int(np.sum(lens))
    layers = []
    for _ in range(n_layers):
        x = torch.randn((total, D), generator=g, device='cuda')
        x = x * torch.exp(torch.randn((1, D), generator=g, device='cuda')) + 5 * torch.randn((1, D), generator=g, device='cuda')
        x[:, ::97] += 200.0
        layers.append(x)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    return layers, offs


def _place(torch, layers, offs, s, case):
    """Writes the inputs of a golden case over sequence s of the batch (a one-layer case: layer 0 only)."""
    xs = gu.build_layers(case)
    for k, x in enumerate(xs):
        layers[k][int(offs[s]):int(offs[s]) + x.shape[0]] = torch.from_numpy(x).cuda()


def _check_goldens(out, table, placed):
    for s, case in placed.items():
        first = list(table.owner).index(s)
        exp = gu.expected(case)
        width = 240 * len(case['layers'])
        for k, key in enumerate(case['keys']):
            assert table.keys[first + k] == key
            np.testing.assert_array_equal(out[first + k, :width].astype(np.int64), exp[key].astype(np.int64),
                                          err_msg=f"{case['id']} {key}")


def _check_oracle(out, table, layers, offs, lens, doms, sample):
    for s in sample:
        a, b = int(offs[s]), int(offs[s]) + int(lens[s])
        q = orc.quantize([x[a:b].cpu().numpy() for x in layers], doms[s], [3, 80] * len(layers))
        first = list(table.owner).index(s)
        for k, (key, exp) in enumerate(q.items()):
            np.testing.assert_array_equal(out[first + k].astype(np.int64), exp, err_msg=f'seq {s} {key}')


def test_config2_shape_batch_default_dispatch_with_goldens_inside(dd):
    """BASELINE config 2: 576 sequences of L = 500, D = 1280, 2 layers in one call -- the bench's kernel
    (walk_ab_kernel, whole-protein variant) under the bench's dispatch; reference goldens at the first, inner and last
    positions; then the same batch with RecCut-shaped multi-domain goldens inside (fused variant)."""
    import torch
    n_seq, L, D = 576, 500, 1280
    lens = [L] * n_seq
    layers, offs = _device_batch(torch, lens, D, 2, 2026)
    placed = {0: CASES['two_L500_D1280'], 1: CASES['grid_gauss_L500_D1280'], 287: CASES['grid_esm_L500_D1280'],
              n_seq - 1: CASES['two_L500_D1280']}
    for s, case in placed.items():
        _place(torch, layers, offs, s, case)
    doms = [['1-500']] * n_seq
    ctx = dd.get_context(torch.cuda.current_device())
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    table = dd.PieceTable.whole_sequences(lens)
    out = dd.quantize_batch(lbs, table).cpu().numpy()
    assert ctx.get_option('last_path') == 2
    _check_goldens(out, table, placed)
    _check_oracle(out, table, layers, offs, lens, doms, (2, 100, 286, 288, 574))
    with _Options(ctx, path=1):
        two = dd.quantize_batch(lbs, table).cpu().numpy()
        assert ctx.get_option('last_path') == 1
    np.testing.assert_array_equal(two, out)
    # fused variant: multi-domain goldens inside the same rows
    doms = [list(d) for d in doms]
    multi = {5: CASES['c4_D1280_L500'], 300: CASES['c4_D1280_L500_17parts'], n_seq - 2: CASES['c4_D1280_L500']}
    for s, case in multi.items():
        _place(torch, layers, offs, s, case)
        doms[s] = list(case['domains'])
    table2 = dd.PieceTable(lens, doms)
    out2 = dd.quantize_batch(lbs, table2).cpu().numpy()
    assert ctx.get_option('last_path') == 2
    _check_goldens(out2, table2, {**placed, **multi})
    _check_oracle(out2, table2, layers, offs, lens, doms, (4, 6, 299, 301))


def test_config3_shape_ragged_batch_default_dispatch_with_goldens_inside(dd):
    """BASELINE config 3: 320 sequences, L in [50, 2000], D = 1280, 2 layers, whole-sequence domains, one call."""
    import torch
    rng = np.random.default_rng(33)
    lens = [int(v) for v in rng.integers(50, 2001, size=320)]
    slots = {0: 'grid_gauss_L2000_D1280', 1: 'grid_esm_L50_D1280', 2: 'two_L1035_D1280', 160: 'grid_esm_L2000_D1280',
             161: 'grid_gauss_L50_D1280', 318: 'two_L50_D1280', 319: 'two_L1035_D1280'}
    for s, cid in slots.items():
        lens[s] = CASES[cid]['layers'][0]['L']
    layers, offs = _device_batch(torch, lens, 1280, 2, 3033)
    placed = {s: CASES[cid] for s, cid in slots.items()}
    for s, case in placed.items():
        _place(torch, layers, offs, s, case)
    ctx = dd.get_context(torch.cuda.current_device())
    table = dd.PieceTable.whole_sequences(lens)
    out = dd.quantize_batch([dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers], table).cpu().numpy()
    assert ctx.get_option('last_path') == 2
    _check_goldens(out, table, placed)
    doms = [[f'1-{L}'] for L in lens]
    shortest, longest = int(np.argmin(lens)), int(np.argmax(lens))
    _check_oracle(out, table, layers, offs, lens, doms, (3, 159, 162, 317, shortest, longest))


def test_one_giant_domain_does_not_take_the_call_off_the_walk_kernel(dd):
    """a single domain above 8 192 rows used to drop the WHOLE call to the two-kernel path.  Now
    dctfp_quantize cuts such domains out into a call of their own; everything else still runs the walk kernel, and every
    fingerprint lands in its own output row (the giant's fused group falls apart: parts and whole protein separately)."""
    import torch
    rng = np.random.default_rng(91)
    lens = [int(v) for v in rng.integers(60, 400, size=300)]
    lens[17] = 9000                          # a titin-sized sequence, whole-sequence domain only
    lens[200] = 8500                         # ... and one given as two parts + the whole protein
    doms = [[f'1-{L}'] for L in lens]
    doms[200] = ['1-4000', '4001-8500', '1-8500']
    doms[5] = ['1-30', '31-' + str(lens[5]), '1-' + str(lens[5])]
    D = 640
    layers, offs = _device_batch(torch, lens, D, 2, 777)
    table = dd.PieceTable(lens, doms)
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    ctx = dd.get_context(torch.cuda.current_device())
    before = ctx.get_option('walk_launches')
    out = dd.quantize_batch(lbs, table).cpu().numpy()
    assert ctx.get_option('walk_launches') == before + 1      # the 300-odd ordinary domains: one walk-kernel launch
    assert ctx.get_option('last_path') == 1                   # ... the two giants after them: two-kernel path
    with _Options(ctx, path=1):
        ref = dd.quantize_batch(lbs, table).cpu().numpy()
    np.testing.assert_array_equal(out, ref)
    _check_oracle(out, table, layers, offs, lens, doms, (5, 16, 17, 18, 200, 299))


@pytest.mark.parametrize('D', [640, 1280, 2560])
def test_matrix_pipe_stage_a_matches_the_vector_path(dd, D):
    """Round-3 experiment kept as an engineering knob of libdctfp_experiments.so (`ab_mfma_a`): the multiply-adds of stage A of
    the fused walks as v_mfma_f64_4x4x4 on 4-row x 64-channel loads, without the first-row shift.  Same bytes as the
    vector path on RecCut-shaped proteins (parts of 3 .. 130 rows that are no multiples of 4, discontinuous domains) --
    including the inputs the shift exists for: channels that are exactly constant over a part, over a whole protein, as
    +0.0 / -0.0, next to a channel that differs from a constant in one bit of one row, and a channel holding a NaN."""
    import torch
    from dctdomain_amd import _lib
    rng = np.random.default_rng(1000 + D)
    lens, doms = [], []
    for i in range(96):
        n_parts = int(rng.integers(2, 7))
        parts = [int(v) for v in (rng.integers(3, 8, size=n_parts) if i % 3 == 0 else rng.integers(9, 131, size=n_parts))]
        edges = np.concatenate([[0], np.cumsum(parts)])
        L = int(edges[-1])
        dl = [f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:])]
        if len(dl) >= 3 and i % 4 == 1:
            dl = [dl[0] + ',' + dl[-1]] + dl[1:-1]            # first + last part as one discontinuous domain
        lens.append(L)
        doms.append(dl + [f'1-{L}'])
    layers, offs = _device_batch(torch, lens, D, 2, 4242 + D)
    first_rows = {}
    for s in range(0, 96, 2):                                  # every other protein carries one kind of special channel
        a, b = int(offs[s]), int(offs[s]) + lens[s]
        pa, pb = (int(v) for v in doms[s][1].split(',')[0].split('-'))     # second domain: one piece
        for k, x in enumerate(layers):
            if s % 8 == 0:                                     # constant over the whole protein (one channel in each half)
                x[a:b, 3] = 1.25 + k
                x[a:b, D - 5] = 7.0
            elif s % 8 == 2:                                   # constant over one part only: the other domains keep their bytes
                x[a + pa - 1:a + pb, 7] = -3.5
            elif s % 8 == 4:                                   # +0.0 / -0.0: equal as numbers, constant
                x[a:b, 11] = 0.0
                x[a:b:2, 11] = -0.0
            else:                                              # one bit in one row: NOT constant; and a NaN
                x[a:b, 19] = 2.0
                x[a + lens[s] // 2, 19] = float(np.nextafter(np.float32(2.0), np.float32(3.0)))
                if s % 16 == 6:
                    x[a + 1, 23] = float('nan')
        first_rows[s] = a
    table = dd.PieceTable(lens, doms)
    assert 2 * table.n_domains >= 256
    lbs = [dd.LayerBatch(x, 3, 80, row_offsets=offs) for x in layers]
    xctx = _lib.experiments_context(torch.cuda.current_device())
    got = {}
    for flag in (0, 1):
        with _Options(xctx, ab_mfma_a=flag, path=2):
            xctx.set_option('degenerate_channels', 0)
            out = dd.quantize_batch(lbs, table, ctx=xctx).cpu().numpy()
            assert xctx.get_option('last_path') == 2
            got[flag] = (out, xctx.get_option('degenerate_channels'))
    assert got[0][1] > 0 and got[1][1] == got[0][1], (got[0][1], got[1][1])      # the same channels were found constant
    np.testing.assert_array_equal(got[1][0], got[0][0])
    # ... and both are the product library's bytes
    np.testing.assert_array_equal(dd.quantize_batch(lbs, table).cpu().numpy(), got[0][0])
