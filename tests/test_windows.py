"""BASELINE config 3 as stated: sequences longer than maxlen exist as the language model's overlapping WINDOWS
(`Embedding.split_seq`, src/embedding.py:83-100), the rows two windows share are averaged
(`run[-olp:] = (run[-olp:] + new[:olp]) / 2`, :185-187) and the result goes through `Fingerprint.quantize`.

`dctfp_quantize_windows` does both in one launch -- the average is taken in the walk kernel's row load, the stitched matrix is
never written.  Checked here against (a) the CPU oracles chained (`stitch_oracle` -> `dct_oracle`, both pinned by
reference-generated goldens) and (b) the materialising form (`dctfp_stitch_sequences` + `dctfp_quantize`), byte for byte,
with `last_path == 2` asserted -- at D = 1280 in batches of >= 256 jobs with the window geometries of the 17 stitch goldens
inside, at D = 640 / 2560, with multi-domain lists whose cuts fall inside the shared rows (fused walks), and for every class of
call the fused path refuses (DCTFP_ERR_UNSUPPORTED -> the caller stitches first)."""

import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle import stitch_oracle as so

with open(os.path.join(gu.GOLD, 'stitch_golden.json')) as fh:
    STITCH_CASES = json.load(fh)['cases']

OVERLAP = 200


def split_lengths(L, maxlen, overlap=OVERLAP):
    """Window lengths of `Embedding.split_seq` / `embed_seq` (src/embedding.py:83-100, :164)."""
    if L <= maxlen:
        return [L]
    return [len(s) for s in so.split_seq('A' * L, maxlen, overlap)]


def test_split_lengths_match_the_reference_goldens():
    """The window geometry this file builds its batches from is the one the reference's own Embedding class produced."""
    for c in STITCH_CASES:
        assert split_lengths(c['L'], c['maxlen']) == c['windows'], c['id']


def test_window_geometry_host_only():
    """dctfp_stitch_sizes through batch.window_geometry: stitched rows per sequence, ValueError where torch would not broadcast."""
    from dctdomain_amd.batch import window_geometry
    rows, counts = [], []
    for c in STITCH_CASES:
        rows += c['windows']
        counts.append(len(c['windows']))
    seq_win, sizes = window_geometry(rows, counts)
    assert sizes.tolist() == [c['embed_shape'][0] for c in STITCH_CASES]
    assert seq_win.tolist() == np.concatenate([[0], np.cumsum(counts)]).tolist()
    with pytest.raises(ValueError):
        window_geometry([500, 200], [2])        # a window not longer than the overlap


def esm_like_windows(torch, gen, rows, D, device='cuda'):
    """One float32 matrix per window; overlapping windows DISAGREE on the rows they share, as a language model's do."""
    scale = torch.exp(torch.randn((1, D), generator=gen, device=device))
    off = 5.0 * torch.randn((1, D), generator=gen, device=device)
    off[0, :: 97] += 200.0
    return [(torch.randn((r, D), generator=gen, device=device) * scale + off).contiguous() for r in rows]


def oracle_stitch(windows, overlap=OVERLAP):
    """src/embedding.py:185-187 with torch-CPU float32 (stitch_oracle, pinned by the reference class's goldens)."""
    return so.stitch_embeddings(windows, overlap)


def _batch(torch, dd, lengths, maxlens, D, seed, n_layers=2):
    """Window matrices of a batch: per layer a flat list of windows, window rows, windows per sequence."""
    gen = torch.Generator(device='cuda')
    gen.manual_seed(seed)
    win_rows, counts = [], []
    for L, ml in zip(lengths, maxlens):
        w = split_lengths(L, ml)
        win_rows += w
        counts.append(len(w))
    layers = [esm_like_windows(torch, gen, win_rows, D) for _ in range(n_layers)]
    return layers, np.asarray(win_rows, dtype=np.int32), np.asarray(counts, dtype=np.int64)


def _check_against_oracle(torch, layers, win_rows, counts, doms, got, keys_of, pick, qdim=(3, 80)):
    from oracle import dct_oracle as orc
    seq_win = np.concatenate([[0], np.cumsum(counts)])
    row = {}
    for r, s in enumerate(keys_of.owner):
        row.setdefault(int(s), r)
    host = got.cpu().numpy()
    for s in pick:
        ws = range(int(seq_win[s]), int(seq_win[s + 1]))
        mats = [oracle_stitch([lay[w].cpu() for w in ws]).numpy() for lay in layers]
        q = orc.quantize(mats, doms[s], list(qdim) * len(layers))
        for k, (key, exp) in enumerate(q.items()):
            assert keys_of.keys[row[s] + k] == key
            np.testing.assert_array_equal(host[row[s] + k].astype(np.int64), exp, err_msg=f'sequence {s} domain {key}')


@pytest.fixture(scope='module')
def dd():
    import torch
    assert torch.cuda.is_available()
    import dctdomain_amd
    return dctdomain_amd


@pytest.mark.gpu
def test_config3_from_windows_whole_sequences_d1280(dd):
    """>= 256 jobs at D = 1280, default dispatch: the stitch goldens' geometries at maxlen 500 (and 1000, 400: every window
    between two others has >= 2 x overlap rows there too) + ragged lengths in [50, 2000]."""
    import torch
    from dctdomain_amd.batch import LayerBatch, PieceTable, quantize_batch, quantize_windows, window_geometry
    from dctdomain_amd.embedding import stitch_embeddings_batch
    rng = np.random.default_rng(11)
    gold = [c for c in STITCH_CASES if c['maxlen'] >= 400]
    lengths = [c['L'] for c in gold] + [int(v) for v in rng.integers(50, 2001, size=150)] + [500, 501, 700, 701, 800, 801, 1100, 1101]
    maxlens = [c['maxlen'] for c in gold] + [500] * 158
    layers, win_rows, counts = _batch(torch, dd, lengths, maxlens, 1280, 5)
    seq_win, sizes = window_geometry(win_rows, counts)
    assert sizes.tolist() == lengths
    table = PieceTable.whole_sequences(sizes)
    assert 2 * table.n_domains >= 256
    ctx = dd.get_context(0)
    before = ctx.get_option('walk_launches')
    lbs = [LayerBatch(lay, 3, 80) for lay in layers]
    got = quantize_windows(lbs, win_rows, counts, table, fallback=False)
    torch.cuda.synchronize()
    assert ctx.get_option('last_path') == 2 and ctx.get_option('walk_launches') == before + 1
    # (b) the materialising form
    stitched = []
    for lay in layers:
        outs = stitch_embeddings_batch([[lay[w] for w in range(int(seq_win[s]), int(seq_win[s + 1]))] for s in range(len(lengths))])
        stitched.append(LayerBatch([outs[s] for s in range(len(lengths))], 3, 80))
    ref = quantize_batch(stitched, table)
    assert torch.equal(got, ref)
    # (a) the oracles chained, on the golden geometries and a sample of the rest
    doms = [[f'1-{L}'] for L in lengths]
    pick = list(range(len(gold))) + [len(gold) + i for i in range(0, 158, 13)] + list(range(len(lengths) - 8, len(lengths)))
    _check_against_oracle(torch, layers, win_rows, counts, doms, got, table, pick)


def _domain_lists(rng, L):
    """RecCut-shaped lists (parts that tile the protein + the whole protein), cuts anywhere -- also inside shared rows --, now
    and then a discontinuous domain, a lone inner domain, or the whole protein alone."""
    kind = rng.integers(0, 6)
    if kind == 0 or L < 40:
        return [f'1-{L}']
    if kind == 1:
        a = int(rng.integers(1, L - 20))
        b = int(rng.integers(a + 5, L + 1))
        return [f'{a}-{b}']
    k = int(rng.integers(2, 7))
    cuts = np.sort(rng.choice(np.arange(8, L - 8), size=k - 1, replace=False))
    cuts = [int(c) for c in cuts if True]
    edges = [0] + cuts + [L]
    edges = [e for i, e in enumerate(edges) if i == 0 or e - edges[i - 1] >= 3]
    if edges[-1] != L:
        edges[-1] = L
    parts = [f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:])]
    if kind == 2 and len(parts) >= 3:
        parts = [parts[0] + ',' + parts[-1]] + parts[1:-1]
    return parts + [f'1-{L}']


@pytest.mark.gpu
@pytest.mark.parametrize('D', [640, 1280, 2560])
def test_windows_multi_domain_fused_walks(dd, D):
    """Parts + whole protein over windows (fused walks read every window row once for both), every width class of the kernel."""
    import torch
    from dctdomain_amd.batch import LayerBatch, PieceTable, quantize_batch, quantize_windows, window_geometry
    from dctdomain_amd.embedding import stitch_embeddings_batch
    rng = np.random.default_rng(100 + D)
    n_seq = 90 if D < 2560 else 60
    lengths = [int(v) for v in rng.integers(230, 1500, size=n_seq)] + [501, 700, 701, 800, 801, 1000]
    layers, win_rows, counts = _batch(torch, dd, lengths, [500] * len(lengths), D, 7 + D)
    seq_win, sizes = window_geometry(win_rows, counts)
    doms = [_domain_lists(rng, L) for L in lengths]
    # cuts exactly at the borders of the shared rows of the fixed-length proteins
    doms[-1] = ['1-300', '301-500', '501-600', '601-800', '801-1000', '1-1000']
    doms[-2] = ['1-299', '300-502', '503-801', '1-801']
    table = PieceTable(sizes, doms)
    assert 2 * table.n_domains >= 256
    ctx = dd.get_context(0)
    lbs = [LayerBatch(lay, 3, 80) for lay in layers]
    got = quantize_windows(lbs, win_rows, counts, table, fallback=False)
    torch.cuda.synchronize()
    assert ctx.get_option('last_path') == 2
    stitched = []
    for lay in layers:
        outs = stitch_embeddings_batch([[lay[w] for w in range(int(seq_win[s]), int(seq_win[s + 1]))] for s in range(len(lengths))])
        stitched.append(LayerBatch([outs[s] for s in range(len(lengths))], 3, 80))
    ref = quantize_batch(stitched, table)
    assert torch.equal(got, ref)
    # unfused (option fuse = 0) gives the same bytes
    ctx.set_option('fuse', 0)
    try:
        assert torch.equal(quantize_windows(lbs, win_rows, counts, table, fallback=False), ref)
    finally:
        ctx.set_option('fuse', 1)
    pick = list(range(0, n_seq, 9 if D < 2560 else 12)) + [len(lengths) - 2, len(lengths) - 1]
    _check_against_oracle(torch, layers, win_rows, counts, doms, got, table, pick)


@pytest.mark.gpu
def test_windows_calls_the_fused_path_refuses(dd):
    """Valid calls that walk_ab_kernel does not take come back DCTFP_ERR_UNSUPPORTED with nothing launched; with
    fallback=True the binding stitches first and the bytes are the oracle's."""
    import torch
    from dctdomain_amd import _lib
    from dctdomain_amd.batch import LayerBatch, PieceTable, quantize_windows, window_geometry
    ctx = dd.get_context(0)

    def run(lengths, maxlen, D, qdim=(3, 80), dtype=None, **kw):
        layers, win_rows, counts = _batch(torch, dd, lengths, [maxlen] * len(lengths), D, 3)
        if dtype is not None:
            layers = [[w.to(dtype) for w in lay] for lay in layers]
        _, sizes = window_geometry(win_rows, counts, OVERLAP)
        table = PieceTable.whole_sequences(sizes)
        lbs = [LayerBatch(lay, *qdim) for lay in layers]
        return layers, win_rows, counts, table, lambda **k2: quantize_windows(lbs, win_rows, counts, table, **{**kw, **k2})

    # (1) a handful of jobs under the default dispatch
    layers, win_rows, counts, table, call = run([700, 801, 1300], 500, 1280)
    with pytest.raises(_lib.DctfpError) as e:
        call(fallback=False)
    assert e.value.code == _lib.DCTFP_ERR_UNSUPPORTED and '256 jobs' in e.value.msg
    got = call()
    _check_against_oracle(torch, layers, win_rows, counts, [[f'1-{int(L)}'] for L in table.seq_rows], got, table, [0, 1, 2])
    # ... which "path" = 2 sends through the fused kernel
    ctx.set_option('path', 2)
    try:
        assert torch.equal(call(fallback=False), got) and ctx.get_option('last_path') == 2
    finally:
        ctx.set_option('path', 0)
    # (2) three windows meet in one row: maxlen 300 (windows every 100 residues) -- the reference class's goldens of that maxlen
    gold300 = [c['L'] for c in STITCH_CASES if c['maxlen'] == 300]
    assert sorted(gold300) == [301, 450, 650]
    layers, win_rows, counts, table, call = run(gold300 * 47, 300, 1280)
    with pytest.raises(_lib.DctfpError) as e:
        call(fallback=False)
    assert e.value.code == _lib.DCTFP_ERR_UNSUPPORTED and 'three windows' in e.value.msg
    _check_against_oracle(torch, layers, win_rows, counts, [[f'1-{int(L)}'] for L in table.seq_rows], call(), table, [0, 1, 2])
    # (3) other kept sizes (PROST's [5, 44]) and a width the kernel does not take
    for D, qdim in ((1280, (5, 44)), (96, (3, 80))):
        layers, win_rows, counts, table, call = run([650, 901] * 70, 500, D, qdim=qdim)
        with pytest.raises(_lib.DctfpError) as e:
            call(fallback=False)
        assert e.value.code == _lib.DCTFP_ERR_UNSUPPORTED
        _check_against_oracle(torch, layers, win_rows, counts, [[f'1-{int(L)}'] for L in table.seq_rows], call(), table, [0, 1], qdim=qdim)
    # (4) half-precision windows: refused (windows are stitched in float32), and the fallback says so too
    layers, win_rows, counts, table, call = run([650, 901] * 70, 500, 1280, dtype=torch.float16)
    with pytest.raises(_lib.DctfpError):
        call(fallback=False)
    with pytest.raises(ValueError):
        call()
    # (5) a window not longer than the overlap: torch would fail to broadcast in the reference
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1)
    lay = esm_like_windows(torch, gen, [500, 200], 1280)
    with pytest.raises(ValueError):
        window_geometry([500, 200], [2])
    arr_rows = np.asarray([500, 200], dtype=np.int32)
    bad_table = PieceTable.whole_sequences([500])
    with pytest.raises(ValueError):
        quantize_windows([LayerBatch(lay, 3, 80)], arr_rows, [2], bad_table)


@pytest.mark.gpu
def test_windows_of_one_window_each_take_every_shape(dd):
    """No shared row anywhere: the call is an ordinary dctfp_quantize with one matrix per sequence, whatever its shape."""
    import torch
    from dctdomain_amd.batch import LayerBatch, PieceTable, quantize_batch, quantize_windows
    for D, qdim, n in ((96, (3, 80), 6), (1280, (5, 44), 300), (640, (3, 80), 300)):
        lengths = [int(v) for v in np.random.default_rng(D).integers(30, 500, size=n)]
        layers, win_rows, counts = _batch(torch, dd, lengths, [500] * n, D, 9, n_layers=1)
        table = PieceTable.whole_sequences(lengths)
        lbs = [LayerBatch(layers[0], *qdim)]
        assert torch.equal(quantize_windows(lbs, win_rows, counts, table, fallback=False), quantize_batch(lbs, table))
