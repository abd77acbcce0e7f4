"""Fences around the inputs where the GPU path does not claim bit-exactness.  The golden data
(tests/golden/fence_golden.*) is made by the imported reference (tests/golden/make_golden_fences.py):

1. exactly constant channel -- mathematically 0/0 = NaN -> the (layer, domain) block is 0.  The reference gets there at
   1773 of the lengths 3..2000 and scales pocketfft round-off noise at the other 225.  The GPU gives the 0 block at EVERY
   length and counts the channel (option "degenerate_channels"), so a caller can tell;
2. D == m -- the channel resample is an identity up to round-off: values at the ties 126|127 may differ by 1;
3. a domain whose rows repeat exactly -- the k = 2 coefficient is a round-off zero in the reference: the middle resampled
   row (values 80..159 of a layer block) is noise-determined there, the two outer rows are not.
The tests pin exactly that behaviour, so that it cannot drift unnoticed."""

import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle import dct_oracle as orc
from recipes import make_input, sha256_of

with open(os.path.join(gu.GOLD, 'fence_golden.json')) as fh:
    DOC = json.load(fh)
ARR = np.load(os.path.join(gu.GOLD, 'fence_golden.npz'))


def _const_input(L):
    cc = DOC['const']
    x = make_input(cc['recipe'], L, cc['D'], cc['seed0'] + L)
    x[:, cc['col']] = np.float32(cc['value'])
    return x


def test_fence_lists_are_complete():
    cc = DOC['const']
    assert sorted(cc['zero_L'] + cc['noisy_L']) == list(range(3, 2001))
    assert len(cc['zero_L']) == 1773 and len(cc['noisy_L']) == 225


def test_oracle_reproduces_the_reference_on_constant_channels():
    """The faithful oracle (scipy's own transform) lands on the same side as the reference at every length probed."""
    cc = DOC['const']
    for L in cc['noisy_L'][:40] + cc['zero_L'][:40] + cc['zero_L'][-5:] + cc['noisy_L'][-5:]:
        q = orc.quantize([_const_input(L)], [f'1-{L}'], cc['qdim'])[f'1-{L}']
        assert bool(q.any()) == (L in cc['noisy_L']), L


@pytest.mark.gpu
def test_gpu_constant_channel_every_length_zero_block_and_counted():
    import torch
    import dctdomain_amd as dd
    cc = DOC['const']
    ctx = dd.get_context(torch.cuda.current_device())
    ctx.set_option('degenerate_channels', 0)
    Ls = list(range(3, 2001))
    xs = [torch.from_numpy(_const_input(L)).cuda() for L in Ls]
    out = dd.quantize_batch([dd.LayerBatch(xs, 3, 80)], dd.PieceTable.whole_sequences(Ls)).cpu().numpy()
    assert not out.any()                                        # the 0 block at every length (= the reference at 1773 of them)
    assert ctx.get_option('degenerate_channels') == len(Ls)    # one constant channel per sequence, each reported
    # the same through the walk kernel (D = 640) and without the constant channel nothing is counted
    ctx.set_option('degenerate_channels', 0)
    Lw = [50, 77, 500]
    xw = []
    for L in Lw:
        x = make_input('gauss', L, 640, 700 + L)
        x[:, 11] = np.float32(-3.25)
        x[:, 600] = np.float32(0.0)
        xw.append(torch.from_numpy(x).cuda())
    ow = dd.quantize_batch([dd.LayerBatch(xw, 3, 80)], dd.PieceTable.whole_sequences(Lw)).cpu().numpy()
    assert not ow.any() and ctx.get_option('degenerate_channels') == 2 * len(Lw)
    ctx.set_option('degenerate_channels', 0)
    clean = [torch.from_numpy(make_input('esm', L, 640, 800 + L)).cuda() for L in Lw]
    oc = dd.quantize_batch([dd.LayerBatch(clean, 3, 80)], dd.PieceTable.whole_sequences(Lw)).cpu().numpy()
    assert oc.any() and ctx.get_option('degenerate_channels') == 0


@pytest.mark.gpu
@pytest.mark.parametrize('case', DOC['d_equals_m'], ids=[c['id'] for c in DOC['d_equals_m']])
def test_gpu_d_equals_m_differs_by_at_most_one(case):
    import dctdomain_amd as dd
    x = make_input(case['recipe'], case['L'], case['D'], case['seed'])
    assert sha256_of(x) == case['sha256']
    fp = dd.Fingerprint(pid='f', seq='A' * case['L'], embed={0: x}, domains=[case['domain']])
    fp.quantize(case['qdim'])
    got, exp = fp.quants[case['key']], ARR[case['id'] + '/out'].astype(np.int64)
    d = np.abs(got - exp)
    assert d.max() <= 1
    assert set(np.unique(np.concatenate([got[d > 0], exp[d > 0]]))) <= {126, 127}      # only at the tie under the maximum
    assert ((got == 0) == (exp == 0)).all()


@pytest.mark.gpu
@pytest.mark.parametrize('case', DOC['repeated_rows'], ids=[c['id'] for c in DOC['repeated_rows']])
def test_gpu_repeated_rows_outer_rows_match(case):
    import torch
    import dctdomain_amd as dd
    x = make_input(case['recipe'], case['L'], case['D'], case['seed'])
    assert sha256_of(x) == case['sha256']
    fp = dd.Fingerprint(pid='f', seq='A' * case['L'], embed={0: x}, domains=[case['domain']])
    fp.quantize(case['qdim'])
    got, exp = fp.quants[case['key']], ARR[case['id'] + '/out'].astype(np.int64)
    n, m = case['qdim']
    np.testing.assert_array_equal(got[:m], exp[:m])                    # first resampled row
    np.testing.assert_array_equal(got[(n - 1) * m:], exp[(n - 1) * m:])  # last resampled row
    # (the middle row is where the reference scales round-off noise: nothing to compare it with)
    ctx = dd.get_context(torch.cuda.current_device())
    old = ctx.get_option('path')
    try:
        ctx.set_option('path', 1)
        fp2 = dd.Fingerprint(pid='f', seq='A' * case['L'], embed={0: x}, domains=[case['domain']])
        fp2.quantize(case['qdim'])
    finally:
        ctx.set_option('path', old)
    got1 = fp2.quants[case['key']]
    np.testing.assert_array_equal(got1[:m], got[:m])                  # (the two kernels sum in different orders: their
    np.testing.assert_array_equal(got1[(n - 1) * m:], got[(n - 1) * m:])  #  middle rows are their own round-off, too)
