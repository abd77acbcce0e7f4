"""Domain-prediction step: contact top-k selection + in-process RecCut.
CPU part: the oracle and libreccut.so against golden vectors made by the reference's writece and
the reference's own RecCut.cpp; GPU part: the top-k kernel and the Fingerprint.reccut surface."""

import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle import contacts_oracle as co
from recipes import make_input, sha256_of
from recipes_contacts import make_contacts

with open(os.path.join(gu.GOLD, 'reccut_golden.json')) as fh:
    CASES = json.load(fh)['cases']
ARR = np.load(os.path.join(gu.GOLD, 'reccut_golden.npz'))
IDS = [c['id'] for c in CASES]


def build_map(case):
    kw = {k: v for k, v in case['kw'].items() if k not in ('pipeline', 'pipeline_D')}
    cmap = make_contacts(case['recipe'], case['L'], case['seed'], **kw)
    assert sha256_of(cmap) == case['map_sha256']
    return cmap


@pytest.mark.parametrize('case', CASES, ids=IDS)
def test_oracle_selection_and_ce_text(case):
    cmap = build_map(case)
    ci, cj, cv = co.top_contacts(cmap, case['t'])
    assert len(ci) == case['n_contacts']
    np.testing.assert_array_equal(ci, ARR[f"{case['id']}/i"])
    np.testing.assert_array_equal(cj, ARR[f"{case['id']}/j"])
    assert [f'{v:.6f}' for v in cv] == list(ARR[f"{case['id']}/vtext"])
    text = co.ce_text(case['id'], case['seq'], ci, cj, cv)
    assert hashlib.sha256(text.encode()).hexdigest() == case['ce_sha256']
    if 'ce_text' in case:
        assert text == case['ce_text']


@pytest.mark.parametrize('case', CASES, ids=IDS)
def test_libreccut_matches_reference_binary_golden(case):
    from dctdomain_amd import reccut
    cmap = build_map(case)
    ci, cj = ARR[f"{case['id']}/i"], ARR[f"{case['id']}/j"]
    cv = cmap[ci, cj] if len(ci) else np.zeros(0, np.float32)
    doms = reccut.domains_from_contacts([case['L']], [0, len(ci)], ci, cj, cv)[0]
    if len(doms) > 1:
        doms = doms + [f"1-{case['L']}"]
    assert case['reccut_rc'] == 0
    assert doms == case['domains']
    assert co.parse_reccut(case['reccut_stdout'], case['L']) == case['domains']


@pytest.mark.skipif(not os.path.exists(co.REF_BIN), reason='oracle/_ref/RecCut not built')
def test_libreccut_fuzz_against_reference_binary():
    """Random block / interleaved graphs straight into both implementations."""
    from dctdomain_amd import reccut
    rng = np.random.default_rng(20240)
    n_multi = n_disc = 0
    for _ in range(120):
        L = int(rng.integers(22, 360))
        nb = int(rng.integers(1, 8))
        bounds = np.sort(rng.choice(np.arange(1, L), size=min(nb - 1, L - 1), replace=False)) if nb > 1 else []
        lab = np.zeros(L, int)
        for b in bounds:
            lab[b:] += 1
        for _m in range(int(rng.integers(0, 3))):
            if nb >= 3:
                a, b = sorted(rng.choice(nb, 2, replace=False))
                lab[lab == b] = a
        same = lab[:, None] == lab[None, :]
        p = rng.random((L, L)) * (same * rng.uniform(0.5, 1.0) + (~same) * rng.uniform(0.0, 0.3))
        mask = np.triu(rng.random((L, L)) < rng.uniform(0.02, 0.3), 5)
        ii, jj = np.nonzero(mask)
        t = int(2.6 * L)
        if len(ii) > t:
            order = np.argsort(-p[ii, jj], kind='stable')[:t]
            ii, jj = ii[order], jj[order]
        pv = p[ii, jj].astype(np.float32)
        text = co.ce_text('x', 'A' * L, ii, jj, pv)
        rc, out = co.run_ref_binary(text)
        assert rc == 0
        exp = out.strip().split()[2].split(';')[:-1]
        got = reccut.domains_from_contacts([L], [0, len(ii)], ii, jj, pv)[0]
        assert got == exp, (L, exp, got)
        n_multi += len(exp) > 1
        n_disc += any(',' in d for d in exp)
    assert n_multi > 30 and n_disc > 3


def test_libreccut_batch_threads():
    from dctdomain_amd import reccut
    sel = [c for c in CASES if c['L'] >= 100][:8]
    n_res, offs, ci, cj, cv = [], [0], [], [], []
    for c in sel:
        cmap = build_map(c)
        i, j = ARR[f"{c['id']}/i"], ARR[f"{c['id']}/j"]
        n_res.append(c['L']); ci.append(i); cj.append(j); cv.append(cmap[i, j]); offs.append(offs[-1] + len(i))
    args = (n_res, offs, np.concatenate(ci), np.concatenate(cj), np.concatenate(cv))
    one = reccut.domains_from_contacts(*args, threads=1)
    four = reccut.domains_from_contacts(*args, threads=4)
    assert one == four
    for c, d in zip(sel, one):
        assert (d + [f"1-{c['L']}"] if len(d) > 1 else d) == c['domains']


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=IDS)
def test_gpu_topk_writece_reccut(case, tmp_path):
    import dctdomain_amd as dd
    from dctdomain_amd import reccut
    cmap = build_map(case)
    fp = dd.Fingerprint(pid=case['id'], seq=case['seq'], contacts=cmap)
    fn = str(tmp_path / 'x.ce')
    fp.writece(fn, case['t'])
    text = open(fn).read()
    assert hashlib.sha256(text.encode()).hexdigest() == case['ce_sha256']
    if abs(case['t'] - 2.6) < 1e-12:
        fp.reccut(case['t'])
        assert fp.domains == case['domains']
    # the selection itself, entry by entry
    import torch
    offs, ci, cj, cv = reccut.top_contacts_batch([torch.from_numpy(cmap).cuda()], case['t'])
    oi, oj, ov = co.top_contacts(cmap, case['t'])
    np.testing.assert_array_equal(ci, oi)
    np.testing.assert_array_equal(cj, oj)
    np.testing.assert_array_equal(cv, ov)


@pytest.mark.gpu
def test_gpu_queue_cpu_pipeline_matches_reference():
    """make_db.queue_cpu (src/make_db.py:29-30): reccut(2.6) then quantize([3,80,3,80])."""
    import dctdomain_amd as dd
    n = 0
    for case in CASES:
        if 'pipeline' not in case:
            continue
        cmap = build_map(case)
        pl = case['pipeline']
        e1 = make_input('esm', case['L'], pl['D'], pl['seeds'][0])
        e2 = make_input('esm', case['L'], pl['D'], pl['seeds'][1])
        assert [sha256_of(e1), sha256_of(e2)] == pl['sha']
        fp = dd.Fingerprint(pid=case['id'], seq=case['seq'], embed={15: e1, 21: e2}, contacts=cmap)
        fp.reccut(2.6)
        fp.quantize([3, 80, 3, 80])
        assert fp.domains == pl['keys']
        for k_i, key in enumerate(pl['keys']):
            np.testing.assert_array_equal(fp.quants[key], ARR[f"{case['id']}/q/{k_i}"].astype(np.int64))
        n += 1
    assert n >= 6


@pytest.mark.gpu
def test_gpu_topk_batch_mixed_lengths():
    import torch
    from dctdomain_amd import reccut
    maps = [make_contacts(r, L, 900 + i) for i, (r, L) in enumerate(
        [('blocks', 3), ('ties', 77), ('sparse', 640), ('flat', 50), ('blocks', 1035), ('negzero', 64), ('blocks', 6),
         # long proteins: a million candidate pairs and more go through the multi-workgroup selection
         ('blocks', 1700), ('flat', 1500), ('sparse', 2300), ('ties', 1601), ('negzero', 1460), ('blocks', 40)])]
    offs, ci, cj, cv = reccut.top_contacts_batch([torch.from_numpy(m).cuda() for m in maps], 2.6)
    for p, m in enumerate(maps):
        oi, oj, ov = co.top_contacts(m, 2.6)
        a, b = offs[p], offs[p + 1]
        np.testing.assert_array_equal(ci[a:b], oi)
        np.testing.assert_array_equal(cj[a:b], oj)
        np.testing.assert_array_equal(cv[a:b], ov)


@pytest.mark.gpu
def test_gpu_topk_two_reads_bands_ties_plateaus_and_large_k():
    """contact_topk2_kernel (round 4: thread minima bound the k-th value, second read collects, bisection selects): maps with
    the contacts in a band along the diagonal (what a real map looks like: the lane rotation must spread it), values
    quantised to a few levels (ties at the threshold, more candidates than the LDS holds -> the radix select behind it),
    one plateau, negative values and -0.0; t from 0.5 to 15 (k up to 7 500 > 6 144: handed to the radix select) -- entry
    by entry against the oracle, in the reference's order."""
    import torch
    from dctdomain_amd import reccut
    rng = np.random.default_rng(77)
    maps = []
    for L in (6, 30, 64, 200, 500, 777, 1400):
        i, j = np.indices((L, L))
        band = np.exp(-np.abs(i - j) / 6.0) * (0.6 + 0.4 * rng.random((L, L)))
        maps.append(band.astype(np.float32))                                        # near-diagonal band
        maps.append(rng.random((L, L)).astype(np.float32))                           # no structure, no ties
        maps.append((np.floor(rng.random((L, L)) * 7) / 7).astype(np.float32))      # seven levels: huge tie classes
        maps.append((rng.standard_normal((L, L)) * (rng.random((L, L)) < 0.3)).astype(np.float32) * np.float32(-1.0))  # zeros, -0.0, negatives
    maps.append(np.full((300, 300), 0.25, dtype=np.float32))                         # one plateau
    dev = [torch.from_numpy(m).cuda() for m in maps]
    for t in (0.5, 2.6, 6.0, 15.0):
        offs, ci, cj, cv = reccut.top_contacts_batch(dev, t)
        for p, m in enumerate(maps):
            oi, oj, ov = co.top_contacts(m, t)
            a, b = offs[p], offs[p + 1]
            np.testing.assert_array_equal(ci[a:b], oi, err_msg=f'map {p} (L = {m.shape[0]}), t = {t}')
            np.testing.assert_array_equal(cj[a:b], oj, err_msg=f'map {p} (L = {m.shape[0]}), t = {t}')
            np.testing.assert_array_equal(cv[a:b].view(np.uint32), ov.view(np.uint32), err_msg=f'map {p} (L = {m.shape[0]}), t = {t}')


def test_contact_weight_matches_text_round_trip():
    """reccut_contact_weight == (int)(strtod("%.6f" % p) * 100 + 0.5) (src/fingerprint.py:72 + src/RecCut.cpp:384),
    checked on random probabilities and on values hugging every rounding boundary."""
    import ctypes as C
    from dctdomain_amd import _lib
    lib = _lib.load_reccut()
    lib.reccut_contact_weight.argtypes = [C.c_float]
    lib.reccut_contact_weight.restype = C.c_int32
    rng = np.random.default_rng(1)
    edge = (np.arange(0, 101)[:, None] / 100.0 - 0.005 + np.linspace(-3e-6, 3e-6, 61)[None, :]).ravel().astype(np.float32)
    vals = np.concatenate([rng.random(20000).astype(np.float32), edge, np.nextafter(edge, np.float32(1)),
                           np.nextafter(edge, np.float32(0)),
                           np.array([0, 1, 0.5, 1e-7, 5e-3, 0.995, -0.2, -1e-9, 2.0], dtype=np.float32)])
    for p in vals:
        assert lib.reccut_contact_weight(float(p)) == int(float('%.6f' % float(p)) * 100 + 0.5), float(p)


# ------------------------------------------------------------------ the cutter's recursion on the GPU (dctfp_reccut, round 5)
def _random_graph(rng, L):
    """Block / interleaved contact lists as in the host library's fuzz: (ii, jj, float32 probabilities), pairs distinct."""
    nb = int(rng.integers(1, 8))
    bounds = np.sort(rng.choice(np.arange(1, L), size=min(nb - 1, L - 1), replace=False)) if nb > 1 else []
    lab = np.zeros(L, int)
    for b in bounds:
        lab[b:] += 1
    for _m in range(int(rng.integers(0, 3))):
        if nb >= 3:
            a, b = sorted(rng.choice(nb, 2, replace=False))
            lab[lab == b] = a
    same = lab[:, None] == lab[None, :]
    p = rng.random((L, L)) * (same * rng.uniform(0.5, 1.0) + (~same) * rng.uniform(0.0, 0.3))
    if rng.random() < 0.3:
        p = np.round(p * 8) / 8      # few distinct weights: ties between cut scores
    mask = np.triu(rng.random((L, L)) < rng.uniform(0.02, 0.3), int(rng.integers(1, 7)))
    ii, jj = np.nonzero(mask)
    t = int(2.6 * L)
    if len(ii) > t:
        order = np.argsort(-p[ii, jj], kind='stable')[:t]
        ii, jj = ii[order], jj[order]
    return ii.astype(np.int32), jj.astype(np.int32), p[ii, jj].astype(np.float32)


@pytest.mark.gpu
def test_gpu_reccut_goldens():
    """The reference binary's domain strings (60 goldens) from the recursion on the GPU, in one batch; only what the kernel's
    tables do not hold (L > 2 048) may come back through the host library."""
    import torch
    from dctdomain_amd import reccut
    sel = [c for c in CASES if abs(c['t'] - 2.6) < 1e-12 and c['reccut_rc'] == 0]
    maps = [torch.from_numpy(build_map(c)).cuda() for c in sel]
    doms = reccut.domains_from_maps(maps, 2.6)
    redone = [sel[p]['L'] for p in reccut.LAST.host_redo]
    assert all(L > 2048 for L in redone), redone
    for c, d in zip(sel, doms):
        assert (d + [f"1-{c['L']}"] if len(d) > 1 else d) == c['domains'], c['id']
    # ... and from the goldens' own contact lists (no selection kernel in front)
    n_res, offs, ci, cj, cv = [], [0], [], [], []
    for c in sel:
        cmap = build_map(c)
        i, j = ARR[f"{c['id']}/i"], ARR[f"{c['id']}/j"]
        n_res.append(c['L']); ci.append(i); cj.append(j); cv.append(cmap[i, j]); offs.append(offs[-1] + len(i))
    got = reccut.domains_from_contacts_gpu(n_res, offs, np.concatenate(ci), np.concatenate(cj), np.concatenate(cv))
    assert got == doms


@pytest.mark.gpu
def test_gpu_reccut_fuzz_against_host_library():
    """4 000 random graphs (blocks, interleaved blocks, tie-rich weights, contacts inside the band, L = 22 .. 700, a few up to
    1 500 and up to 2 048) through both: identical strings; the host library itself is held to the reference's binary by the tests above."""
    from dctdomain_amd import reccut
    rng = np.random.default_rng(777)
    n_multi = n_disc = n_total = 0
    for batch in range(8):
        n_res, offs, ci, cj, cv = [], [0], [], [], []
        for k in range(500):
            # (every LDS class of the kernel: 512 / 1 024 / 1 536 / 2 048 residues; its scan runs in 1 .. 4 row bands by node size)
            L = int(rng.integers(22, 700)) if k % 50 else (int(rng.integers(700, 1500)) if k % 100 else int(rng.integers(1500, 2049)))
            ii, jj, pv = _random_graph(rng, L)
            n_res.append(L); ci.append(ii); cj.append(jj); cv.append(pv); offs.append(offs[-1] + len(ii))
        args = (n_res, offs, np.concatenate(ci), np.concatenate(cj), np.concatenate(cv))
        exp = reccut.domains_from_contacts(*args, threads=8)
        got = reccut.domains_from_contacts_gpu(*args)
        assert reccut.LAST.host_redo == [], [n_res[p] for p in reccut.LAST.host_redo]
        for p, (e, g) in enumerate(zip(exp, got)):
            assert e == g, (batch, p, n_res[p], e, g)
        n_multi += sum(len(e) > 1 for e in exp)
        n_disc += sum(any(',' in d for d in e) for e in exp)
        n_total += len(exp)
    assert n_total == 4000 and n_multi > 1000 and n_disc > 100


def test_contact_weight_exact_is_the_text_round_trip():
    """(int)(strtod("%.6f" % p) * 100 + 0.5) without the text: what dctfp_reccut's kernel computes per contact."""
    from dctdomain_amd import _lib
    lib = _lib.load_reccut()
    rng = np.random.default_rng(1)
    n = rng.integers(0, 100, 20000)
    b = (n + 0.5) / 100
    m = rng.integers(0, 1000000, 20000)
    vals = np.concatenate([rng.random(20000), b + rng.normal(0, 3e-7, len(b)), b + rng.integers(-3, 4, len(b)) * 5e-8,
                           (m + 0.5) / 1e6 + rng.integers(-2, 3, len(m)) * 1e-9, -rng.random(500),
                           [1 / 128, 3 / 128, 5 / 256, 0.0, 1.0, 0.999999, 0.9999995, 0.5, 0.005, 0.015, 0.025, 1e-7]]).astype(np.float32)
    for p in vals.tolist():
        assert lib.reccut_contact_weight_exact(p) == int(float('%.6f' % p) * 100 + 0.5) == lib.reccut_contact_weight(p), p
