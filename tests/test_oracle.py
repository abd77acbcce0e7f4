"""The oracle (oracle/dct_oracle.py) against the golden vectors made by the reference itself."""

import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle import dct_oracle as orc

ALL_OK = gu.cases(expect='ok')
ALL_ERR = gu.cases(expect='ValueError')


@pytest.mark.parametrize('case', ALL_OK, ids=gu.case_ids(ALL_OK))
def test_faithful_form_bit_exact(case):
    layers = gu.build_layers(case)
    got = orc.quantize(layers, case['domains'], case['qdim'])
    exp = gu.expected(case)
    assert list(got.keys()) == case['keys']
    for k in exp:
        np.testing.assert_array_equal(got[k], exp[k].astype(np.int64), err_msg=f"{case['id']} {k}")


def _ill_conditioned(case):
    # constant channels / non-finite inputs: the reference's own output depends on
    # pocketfft round-off noise; only the faithful form can be expected to follow it.
    return any('patch' in sp for sp in case['layers'])


@pytest.mark.parametrize('case', ALL_OK, ids=gu.case_ids(ALL_OK))
def test_matrix_form_bit_exact(case):
    layers = gu.build_layers(case)
    got = orc.quantize_matrix(layers, case['domains'], case['qdim'])
    exp = gu.expected(case)
    assert list(got.keys()) == case['keys']
    for k in exp:
        np.testing.assert_array_equal(got[k], exp[k].astype(np.int64), err_msg=f"{case['id']} {k}")


@pytest.mark.parametrize('case', ALL_ERR, ids=gu.case_ids(ALL_ERR))
def test_errors(case):
    layers = gu.build_layers(case)
    with pytest.raises(ValueError):
        orc.quantize(layers, case['domains'], case['qdim'])
    with pytest.raises(ValueError):
        orc.quantize_matrix(layers, case['domains'], case['qdim'])


def test_intermediates_match_reference():
    n_checked = 0
    for case in ALL_OK:
        if not case.get('intermediates'):
            continue
        layers = gu.build_layers(case)
        arr = gu.arrays()
        x, _ = orc.get_doms(layers[0], case['domains'][0])
        n, m = case['qdim'][0], case['qdim'][1]
        np.testing.assert_allclose(orc.coefficients(x, n), arr[f"{case['id']}/coef"], rtol=0, atol=1e-12)
        _, im = orc.quantize_layer_matrix(x, n, m, want_intermediates=True)
        np.testing.assert_allclose(im['Yp'], arr[f"{case['id']}/Yp"], rtol=0, atol=1e-11)
        np.testing.assert_allclose(im['Z'], arr[f"{case['id']}/Z"], rtol=0, atol=1e-10)
        # true ortho coefficients from the matrix form
        c = (orc.dct2_ortho_matrix(n, x.shape[0]) @ x).T
        scale = max(1.0, np.abs(arr[f"{case['id']}/coef"]).max())
        np.testing.assert_allclose(c, arr[f"{case['id']}/coef"], rtol=0, atol=1e-12 * scale)
        n_checked += 1
    assert n_checked >= 6


def test_get_doms_table():
    with open(os.path.join(gu.GOLD, 'getdoms_golden.json')) as fh:
        table = json.load(fh)
    for row in table:
        x = np.arange(row['L'] * 4, dtype=np.float32).reshape(row['L'], 4)
        mat, key = orc.get_doms(x, row['dom'])
        assert key == row['key'], row
        assert mat.dtype == np.float64
        assert [int(v) for v in (mat[:, 0] / 4).astype(int)] == row['rows'], row


# ---------------------------------------------------------------------------
# plain-C restatement (oracle/dct_oracle.c -> oracle/_build/liboracle.so)
# ---------------------------------------------------------------------------
import ctypes as C
import subprocess


def _c_oracle():
    here = os.path.join(os.path.dirname(gu.GOLD), '..', 'oracle')
    so = os.path.abspath(os.path.join(here, '_build', 'liboracle.so'))
    if not os.path.exists(so):
        subprocess.run(['make', '-C', os.path.abspath(here), '_build/liboracle.so'], check=True)
    lib = C.CDLL(so)
    lib.oracle_quantize_layer.restype = C.c_int
    lib.oracle_quantize_layer.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def c_quantize(lib, layers, domains, qdim):
    quants = {}
    for i, embed in enumerate(layers):
        n, m = qdim[2 * i], qdim[2 * i + 1]
        for dom in domains:
            pieces, key = orc.split_domain(dom, embed.shape[0])
            rows = [embed[b:e] for b, e in pieces]
            x = np.ascontiguousarray(np.concatenate(rows, axis=0)) if rows else np.zeros((0, embed.shape[1]), np.float32)
            if not x.size:
                continue
            out = np.zeros(n * m, np.int8)
            rc = lib.oracle_quantize_layer(x.ctypes.data, x.shape[0], x.shape[1], x.shape[1], n, m,
                                           out.ctypes.data, None, None, None)
            if rc == -2:
                raise ValueError('reshape')
            assert rc == 0
            quants.setdefault(key, []).extend(out.astype(np.int64).tolist())
    return {k: np.array(v) for k, v in quants.items()}


C_CASES = [c for c in ALL_OK if all(sp['D'] * sp['L'] <= 1300 * 1100 for sp in c['layers'])]


@pytest.mark.parametrize('case', C_CASES, ids=gu.case_ids(C_CASES))
def test_c_form_bit_exact(case):
    lib = _c_oracle()
    layers = gu.build_layers(case)
    got = c_quantize(lib, layers, case['domains'], case['qdim'])
    exp = gu.expected(case)
    assert list(got.keys()) == case['keys']
    for k in exp:
        np.testing.assert_array_equal(got[k], exp[k].astype(np.int64), err_msg=f"{case['id']} {k}")


def test_c_form_errors_and_intermediates():
    lib = _c_oracle()
    for case in ALL_ERR:
        layers = gu.build_layers(case)
        with pytest.raises(ValueError):
            c_quantize(lib, layers, case['domains'], case['qdim'])
    arr = gu.arrays()
    n_checked = 0
    for case in ALL_OK:
        if not case.get('intermediates'):
            continue
        layers = gu.build_layers(case)
        xd, _ = orc.get_doms(layers[0], case['domains'][0])
        x = np.ascontiguousarray(xd.astype(np.float32))
        n, m = case['qdim'][0], case['qdim'][1]
        out = np.zeros(n * m, np.int8)
        coef = np.zeros((x.shape[1], n)); yp = np.zeros((n, x.shape[1])); z = np.zeros((n, m))
        rc = lib.oracle_quantize_layer(x.ctypes.data, x.shape[0], x.shape[1], x.shape[1], n, m, out.ctypes.data,
                                       coef.ctypes.data, yp.ctypes.data, z.ctypes.data)
        assert rc == 0
        scale = max(1.0, np.abs(arr[f"{case['id']}/coef"]).max())
        np.testing.assert_allclose(coef, arr[f"{case['id']}/coef"], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(yp, arr[f"{case['id']}/Yp"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(z, arr[f"{case['id']}/Z"], rtol=0, atol=1e-8)
        n_checked += 1
    assert n_checked >= 6
