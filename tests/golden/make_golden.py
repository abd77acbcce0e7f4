#!/usr/bin/env python3
"""Generates the golden vectors of tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference checkout does not exist on the
GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``Fingerprint`` from /root/reference/src/fingerprint.py (numpy +
scipy only), runs ``quantize`` on the seed-defined inputs of ``recipes.py`` and
stores inputs' sha256 + expected outputs (data only; no reference source is
copied).  Outputs:

    quantize_golden.json   case manifest (recipes, domains, qdim, expected keys / errors)
    quantize_golden.npz    int8 outputs, float64 intermediates, a few tiny full inputs
    getdoms_golden.json    get_doms() behaviour table (rows gathered, cleaned key)
"""

from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference/src')

from fingerprint import Fingerprint  # noqa: E402  (the reference itself)
from scipy.fft import dct  # noqa: E402
from recipes import make_input, sha256_of  # noqa: E402

warnings.simplefilter('ignore', RuntimeWarning)

arrays = {}
cases = []


def run_reference(layers, domains, qdim):
    """layers: list of float32 arrays. Returns (status, keys, out[int8 (n_keys, total)])."""
    embed = {i: x for i, x in enumerate(layers)}
    seq_len = layers[0].shape[0]
    fp = Fingerprint(pid='g', seq='A' * seq_len, embed=embed, domains=list(domains))
    try:
        fp.quantize(list(qdim))
    except ValueError as exc:
        return 'ValueError', [], None, str(exc)
    keys = list(fp.quants.keys())
    assert keys == fp.domains
    out = []
    for k in keys:      # rows may be ragged: duplicate keys are extended twice (fingerprint.py:196)
        v = np.asarray(fp.quants[k]).astype(np.int64)
        assert v.min() >= 0 and v.max() <= 127
        out.append(v.astype(np.int8))
    return 'ok', keys, out, ''


def add_case(cid, layer_specs, domains, qdim, inline=False, intermediates=False):
    layers = []
    spec_out = []
    for li, sp in enumerate(layer_specs):
        x = make_input(sp['recipe'], sp['L'], sp['D'], sp['seed'])
        if 'patch' in sp:
            for op in sp['patch']:
                if op[0] == 'const_col':
                    x[:, op[1]] = np.float32(op[2])
                elif op[0] == 'zero_all':
                    x[:] = 0
                elif op[0] == 'nan_at':
                    x[op[1], op[2]] = np.nan
                elif op[0] == 'inf_at':
                    x[op[1], op[2]] = np.inf
        layers.append(x)
        s = dict(sp)
        s['sha256'] = sha256_of(x)
        if inline:
            key = f'{cid}/x{li}'
            arrays[key] = x
            s['inline'] = key
        spec_out.append(s)
    status, keys, out, msg = run_reference(layers, domains, qdim)
    case = {'id': cid, 'layers': spec_out, 'domains': list(domains), 'qdim': list(qdim),
            'expect': status, 'keys': keys, 'error': msg}
    if status == 'ok':
        for j, v in enumerate(out):
            arrays[f'{cid}/out/{j}'] = v
    if intermediates and status == 'ok':
        # layer 0, first input domain only, through the reference's own methods
        fp = Fingerprint(pid='g', seq='A' * layers[0].shape[0], embed={0: layers[0]})
        dom_emb, _ = fp.get_doms(layers[0], domains[0])
        n, m = qdim[0], qdim[1]
        arrays[f'{cid}/coef'] = dct(dom_emb.T, type=2, norm='ortho')[:, :n].copy()   # fingerprint.py:137
        yp = fp.idct_quant(dom_emb, n)                                               # (n, D)
        arrays[f'{cid}/Yp'] = np.array(yp)
        arrays[f'{cid}/Z'] = np.array(fp.idct_quant(yp.T, m).T)                      # (n, m) scaled
        case['intermediates'] = True
    cases.append(case)
    return case


seed_ctr = [1000]


def nseed():
    seed_ctr[0] += 1
    return seed_ctr[0]


# 1. size grid, one layer, whole-sequence domain -------------------------------
LS = [3, 4, 5, 22, 23, 50, 81, 158, 196, 374, 500, 501, 700, 1035, 2000]
for D in (96, 640, 1280, 2560):
    for L in LS:
        for recipe in ('gauss', 'esm'):
            if D == 2560 and L > 700 and recipe == 'gauss':
                continue
            add_case(f'grid_{recipe}_L{L}_D{D}', [dict(recipe=recipe, L=L, D=D, seed=nseed())],
                     [f'1-{L}'], [3, 80], intermediates=(L in (5, 50, 500) and D in (96, 1280)))

# 2. two layers (the production qdim), incl. the headline shape -----------------
for D in (640, 1280, 2560):
    for L in (50, 158, 374, 500, 1035):
        add_case(f'two_L{L}_D{D}',
                 [dict(recipe='esm', L=L, D=D, seed=nseed()), dict(recipe='esm', L=L, D=D, seed=nseed())],
                 [f'1-{L}'], [3, 80, 3, 80])

# 3. other qdims (PROST-style and extremes supported by the kernels) -----------
for qd in ([5, 44], [3, 85], [4, 64], [8, 128], [2, 16], [6, 100], [7, 33]):
    for (L, D) in ((64, 640), (300, 1280), (97, 200)):
        add_case(f'qdim_{qd[0]}x{qd[1]}_L{L}_D{D}', [dict(recipe='esm', L=L, D=D, seed=nseed())],
                 [f'1-{L}'], qd)
add_case('qdim_mixed', [dict(recipe='esm', L=120, D=640, seed=nseed()), dict(recipe='gauss', L=120, D=640, seed=nseed())],
         ['1-120', '5-60'], [5, 44, 3, 80])
add_case('qdim_n1', [dict(recipe='gauss', L=40, D=128, seed=nseed())], ['1-40'], [1, 80])

# 4. multi-domain, discontinuous, get_doms quirks --------------------------------
add_case('dom_multi', [dict(recipe='esm', L=158, D=640, seed=nseed()), dict(recipe='esm', L=158, D=640, seed=nseed())],
         ['1-81', '82-158', '1-158'], [3, 80, 3, 80])
add_case('dom_discont', [dict(recipe='esm', L=100, D=640, seed=nseed()), dict(recipe='gauss', L=100, D=640, seed=nseed())],
         ['1-30,61-100', '31-60', '1-100'], [3, 80, 3, 80])
add_case('dom_quirks', [dict(recipe='gauss', L=100, D=128, seed=nseed())],
         ['1-100', '1-30,61-100', '90-120', '101-120', '1-30,101-120', '1-30,101-120,40-50', '0-10',
          '7-9', '50-40,1-10', '20-25,22-27', '3-5', '150-160,200-300,1-5,200-300', '0-100,1-2'],
         [3, 80])
add_case('dom_quirk_0-100', [dict(recipe='gauss', L=100, D=128, seed=nseed())], ['0-100'], [3, 80])
add_case('dom_many', [dict(recipe='esm', L=500, D=1280, seed=nseed()), dict(recipe='esm', L=500, D=1280, seed=nseed())],
         ['1-95', '96-210', '211-330,401-440', '331-400', '441-500', '1-500'], [3, 80, 3, 80])
add_case('dom_overlap3', [dict(recipe='gauss', L=60, D=256, seed=nseed())],
         ['1-60', '1-60', '10-50', '1-3', '58-60'], [3, 80])

# 5. degenerate numerics ---------------------------------------------------------
add_case('deg_short_1-2', [dict(recipe='gauss', L=10, D=128, seed=nseed())], ['1-2'], [3, 80])
add_case('deg_short_5-5', [dict(recipe='gauss', L=10, D=128, seed=nseed())], ['1-10', '5-5'], [3, 80])
add_case('deg_narrow', [dict(recipe='gauss', L=30, D=44, seed=nseed())], ['1-30'], [3, 80])
for L in (50, 500):
    add_case(f'deg_constcol_L{L}', [dict(recipe='gauss', L=L, D=640, seed=nseed(), patch=[['const_col', 7, 1.2345]]),
                                    dict(recipe='gauss', L=L, D=640, seed=nseed())],
             [f'1-{L}'], [3, 80, 3, 80])
add_case('deg_zeros', [dict(recipe='gauss', L=64, D=640, seed=nseed(), patch=[['zero_all']])], ['1-64'], [3, 80])
add_case('deg_nan', [dict(recipe='gauss', L=64, D=640, seed=nseed(), patch=[['nan_at', 3, 17]]),
                     dict(recipe='gauss', L=64, D=640, seed=nseed())], ['1-64', '10-64'], [3, 80, 3, 80])
add_case('deg_inf', [dict(recipe='gauss', L=64, D=640, seed=nseed(), patch=[['inf_at', 60, 5]])], ['1-64', '1-50'], [3, 80])
add_case('deg_empty_only', [dict(recipe='gauss', L=20, D=128, seed=nseed())], ['21-30', '0-5'], [3, 80])

# 6. tiny full matrices stored inline (no RNG dependence) ------------------------
add_case('inline_24x96', [dict(recipe='gauss', L=24, D=96, seed=nseed())], ['1-24', '3-20'], [3, 80],
         inline=True, intermediates=True)
add_case('inline_9x130', [dict(recipe='esm', L=9, D=130, seed=nseed())], ['1-9'], [5, 44], inline=True, intermediates=True)

# 7. contact map as an extra "layer" (generic over any 2-D matrix) ------
for L in (64, 120, 257):
    add_case(f'contact_layer_L{L}',
             [dict(recipe='esm', L=L, D=640, seed=nseed()), dict(recipe='contact', L=L, D=L, seed=nseed())],
             [f'1-{L}', f'1-{L // 2}', f'{L // 2 + 1}-{L}'], [3, 80, 5, 44])

# 8. BASELINE config 4 as stated: D = 2560 x RecCut-shaped domain lists (parts that tile the protein, a
#    discontinuous part, then the whole protein '1-L' -- src/fingerprint.py:103-107) x two layers; these are the
#    fused walks of the GPU path.  Part boundaries as RecCut emits them (multiples of nothing in particular).
C4 = {
    100: [['1-47', '48-100', '1-100'], ['1-30,71-100', '31-70', '1-100']],
    374: [['1-120', '121-250', '251-374', '1-374'], ['1-88,301-374', '89-200', '201-300', '1-374'],
          ['1-25', '26-51', '52-77', '78-374', '1-374']],
    500: [['1-95', '96-210', '211-330,401-440', '331-400', '441-500', '1-500'], ['1-250', '251-500', '1-500'],
          ['1-22', '23-44', '45-140,300-322', '141-299', '323-500', '1-500'], ['1-500']],
}
for L, lists in C4.items():
    for n_, doms in enumerate(lists):
        add_case(f'c4_D2560_L{L}_{n_}',
                 [dict(recipe='esm', L=L, D=2560, seed=nseed()), dict(recipe='esm', L=L, D=2560, seed=nseed())],
                 doms, [3, 80, 3, 80])
# the same shapes at the other widths (fused walks at D = 640 / 1280, incl. a width that is not a multiple of 256)
for D in (640, 1280, 1000):
    add_case(f'c4_D{D}_L374', [dict(recipe='esm', L=374, D=D, seed=nseed()), dict(recipe='esm', L=374, D=D, seed=nseed())],
             C4[374][1], [3, 80, 3, 80])
    add_case(f'c4_D{D}_L500', [dict(recipe='esm', L=500, D=D, seed=nseed()), dict(recipe='esm', L=500, D=D, seed=nseed())],
             C4[500][2], [3, 80, 3, 80])
# many short parts (more jobs than one stage-B group of the fused kernel holds)
edges = list(range(0, 500, 31)) + [500]
add_case('c4_D1280_L500_17parts', [dict(recipe='esm', L=500, D=1280, seed=nseed()), dict(recipe='gauss', L=500, D=1280, seed=nseed())],
         [f'{a + 1}-{b}' for a, b in zip(edges[:-1], edges[1:])] + ['1-500'], [3, 80, 3, 80])
# contact map as an extra layer at the config's largest L
add_case('contact_layer_L500',
         [dict(recipe='esm', L=500, D=2560, seed=nseed()), dict(recipe='contact', L=500, D=500, seed=nseed())],
         ['1-500', '1-250', '251-500'], [3, 80, 5, 44])

# get_doms table ------------------------------------------------------------------
gd = []
xg = np.arange(100 * 4, dtype=np.float32).reshape(100, 4)
fpg = Fingerprint(pid='g', seq='A' * 100, embed={0: xg})
for dom in ['1-100', '1-30,61-100', '90-120', '101-120', '1-30,101-120', '1-30,101-120,40-50', '0-10', '0-100',
            '0-200', '5-5', '1-2', '50-40,1-10', '20-25,20-25', '150-160,200-300,1-5,200-300', '100-100',
            '101-101,1-1', '150-160,200-300,200-300', '2-1']:
    mat, key = fpg.get_doms(xg, dom)
    gd.append({'dom': dom, 'L': 100, 'rows': [int(v) for v in (mat[:, 0] / 4).astype(int)], 'key': key,
               'dtype': str(mat.dtype)})

with open(os.path.join(HERE, 'quantize_golden.json'), 'w') as fh:
    json.dump({'generator': 'tests/golden/make_golden.py', 'reference': 'mgtools/DCTdomain @ 2024_10_08 src/fingerprint.py',
               'numpy': np.__version__, 'cases': cases}, fh, indent=1)
np.savez_compressed(os.path.join(HERE, 'quantize_golden.npz'), **arrays)
with open(os.path.join(HERE, 'getdoms_golden.json'), 'w') as fh:
    json.dump(gd, fh, indent=1)
print(f'{len(cases)} cases, {len(arrays)} arrays, '
      f'{sum(1 for c in cases if c["expect"] != "ok")} expected errors')
