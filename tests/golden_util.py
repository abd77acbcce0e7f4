"""Loads the committed golden vectors (tests/golden/) and rebuilds their inputs."""

from __future__ import annotations

import json
import os

import numpy as np

from recipes import make_input, sha256_of

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

_manifest = None
_arrays = None


def manifest():
    global _manifest
    if _manifest is None:
        with open(os.path.join(GOLD, 'quantize_golden.json')) as fh:
            _manifest = json.load(fh)
    return _manifest


def arrays():
    global _arrays
    if _arrays is None:
        _arrays = np.load(os.path.join(GOLD, 'quantize_golden.npz'))
    return _arrays


def cases(prefix=None, expect=None):
    out = []
    for c in manifest()['cases']:
        if prefix is not None and not c['id'].startswith(prefix):
            continue
        if expect is not None and c['expect'] != expect:
            continue
        out.append(c)
    return out


def case_ids(cs):
    return [c['id'] for c in cs]


def build_layers(case):
    """Regenerates the float32 inputs of a case and checks their sha256."""
    layers = []
    for sp in case['layers']:
        if 'inline' in sp:
            x = np.array(arrays()[sp['inline']])
        else:
            x = make_input(sp['recipe'], sp['L'], sp['D'], sp['seed'])
            for op in sp.get('patch', []):
                if op[0] == 'const_col':
                    x[:, op[1]] = np.float32(op[2])
                elif op[0] == 'zero_all':
                    x[:] = 0
                elif op[0] == 'nan_at':
                    x[op[1], op[2]] = np.nan
                elif op[0] == 'inf_at':
                    x[op[1], op[2]] = np.inf
        assert sha256_of(x) == sp['sha256'], f"input drift for {case['id']} (numpy RNG stream changed?)"
        layers.append(x)
    return layers


def expected(case):
    """{key: int8 array} in key order."""
    return {k: np.array(arrays()[f"{case['id']}/out/{j}"]) for j, k in enumerate(case['keys'])}
