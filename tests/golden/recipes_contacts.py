"""Seed-defined synthetic contact maps (float32, symmetric) for the domain-prediction tests."""

import numpy as np


def make_contacts(recipe: str, L: int, seed: int, nb: int = 3) -> np.ndarray:
    rng = np.random.default_rng(seed)
    i = np.arange(L)[:, None]
    j = np.arange(L)[None, :]
    near = 0.9 * np.exp(-np.abs(i - j) / 12.0)
    if recipe in ('blocks', 'interleaved'):
        nb = max(1, nb)
        lab = (np.arange(L) * nb // max(L, 1))
        if recipe == 'interleaved' and nb >= 3:
            lab = np.where(lab == nb - 1, 0, lab)          # last block folds onto the first: discontinuous domain
        same = lab[i] == lab[j]
        if recipe == 'interleaved':
            p = 0.6 * near + 0.9 * (rng.random((L, L)) < 0.12) * same * (0.6 + 0.4 * rng.random((L, L)))
        else:
            p = near + 0.3 * rng.random((L, L)) * same
    elif recipe == 'ties':
        p = np.round((near + 0.3 * rng.random((L, L))) * 8) / 8.0
    elif recipe == 'sparse':
        p = near * (np.abs(i - j) < 30) * (rng.random((L, L)) < 0.2)
    elif recipe == 'flat':
        p = np.full((L, L), 0.25)
    elif recipe == 'negzero':
        p = np.where(rng.random((L, L)) < 0.5, -0.0, 0.0) + (rng.random((L, L)) < 0.05) * 0.5
    else:
        raise KeyError(recipe)
    p = 0.5 * (p + p.T)
    return np.ascontiguousarray(p, dtype=np.float32)
