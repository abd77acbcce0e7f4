"""Seed-defined synthetic inputs shared by the golden-vector generator and the tests.

Pure numpy (``np.random.default_rng``), so a case is reproduced from
``(recipe, L, D, seed)`` alone; every golden case also stores the sha256 of the
regenerated float32 bytes so that a drift of the RNG stream is caught loudly
instead of showing up as a parity failure.
"""

from __future__ import annotations

import hashlib

import numpy as np


def make_input(recipe: str, n_rows: int, n_cols: int, seed: int) -> np.ndarray:
    """(n_rows, n_cols) float32, C-contiguous."""
    rng = np.random.default_rng(seed)
    if recipe == 'gauss':
        sigma = 0.1 + 2.9 * rng.random()
        x = rng.standard_normal((n_rows, n_cols)) * sigma
    elif recipe == 'esm':
        # ESM-like: per-channel scale and offset, a few huge-offset channels.
        # This is the family that breaks an all-fp32 restatement.
        ch_scale = np.exp(rng.standard_normal(n_cols))
        ch_off = 5.0 * rng.standard_normal(n_cols)
        n_out = max(1, n_cols // 100)
        idx = rng.choice(n_cols, size=n_out, replace=False)
        ch_off[idx] += 200.0 * rng.choice([-1.0, 1.0], size=n_out)
        x = rng.standard_normal((n_rows, n_cols)) * ch_scale + ch_off
    elif recipe == 'smooth':
        # low-frequency structure along the sequence plus noise
        t = np.linspace(0.0, 1.0, n_rows)[:, None]
        ph = rng.random((1, n_cols)) * 2 * np.pi
        fr = rng.integers(1, 6, size=(1, n_cols))
        x = np.sin(2 * np.pi * fr * t + ph) * (0.5 + rng.random((1, n_cols)))
        x = x + 0.05 * rng.standard_normal((n_rows, n_cols))
    elif recipe == 'contact':
        # square symmetric "contact map" (SURVEY App. B recipe); n_cols must equal n_rows
        assert n_rows == n_cols
        i = np.arange(n_rows)[:, None]
        j = np.arange(n_rows)[None, :]
        half = (i < n_rows // 2) == (j < n_rows // 2)
        p = 0.9 * np.exp(-np.abs(i - j) / 12.0) + 0.25 * rng.random((n_rows, n_rows)) * half
        x = 0.5 * (p + p.T)
    else:
        raise KeyError(recipe)
    return np.ascontiguousarray(x, dtype=np.float32)


def sha256_of(arr: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
