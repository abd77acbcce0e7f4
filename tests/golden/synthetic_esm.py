"""Deterministic fake language model for the stitcher tests: every window's output is a pure
function of the window's residues (seeded by their CRC), so the reference code and the build see
identical per-window tensors.  Same duck-typed interface as the reference's Model
(esm_tokenizer / esm_encoder, src/embedding.py:108-112)."""

import zlib

import numpy as np
import torch


class SyntheticESM:
    def __init__(self, dim=32, layers=(15, 21), device='cpu'):
        self.dim, self.layers, self.device = dim, tuple(layers), torch.device(device)
        self.padding_idx = 1

    def esm_tokenizer(self, pairs):
        width = max(len(s) for _, s in pairs) + 2
        tok = torch.full((len(pairs), width), self.padding_idx, dtype=torch.long)
        for b, (_, s) in enumerate(pairs):
            tok[b, 0] = 0
            tok[b, 1:1 + len(s)] = torch.tensor([ord(c) for c in s], dtype=torch.long)
            tok[b, 1 + len(s)] = 2
        return None, None, tok

    def esm_encoder(self, tokens, repr_layers=None, return_contacts=True):
        tokens = tokens.cpu()
        b, t = tokens.shape
        reps = {l: torch.zeros((b, t, self.dim), dtype=torch.float32) for l in self.layers}
        cts = torch.zeros((b, t - 2, t - 2), dtype=torch.float32)
        for k in range(b):
            n = int((tokens[k] != self.padding_idx).sum()) - 2
            seed = zlib.crc32(bytes(int(v) & 0xff for v in tokens[k, 1:1 + n]))
            rng = np.random.default_rng(seed)
            for l in self.layers:
                x = rng.standard_normal((n + 2, self.dim)) * np.exp(rng.standard_normal(self.dim)) + 5 * rng.standard_normal(self.dim)
                reps[l][k, :n + 2] = torch.from_numpy(x.astype(np.float32))
            c = rng.random((n, n)) * np.exp(-np.abs(np.arange(n)[:, None] - np.arange(n)[None, :]) / 20.0)
            cts[k, :n, :n] = torch.from_numpy((0.5 * (c + c.T)).astype(np.float32))
        return {'representations': {l: r.to(self.device) for l, r in reps.items()}, 'contacts': cts.to(self.device)}


def make_sequence(length, seed):
    rng = np.random.default_rng(seed)
    return ''.join('ACDEFGHIKLMNPQRSTVWY'[int(v)] for v in rng.integers(0, 20, size=length))
