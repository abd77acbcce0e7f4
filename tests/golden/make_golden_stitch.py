#!/usr/bin/env python3
"""Golden vectors for the chunk stitcher, generated with the REFERENCE's own
``Embedding`` class.  src/embedding.py cannot be imported as a module here (its top-level
``import esm`` is a third-party package that is not installed), so this script -- build container
only, nothing of it is committed but the data -- takes the source text of that one class out of
/root/reference/src/embedding.py with ``ast``, executes it as it stands, and drives its
``split_seq`` / ``combine_contacts`` / ``embed_seq`` with a fake model object passed as the
``model`` argument (tests/golden/synthetic_esm.py; the class only calls model.esm_tokenizer and
model.esm_encoder).  No stand-in for the esm library is created."""

import ast
import hashlib
import json
import os
import sys
from dataclasses import dataclass, field

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from synthetic_esm import SyntheticESM, make_sequence  # noqa: E402

SRC = '/root/reference/src/embedding.py'
text = open(SRC).read()
node = next(n for n in ast.parse(text).body if isinstance(n, ast.ClassDef) and n.name == 'Embedding')
ns = {'torch': torch, 'np': np, 'dataclass': dataclass, 'field': field, 'Model': object}
exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, 'exec'), ns)
RefEmbedding = ns['Embedding']


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


cases, arrays = [], {}
model = SyntheticESM(dim=32)
for maxlen, lengths in ((500, [150, 500, 501, 700, 750, 800, 1000, 1035, 1400, 2000]), (400, [401, 650, 1000]),
                        (300, [301, 450, 650]), (1000, [1700])):
    for L in lengths:
        seq = make_sequence(L, 31 * L + maxlen)
        emb = RefEmbedding(pid=f'p{L}', seq=seq)
        windows = emb.split_seq(maxlen, 200) if L > maxlen else [seq]
        emb.embed_seq(model, 'cpu', [15, 21], maxlen)                       # src/embedding.py:153-192
        cid = f'stitch_m{maxlen}_L{L}'
        case = {'id': cid, 'L': L, 'maxlen': maxlen, 'seq_seed': 31 * L + maxlen,
                'windows': [len(w) for w in windows],
                'embed_sha': {str(k): sha(v) for k, v in emb.embed.items()},
                'embed_shape': list(emb.embed[15].shape), 'contacts_sha': sha(emb.contacts),
                'contacts_shape': list(emb.contacts.shape),
                'contacts_rowsum_sha': sha(emb.contacts.astype(np.float64).sum(axis=1))}
        if L <= 800:
            arrays[f'{cid}/e15'] = emb.embed[15]
        cases.append(case)

# combine_contacts on its own (hand-sized)
e = RefEmbedding(pid='c', seq='A' * 10)
rng = np.random.default_rng(5)
m1 = torch.from_numpy(rng.random((7, 7)).astype(np.float32))
m2 = torch.from_numpy(rng.random((6, 6)).astype(np.float32))
arrays['combine/m1'], arrays['combine/m2'] = m1.numpy(), m2.numpy()
arrays['combine/out_inc3_t1'] = e.combine_contacts(m1, m2, 3, 1).numpy()
arrays['combine/out_inc2_t2'] = e.combine_contacts(m1, m2, 2, 2).numpy()

with open(os.path.join(HERE, 'stitch_golden.json'), 'w') as fh:
    json.dump({'generator': 'tests/golden/make_golden_stitch.py', 'cases': cases}, fh, indent=1)
np.savez_compressed(os.path.join(HERE, 'stitch_golden.npz'), **arrays)
print(len(cases), 'cases')
for c in cases:
    print(c['id'], c['windows'], c['embed_shape'])
