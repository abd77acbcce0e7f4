"""What the GPU path does on the three input classes where bit-exactness is not claimed (tests/golden/fence_golden.*)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import numpy as np, torch
import dctdomain_amd as dd
from recipes import make_input
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
doc = json.load(open(os.path.join(G, 'fence_golden.json')))
arr = np.load(os.path.join(G, 'fence_golden.npz'))
ctx = dd.get_context(0)
for key in ('d_equals_m', 'repeated_rows'):
    for c in doc[key]:
        x = make_input(c['recipe'], c['L'], c['D'], c['seed'])
        fp = dd.Fingerprint(pid='f', seq='A' * c['L'], embed={0: x}, domains=[c['domain']])
        fp.quantize(c['qdim'])
        got = fp.quants[c['key']]
        exp = arr[c['id'] + '/out'].astype(np.int64)
        d = got - exp
        print(key, c['id'], 'n', len(exp), 'mismatch', int((d != 0).sum()), 'max|d|', int(np.abs(d).max()), 'got zeros', int((got == 0).sum()), 'exp zeros', int((exp == 0).sum()),
              'where', [(int(i), int(exp[i]), int(got[i])) for i in np.nonzero(d)[0][:8]])
cc = doc['const']
ctx.set_option('degenerate_channels', 0)
Ls = list(range(3, 2001))
xs = []
for L in Ls:
    x = make_input(cc['recipe'], L, cc['D'], cc['seed0'] + L)
    x[:, cc['col']] = np.float32(cc['value'])
    xs.append(torch.from_numpy(x).cuda())
table = dd.PieceTable.whole_sequences(Ls)
out = dd.quantize_batch([dd.LayerBatch(xs, 3, 80)], table).cpu().numpy()
nz = [L for L, row in zip(Ls, out) if row.any()]
print('constant channel: lengths with a non-zero GPU block:', len(nz), nz[:10], ' counter', ctx.get_option('degenerate_channels'), 'of', len(Ls))
