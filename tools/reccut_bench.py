"""Local CPU benchmark of libreccut on SyntheticModel-like contact maps (pfam-like lengths)."""
import sys, time, os, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from dctdomain_amd.embedding import SyntheticModel
from oracle import contacts_oracle as co
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib_path = sys.argv[2] if len(sys.argv) > 2 else '/root/repo/dctdomain_amd/libreccut.so'
cache = '/tmp/rc/inputs_%d.npz' % n
if not os.path.exists(cache):
    rng = np.random.default_rng(7)
    lens = np.clip(rng.gamma(2.2, 170.0, size=n).astype(int), 81, 1330)
    aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
    model = SyntheticModel()
    ci, cj, cv, offs = [], [], [], [0]
    from dctdomain_amd.embedding import Embedding
    for k, L in enumerate(lens):
        seq = aa[rng.integers(0, 20, size=L)].tobytes().decode()
        # windows of 500 stitched like embed_seq does, on the CPU (numpy restatement is in oracle/stitch_oracle; here plain)
        if L <= 500:
            _, _, tok = model.esm_tokenizer([('x', seq)])
            ct = model.esm_encoder(tok)['contacts'][0].numpy()
        else:
            from oracle import stitch_oracle as so
            e = Embedding(pid='x', seq=seq)
            subs = e.split_seq(500, 200)
            cts = []
            for sseq in subs:
                _, _, tok = model.esm_tokenizer([('x', sseq)])
                cts.append(model.esm_encoder(tok)['contacts'][0])
            ct = cts[0]
            for i, c in enumerate(cts[1:], start=1):
                ct = so.combine_contacts(ct, c, 300, i)
            ct = ct.numpy()
        a, b, v = co.top_contacts(ct, 2.6)
        ci.append(a); cj.append(b); cv.append(v); offs.append(offs[-1] + len(a))
    np.savez(cache, lens=lens.astype(np.int32), offs=np.array(offs, np.int64), ci=np.concatenate(ci), cj=np.concatenate(cj), cv=np.concatenate(cv))
z = np.load(cache)
lens, offs, ci, cj, cv = z['lens'], z['offs'], z['ci'], z['cj'], z['cv']
lib = C.CDLL(lib_path)
lib.reccut_predict_batch.restype = C.c_int
stride = int(16 * lens.max())
buf = np.zeros((n, stride), np.uint8); nd = np.zeros(n, np.int32); rc = np.zeros(n, np.int32)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    lib.reccut_predict_batch(C.c_int64(n), C.c_void_p(lens.ctypes.data), C.c_void_p(offs.ctypes.data), C.c_void_p(ci.ctypes.data), C.c_void_p(cj.ctypes.data), C.c_void_p(cv.ctypes.data),
                             C.c_double(0.08), C.c_double(0.07), C.c_void_p(buf.ctypes.data), C.c_int64(stride), C.c_void_p(nd.ctypes.data), C.c_void_p(rc.ctypes.data), C.c_int32(1))
    best = min(best, time.perf_counter() - t0)
import hashlib
h = hashlib.sha256(b''.join(bytes(buf[p]).split(b'\0', 1)[0] + b'\n' for p in range(n))).hexdigest()
print(f'{lib_path}: {1e6 * best / n:.1f} us per protein on 1 thread (mean L {lens.mean():.0f}, mean domains {nd.mean():.2f}), rc!=0: {(rc != 0).sum()}, sha {h[:16]}')
