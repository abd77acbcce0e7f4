"""Where the host time of one make_db flush goes: cProfile over `fingerprint_batch` + `_records` for N synthetic proteins
with a pfam-like length mix (the fingerprint stage of profiles/r03/db_build_1M.txt: 117 us per protein, kernels < 1 us)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import make_db
from dctdomain_amd.embedding import Batch, SyntheticModel
from dctdomain_amd.fingerprint import Fingerprint

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device('cuda', 0)
rng = np.random.default_rng(7)
lens = np.clip(rng.gamma(2.2, 170.0, size=n).astype(int), 81, 1330)
aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
seqs = [(f'sp{i:07d}', aa[rng.integers(0, 20, size=L)].tobytes().decode()) for i, L in enumerate(lens)]
model = SyntheticModel()
model.to_device(dev)
def make_queue():
    q = []
    for pid, seq in seqs:
        bt = Batch([(pid, seq)], model, dev)
        bt.embed_batch(make_db.LAYERS, 500)
        for emb in bt.embeds:
            q.append(Fingerprint(pid=emb.pid, seq=emb.seq, embed=emb.embed, contacts=emb.contacts))
    return q
t0 = time.perf_counter()
queue = make_queue()
torch.cuda.synchronize()
print(f'embed stage: {1e6 * (time.perf_counter() - t0) / n:.1f} us per protein')
make_db._records(make_db.fingerprint_batch(queue[:256], threads=16))        # warm-up (tables, cosine cache)
queue = make_queue()
torch.cuda.synchronize()
for rep in range(2):
    q = [Fingerprint(pid=f.pid, seq=f.seq, embed=f.embed, contacts=f.contacts) for f in queue]
    t0 = time.perf_counter()
    recs = make_db._records(make_db.fingerprint_batch(q, threads=16))
    dt = time.perf_counter() - t0
    print(f'flush of {n}: {1e3 * dt:.1f} ms = {1e6 * dt / n:.1f} us per protein, {sum(len(r[1]) for r in recs)} fingerprints')
q = [Fingerprint(pid=f.pid, seq=f.seq, embed=f.embed, contacts=f.contacts) for f in queue]
pr = cProfile.Profile()
pr.enable()
recs = make_db._records(make_db.fingerprint_batch(q, threads=16))
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue())
