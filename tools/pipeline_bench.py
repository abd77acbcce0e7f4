"""End-to-end timing of the make_db pipeline on one GPU with the synthetic language model:
where does the time go once the fingerprint kernels run at HBM speed?  (SURVEY 8f rows.)

    python tools/pipeline_bench.py [n_proteins] [threads]
"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dctdomain_amd as dd
from dctdomain_amd import make_db, reccut
from dctdomain_amd.embedding import Batch, SyntheticModel
from dctdomain_amd.batch import LayerBatch, PieceTable, quantize_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(5)
# pfam_max50-like length distribution: 81..1330, mean ~374
lens = np.clip(rng.gamma(2.2, 170.0, size=n).astype(int), 81, 1330)
seqs = [(f'p{i}', ''.join('ACDEFGHIKLMNPQRSTVWY'[int(v)] for v in rng.integers(0, 20, size=L))) for i, L in enumerate(lens)]
dev = torch.device('cuda', 0)
model = SyntheticModel()
model.to_device(dev)

def sync():
    torch.cuda.synchronize()

t = {}
sync(); t0 = time.perf_counter()
fps = []
for pid, seq in seqs:
    b = Batch([(pid, seq)], model, dev)
    b.embed_batch([15, 21], 500)
    e = b.embeds[0]
    fps.append(dd.Fingerprint(pid=e.pid, seq=e.seq, embed=e.embed, contacts=e.contacts))
sync(); t['embed (synthetic model + stitch)'] = time.perf_counter() - t0

lens_l = [len(fp.seq) for fp in fps]
maps = [reccut._contact_tensor(fp.contacts, L) for fp, L in zip(fps, lens_l)]
sync(); t0 = time.perf_counter()
offs, ci, cj, cv = reccut.top_contacts_batch(maps, 2.6, sort=False)      # what make_db.fingerprint_batch does
sync(); t['contact top-k (GPU) + D2H'] = time.perf_counter() - t0
t0 = time.perf_counter()
reccut.top_contacts_batch(maps, 2.6, sort=True)
sync(); t['(same with the .ce ordering: host sort, only for writece)'] = time.perf_counter() - t0
t0 = time.perf_counter()
doms = reccut.domains_from_contacts(lens_l, offs, ci, cj, cv, threads=threads)
t[f'RecCut in-process ({threads} threads)'] = time.perf_counter() - t0
for fp, d, L in zip(fps, doms, lens_l):
    fp.domains = list(d) + ([f'1-{L}'] if len(d) > 1 else [])
t0 = time.perf_counter()
table = PieceTable(lens_l, [fp.domains for fp in fps])
t['piece table (host, python)'] = time.perf_counter() - t0
layers = [LayerBatch([fp.embed[k] for fp in fps], 3, 80) for k in (15, 21)]
quantize_batch(layers, table); sync()
t0 = time.perf_counter()
out = quantize_batch(layers, table)
sync(); t['quantize (GPU, dctfp_quantize)'] = time.perf_counter() - t0
t0 = time.perf_counter()
host = out.cpu().numpy()
t['D2H of fingerprints'] = time.perf_counter() - t0
import tempfile
from dctdomain_amd.database import Database
class R:  # writer records
    pass
recs = []
row = 0
per = {}
for r_, s in enumerate(table.owner):
    per.setdefault(s, []).append(r_)
for s, fp in enumerate(fps):
    r = R(); r.pid = fp.pid; r.domains = [table.keys[q] for q in per[s]]
    r.quants = {table.keys[q]: host[q] for q in per[s]}
    recs.append(r)
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, 'x.fasta')
    with open(fa, 'w') as fh:
        for pid, seq in seqs:
            fh.write(f'>{pid}\n{seq}\n')
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        db = Database(os.path.join(td, 'x'), fa)
        t0 = time.perf_counter()
        db.add_fprints(recs)
        db.rename_vid()
        t['SQLite insert + renumber'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        db.save_fprints(os.path.join(td, 'x-dct.npz')); db.save_doms(os.path.join(td, 'x.dom'))
        t['npz + dom files'] = time.perf_counter() - t0
        db.close()
n_fp = table.n_domains
print(json.dumps({'proteins': n, 'residues': int(sum(lens_l)), 'fingerprints': n_fp,
                  'multi_domain_fraction': float(np.mean([len(d) > 1 for d in doms])),
                  'seconds': {k: round(v, 4) for k, v in t.items()},
                  'per_protein_us': {k: round(1e6 * v / n, 1) for k, v in t.items()}}, indent=1))
