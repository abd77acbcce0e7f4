"""Stage boundaries of one make_db flush on the host clock (make_db.MARKS), with the GPU's own time for the same flush from
hipEvents around it: where the per-protein time of `fingerprint_batch` goes (cProfile in tools/flush_profile.py inflates the
Python share; this does not)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dctdomain_amd import make_db, reccut
from dctdomain_amd.embedding import Batch, SyntheticModel
from dctdomain_amd.fingerprint import Fingerprint

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
smooth = len(sys.argv) > 2 and sys.argv[2] == 'tiefree'
dev = torch.device('cuda', 0)
rng = np.random.default_rng(7)
lens = np.clip(rng.gamma(2.2, 170.0, size=n).astype(int), 81, 1330)
aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
seqs = [(f'sp{i:07d}', aa[rng.integers(0, 20, size=L)].tobytes().decode()) for i, L in enumerate(lens)]
model = SyntheticModel()
model.to_device(dev)
queue = []
for pid, seq in seqs:
    bt = Batch([(pid, seq)], model, dev)
    bt.embed_batch(make_db.LAYERS, 500)
    for emb in bt.embeds:
        c = emb.contacts
        if smooth:      # a real model's maps have no plateaus: add a tie-breaking ramp far below the synthetic model's level spacing
            c = c + torch.rand_like(c) * 1e-4
        queue.append(Fingerprint(pid=emb.pid, seq=emb.seq, embed=emb.embed, contacts=c))
torch.cuda.synchronize()
fresh = lambda: [Fingerprint(pid=f.pid, seq=f.seq, embed=f.embed, contacts=f.contacts) for f in queue]
make_db._records(make_db.fingerprint_batch(fresh()[:256], threads=16))
make_db._records(make_db.fingerprint_batch(fresh(), threads=16))


def timed(label, fn, reps=5):
    best = None
    for rep in range(reps):
        q = fresh()
        torch.cuda.synchronize()
        make_db.MARKS = []
        t0 = time.perf_counter()
        out = fn(q)
        t2 = time.perf_counter()
        marks = make_db.MARKS + [('end', t2)]
        make_db.MARKS = None
        if best is None or t2 - t0 < best[0]:
            best = (t2 - t0, marks, t0)
    dt, marks, t0 = best
    print(f'{label}: best of {reps} {1e3 * dt:.2f} ms = {1e6 * dt / n:.2f} us per protein ({"tie-free" if smooth else "synthetic"} maps; '
          f'path {make_db.LAST_PATH[0]}, host redo {len(reccut.LAST.host_redo)})')
    if getattr(reccut.LAST, 'gpu_ms', None) and make_db.LAST_PATH[0] == 'flush':
        g = reccut.LAST.gpu_ms
        print(f'  on the GPU (events on the side stream): contact top-k {g[0]:.2f} ms, cutter {g[1]:.2f} ms, results to the host {g[2]:.2f} ms')
    prev = t0
    for name, t in marks:
        print(f'  {1e3 * (t - t0):8.2f} ms  (+{1e3 * (t - prev):6.2f})  {name}')
        prev = t
    return out


timed(f'fingerprint_batch + _records of {n} (objects filled as queue_cpu does)', lambda q: make_db._records(make_db.fingerprint_batch(q, threads=16)))
timed(f'flush_records of {n} (what make_db hands its writer)', lambda q: make_db.flush_records(q, threads=16))
timed(f'general path of {n} (round 5, first half)', lambda q: make_db._records(make_db._fingerprint_batch_generic(q, threads=16)), reps=3)


def two_halves(q):
    fl = make_db._Flush(q, threads=16)
    assert fl.start()
    make_db._mark('first half done (process_sequences embeds the next proteins here)')
    fl.cut.done.synchronize()
    make_db._mark('-- the cutter, waited for here only to take it out of the sum below')
    t_mid = time.perf_counter()
    recs = fl.finish(objects=False)
    two_halves.second = time.perf_counter() - t_mid
    make_db.LAST_PATH[0] = 'flush'
    return recs


timed(f'the two halves of a flush of {n} apart (as process_sequences runs them)', two_halves)
# the GPU's share of the same flush: the chains alone, waited for
maps = [reccut._contact_tensor(f.contacts, len(f.seq)) for f in queue]
for rep in range(2):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    sel = reccut._select_on_device(maps, make_db.THRESHOLD)
    e[1].record()
    doms = reccut._cut_on_device(*sel, reccut.CUT1_DEFAULT, reccut.CUT2_DEFAULT, 16, lambda: e[2].record())
    torch.cuda.synchronize()
print(f'GPU: contact top-k {e[0].elapsed_time(e[1]):.2f} ms (incl. its enqueue), cutter {e[1].elapsed_time(e[2]):.2f} ms')
