"""make_db drop-in on the GPU with the synthetic language model: the batched path
(fingerprint_batch) must equal queue_cpu protein by protein, which in turn is checked against the
CPU oracles; files must have the reference layout; one GPU and two worker processes must write
identical files."""

import os

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
FIX = os.path.join(gu.GOLD, 'ref_fixtures')


def test_fingerprint_batch_equals_queue_cpu_and_oracle():
    import dctdomain_amd as dd
    from dctdomain_amd import make_db
    from dctdomain_amd.embedding import Batch, SyntheticModel
    from oracle import contacts_oracle as co
    from oracle import dct_oracle as orc
    seqs = []
    with open(os.path.join(FIX, 'example.fasta')) as fh:
        for line in fh:
            if line.startswith('>'):
                seqs.append([line.split()[0][1:], ''])
            else:
                seqs[-1][1] += line.strip()
    dev = torch.device('cuda', 0)
    model = SyntheticModel()
    model.to_device(dev)

    def embed_all():
        out = []
        for pid, seq in seqs:
            b = Batch([(pid, seq)], model, dev)
            b.embed_batch([15, 21], 500)
            e = b.embeds[0]
            out.append(dd.Fingerprint(pid=e.pid, seq=e.seq, embed=e.embed, contacts=e.contacts))
        return out

    one_by_one = [make_db.queue_cpu(fp) for fp in embed_all()]
    batched = make_db.fingerprint_batch(embed_all(), threads=4)
    n_multi = 0
    for a, b in zip(one_by_one, batched):
        assert a.domains == b.domains
        for k in a.domains:
            np.testing.assert_array_equal(a.quants[k], b.quants[k])
        n_multi += len(a.domains) > 1
        # CPU chain: oracle top-k -> reference RecCut binary -> oracle quantize
        ct = a.contacts.cpu().numpy()
        ci, cj, cv = co.top_contacts(ct, 2.6)
        if os.path.exists(co.REF_BIN):
            rc, out = co.run_ref_binary(co.ce_text(a.pid, a.seq, ci, cj, cv), a.pid)
            assert rc == 0 and co.parse_reccut(out, len(a.seq)) == a.domains
        q = orc.quantize([a.embed[15].cpu().numpy(), a.embed[21].cpu().numpy()], a.domains, [3, 80, 3, 80])
        for k in q:
            np.testing.assert_array_equal(a.quants[k], q[k])
    assert n_multi >= 3


def _run(tmp_path, name, extra):
    from dctdomain_amd import make_db
    dbfile = str(tmp_path / name)
    make_db.main(['--fafile', os.path.join(FIX, 'example.fasta'), '--dbfile', dbfile, '--model', 'synthetic',
                  '--cpu', '4', '--out', str(tmp_path / f'{name}.log')] + extra)
    return dbfile


def test_cli_outputs_and_two_workers(tmp_path):
    a = _run(tmp_path, 'one', [])
    za = np.load(a + '-dct.npz')
    assert za['sid'].shape == (8,) and za['idx'][0] == 0 and za['idx'][-1] == za['dct'].shape[0]
    assert za['dct'].dtype == np.int8 and za['dct'].shape[1] == 480 and za['idx'].dtype == np.int64
    rows = za['dct'].reshape(-1, 6, 80)
    assert ((rows == 127).sum(axis=2) == 1).all() and ((rows == 0).sum(axis=2) >= 1).all()
    dom_lines = open(a + '.dom').read().strip().split('\n')
    assert len(dom_lines) == 8 and all(len(l.split()) == 3 for l in dom_lines)
    assert os.path.exists(a + '.index') and os.path.exists(a + '.db')
    assert 'Fingerprinted' in open(str(tmp_path / 'one.log')).read()
    # resume: a second run over the finished database fingerprints nothing new
    from dctdomain_amd.database import Database
    db = Database(a + '.db')
    assert db.pending() == []
    db.close()
    # two worker processes (both on the one visible GPU here) write the same files
    b = _run(tmp_path, 'two', ['--gpu', '2', '--flush', '3'])
    zb = np.load(b + '-dct.npz')
    for k in za.files:
        np.testing.assert_array_equal(za[k], zb[k])
    assert open(a + '.dom').read() == open(b + '.dom').read()


def test_config5_scale_build_two_workers_against_one_and_oracle(tmp_path):
    """BASELINE config 5, scaled to one GPU box: 5 200 proteins with pfam-like lengths (81..1330) through the
    make_db drop-in -- two worker processes against one (identical files: the single writer commits every flush in
    sequence order), then a sample of proteins recomputed by the CPU chain (oracle top-k -> the reference's RecCut
    binary when it is there -> oracle quantize) against the rows of the written -dct.npz."""
    from dctdomain_amd import make_db
    from dctdomain_amd.embedding import Batch, SyntheticModel
    from oracle import contacts_oracle as co
    from oracle import dct_oracle as orc
    rng = np.random.default_rng(55)
    n_prot = 5200
    lengths = np.clip(rng.gamma(2.2, 170.0, size=n_prot).astype(np.int64), 81, 1330)
    fa = str(tmp_path / 'c5.fasta')
    seqs = {}
    with open(fa, 'w') as fh:
        for i, L in enumerate(lengths):
            s = ''.join('ACDEFGHIKLMNPQRSTVWY'[int(v)] for v in rng.integers(0, 20, size=int(L)))
            seqs[f'P{i:05d}'] = s
            fh.write(f'>P{i:05d} synthetic\n{s}\n')

    def build(name, extra):
        dbfile = str(tmp_path / name)
        make_db.main(['--fafile', fa, '--dbfile', dbfile, '--model', 'synthetic', '--cpu', '8', '--noindex',
                      '--flush', '512', '--out', str(tmp_path / f'{name}.log')] + extra)
        return dbfile

    two = build('two', ['--gpu', '2'])
    one = build('one', [])
    za, zb = np.load(one + '-dct.npz'), np.load(two + '-dct.npz')
    for k in za.files:
        np.testing.assert_array_equal(za[k], zb[k])
    assert open(one + '.dom').read() == open(two + '.dom').read()
    assert za['sid'].shape == (n_prot,) and za['dct'].shape[1] == 480 and za['dct'].shape[0] == za['idx'][-1]
    assert za['dct'].shape[0] > 2 * n_prot                      # mostly multi-domain: parts + whole protein
    # table order = ascending length (stable), whatever the number of workers
    assert list(za['sid']) == sorted(seqs, key=lambda p: len(seqs[p]))
    rows = za['dct'].reshape(-1, 6, 80)
    assert ((rows == 127).sum(axis=2) == 1).all() and ((rows == 0).sum(axis=2) >= 1).all()
    # the database is complete and resumable: nothing pending, fpcount = rows per protein
    from dctdomain_amd.database import Database
    db = Database(two + '.db')
    assert db.pending() == []
    db.close()

    # sample: CPU chain on the same embeddings
    dev = torch.device('cuda', 0)
    model = SyntheticModel()
    model.to_device(dev)
    sid = list(za['sid'])
    n_multi = 0
    for pid in [sid[i] for i in (0, 7, 601, 1999, 2600, 3333, 4100, 4800, 5100, 5199)]:
        b = Batch([(pid, seqs[pid])], model, dev)
        b.embed_batch([15, 21], 500)
        e = b.embeds[0]
        ct = e.contacts.cpu().numpy()
        ci, cj, cv = co.top_contacts(ct, 2.6)
        k = sid.index(pid)
        doms = [str(d) for d in za['dom'][za['idx'][k]:za['idx'][k + 1]]]
        if os.path.exists(co.REF_BIN):
            rc, out = co.run_ref_binary(co.ce_text(pid, seqs[pid], ci, cj, cv), pid)
            assert rc == 0 and co.parse_reccut(out, len(seqs[pid])) == doms
        q = orc.quantize([e.embed[15].cpu().numpy(), e.embed[21].cpu().numpy()], doms, [3, 80, 3, 80])
        assert list(q) == doms
        for r, key in enumerate(doms):
            np.testing.assert_array_equal(za['dct'][za['idx'][k] + r].astype(np.int64), q[key], err_msg=f'{pid} {key}')
        n_multi += len(doms) > 1
    assert n_multi >= 5


def test_constant_channel_is_reported_by_the_drop_in(tmp_path, monkeypatch):
    """the one documented deviation from the reference -- an exactly constant channel gives 0/0 -> the
    all-zero (layer, domain) block at EVERY domain length, where scipy's pocketfft scales round-off noise at 225 of the
    lengths 3..2000 (tests/golden/fence_golden.json) -- must not be silent for a user of the drop-in: make_db logs a
    warning naming the proteins (into --out), Fingerprint.quantize does the same for its one protein."""
    import json
    import logging
    import dctdomain_amd as dd
    from dctdomain_amd import make_db
    from dctdomain_amd.embedding import SyntheticModel
    noisy_lengths = set(json.load(open(os.path.join(gu.GOLD, 'fence_golden.json')))['const']['noisy_L'])
    assert len(noisy_lengths) == 225

    class OneDeadChannel(SyntheticModel):
        """layer 15, channel 7 carries the same value at every residue (a dead unit of the language model)."""
        def esm_encoder(self, tokens, repr_layers=None, return_contacts=True):
            out = super().esm_encoder(tokens, repr_layers, return_contacts)
            out['representations'][15][..., 7] = 1.5
            return out

    def load_model(name, device):
        m = OneDeadChannel()
        m.to_device(device)
        return m

    monkeypatch.setattr(make_db, 'load_model', load_model)
    L = min(n for n in noisy_lengths if n >= 90)               # a length at which the reference would NOT give the 0 block
    rng = np.random.default_rng(3)
    fa = str(tmp_path / 'dead.fasta')
    with open(fa, 'w') as fh:
        for pid, n in (('NOISY', L), ('OTHER', 130)):
            fh.write(f'>{pid} synthetic\n' + ''.join('ACDEFGHIKLMNPQRSTVWY'[int(v)] for v in rng.integers(0, 20, size=n)) + '\n')
    dbfile, log = str(tmp_path / 'dead'), str(tmp_path / 'dead.log')
    make_db.main(['--fafile', fa, '--dbfile', dbfile, '--model', 'synthetic', '--cpu', '2', '--noindex', '--out', log])
    text = open(log).read()
    assert 'constant channel in' in text and 'NOISY' in text and 'OTHER' in text and '225 of the lengths' in text
    z = np.load(dbfile + '-dct.npz')
    assert (z['dct'][:, :240] == 0).all()                       # layer 15: every (domain) block is the 0 block
    rows = z['dct'][:, 240:].reshape(-1, 3, 80)                 # layer 21 is untouched
    assert ((rows == 127).sum(axis=2) == 1).all()
    # the class itself, one protein per call (own handler: make_db's --out replaced the root logger's handlers)
    seen = []
    grab = logging.Handler()
    grab.emit = lambda record: seen.append(record.getMessage())
    logging.getLogger().addHandler(grab)
    try:
        x = np.random.default_rng(4).standard_normal((L, 640)).astype(np.float32)
        x[:, 11] = -0.25
        fp = dd.Fingerprint(pid='ONE', seq='A' * L, embed={0: x}, domains=[f'1-{L}'])
        fp.quantize([3, 80])
        assert any('constant channel in ONE' in m for m in seen)
        assert (fp.quants[f'1-{L}'] == 0).all()
        del seen[:]
        y = x.copy()
        y[:, 11] = x[:, 12]
        fp = dd.Fingerprint(pid='FINE', seq='A' * L, embed={0: y}, domains=[f'1-{L}'])
        fp.quantize([3, 80])
        assert not any('constant channel' in m for m in seen)
    finally:
        logging.getLogger().removeHandler(grab)


def _embedded(lens, seed=5, dim=640):
    import dctdomain_amd as dd
    from dctdomain_amd.embedding import Batch, SyntheticModel
    rng = np.random.default_rng(seed)
    aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
    dev = torch.device('cuda', 0)
    model = SyntheticModel(dim=dim)
    model.to_device(dev)
    out = []
    for i, L in enumerate(lens):
        b = Batch([(f'p{i:04d}', aa[rng.integers(0, 20, size=int(L))].tobytes().decode())], model, dev)
        b.embed_batch([15, 21], 500)
        e = b.embeds[0]
        out.append((e.pid, e.seq, e.embed, e.contacts))
    fresh = lambda: [dd.Fingerprint(pid=p, seq=s, embed=e, contacts=c) for p, s, e, c in out]
    return fresh


def test_two_phase_flush_equals_the_general_path_and_the_records():
    """The flush as make_db runs it (geometry in one pass, strings + pieces from the cutter's integers, objects laid out while the
    kernels run) against the general path it replaced, object for object; the writer's records against both; proteins the GPU
    cutter hands back (L > 2 048), single-domain and 30-residue proteins inside."""
    from dctdomain_amd import make_db, reccut
    rng = np.random.default_rng(3)
    lens = np.clip(rng.gamma(2.2, 150.0, size=160).astype(int), 30, 1500).tolist() + [30, 31, 45, 2100, 640, 2300, 97]
    fresh = _embedded(lens)
    a = make_db.fingerprint_batch(fresh(), threads=4)
    assert make_db.LAST_PATH[0] == 'flush' and sorted(reccut.LAST.host_redo) == [163, 165]
    b = make_db._fingerprint_batch_generic(fresh(), threads=4)
    recs = make_db.flush_records(fresh(), threads=4)
    assert make_db.LAST_PATH[0] == 'flush'
    recs_b = make_db._records(b)
    n_multi = 0
    for fa, fb, ra, rb in zip(a, b, recs, recs_b):
        assert fa.domains == fb.domains and list(fa.quants) == list(fb.quants) == fa.domains
        n_multi += len(fa.domains) > 1
        for k in fa.domains:
            assert fa.quants[k].dtype == np.int64 == fb.quants[k].dtype
            np.testing.assert_array_equal(fa.quants[k], fb.quants[k])
        assert ra[0] == rb[0] == fa.pid and ra[1] == rb[1] == fa.domains
        assert ra[2].dtype == np.int8 and ra[2].shape == (len(fa.domains), 480)
        np.testing.assert_array_equal(ra[2], rb[2])
        np.testing.assert_array_equal(ra[2].astype(np.int64), np.array([fa.quants[k] for k in fa.domains]))
    assert n_multi >= 100
    np.testing.assert_array_equal(make_db._records(a)[7][2], recs[7][2])
    # what the plain flush does not take goes the general way, with the same results: numpy inputs, domains given by the caller
    c = fresh()
    for fp in c[:5]:
        fp.embed = {k: v.cpu().numpy() for k, v in fp.embed.items()}
    c = make_db.fingerprint_batch(c, threads=4)
    assert make_db.LAST_PATH[0] == 'generic'
    d = fresh()
    d[2].domains = ['1-20']
    d = make_db.fingerprint_batch(d, threads=4)
    assert make_db.LAST_PATH[0] == 'generic' and d[2].domains[0] == '1-20'
    for fa, fc in zip(a, c):
        assert fa.domains == fc.domains
        for k in fa.domains:
            np.testing.assert_array_equal(fa.quants[k], fc.quants[k])


def test_process_sequences_with_deferred_second_halves_keeps_the_order():
    """process_sequences finishes a flush when the next one has been started: the records must reach the writer in sequence order
    and equal those of one big flush."""
    from dctdomain_amd import make_db
    from dctdomain_amd.embedding import SyntheticModel
    rng = np.random.default_rng(9)
    aa = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
    seqs = [(f's{i:03d}', aa[rng.integers(0, 20, size=int(L))].tobytes().decode())
            for i, L in enumerate(np.clip(rng.gamma(2.0, 120.0, size=90).astype(int), 25, 900))]
    dev = torch.device('cuda', 0)
    model = SyntheticModel()
    model.to_device(dev)

    def run(flush):
        got, sizes = [], []
        make_db.process_sequences(seqs, model, dev, 500, 4, flush, lambda recs: (got.extend(recs), sizes.append(len(recs))))
        return got, sizes
    small, sizes = run(16)
    big, one = run(10 ** 6)
    assert len(sizes) >= 4 and len(one) == 1 and [r[0] for r in small] == [p for p, _ in seqs] == [r[0] for r in big]
    for x, y in zip(small, big):
        assert x[1] == y[1]
        np.testing.assert_array_equal(x[2], y[2])


def test_cutter_in_flight_while_the_callers_stream_uses_the_context():
    """The first half of a flush leaves its contact selection and cutter running on a side stream; the caller's stream goes on using
    the SAME context -- window stitching (job tables in the context's double-buffered table block), another domain prediction
    (the cutter's adjacency scratch), a fingerprint call.  Every one of them must queue behind the readers of what it overwrites
    (a build once did not: stitch_impl uploaded its table over the cutter's) -- the flush's results must be
    those of the flush run alone."""
    from dctdomain_amd import make_db, reccut
    from dctdomain_amd.embedding import stitch_embeddings_batch, stitch_contacts_batch
    rng = np.random.default_rng(12)
    lens = np.clip(rng.gamma(2.2, 170.0, size=700).astype(int), 60, 1300).tolist()
    fresh = _embedded(lens, seed=8)
    alone = make_db.flush_records(fresh(), threads=4)
    dev = torch.device('cuda', 0)
    wins = [[torch.randn(500, 640, device=dev), torch.randn(350, 640, device=dev)] for _ in range(3)]
    cwins = [[torch.rand(500, 500, device=dev), torch.rand(400, 400, device=dev)]]
    other = [fp.contacts for fp in fresh()[:40]]
    other_alone = reccut.domains_from_maps(other, 2.6)
    for rep in range(3):
        fl = make_db._Flush(fresh(), threads=4)
        assert fl.start()
        # ... the caller's stream, while the cutter runs: both table buffers several times over, the cutter's scratch, a quantize
        for _ in range(4):
            stitch_embeddings_batch(wins)
            stitch_contacts_batch(cwins, 300)
        assert reccut.domains_from_maps(other, 2.6) == other_alone
        few = make_db.flush_records(fresh()[:8], threads=2)
        recs = fl.finish(objects=False)
        assert recs is not None and len(recs) == len(alone)
        for a, b in zip(alone, recs):
            assert a[0] == b[0] and a[1] == b[1]
            np.testing.assert_array_equal(a[2], b[2])
        for a, b in zip(alone[:8], few):
            assert a[1] == b[1]
            np.testing.assert_array_equal(a[2], b[2])
