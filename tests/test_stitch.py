"""Chunk + overlap-average stitcher against golden vectors produced by the
reference's own Embedding class (tests/golden/make_golden_stitch.py)."""

import hashlib
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import stitch_oracle as so
from synthetic_esm import SyntheticESM, make_sequence

with open(os.path.join(gu.GOLD, 'stitch_golden.json')) as fh:
    CASES = json.load(fh)['cases']
ARR = np.load(os.path.join(gu.GOLD, 'stitch_golden.npz'))
IDS = [c['id'] for c in CASES]


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def windows_of(case, device='cpu'):
    seq = make_sequence(case['L'], case['seq_seed'])
    subs = so.split_seq(seq, case['maxlen'], 200) if case['L'] > case['maxlen'] else [seq]
    assert [len(s) for s in subs] == case['windows']
    model = SyntheticESM(dim=32, device=device)
    embs, cts = [], []
    for s in subs:
        _, _, tok = model.esm_tokenizer([('p', s)])
        res = model.esm_encoder(tok)
        embs.append({l: res['representations'][l][0][1:-1] for l in (15, 21)})
        cts.append(res['contacts'][0])
    return seq, embs, cts


@pytest.mark.parametrize('case', CASES, ids=IDS)
def test_oracle_matches_reference_class(case):
    _, embs, cts = windows_of(case)
    ed, ct = so.stitch(embs, cts, case['maxlen'])
    assert list(ed[15].shape) == case['embed_shape'] and list(ct.shape) == case['contacts_shape']
    for l in (15, 21):
        assert sha(ed[l].numpy()) == case['embed_sha'][str(l)]
    assert sha(ct.numpy()) == case['contacts_sha']
    if f"{case['id']}/e15" in ARR:
        np.testing.assert_array_equal(ed[15].numpy(), ARR[f"{case['id']}/e15"])


def test_oracle_combine_contacts_small():
    m1, m2 = torch.from_numpy(ARR['combine/m1']), torch.from_numpy(ARR['combine/m2'])
    np.testing.assert_array_equal(so.combine_contacts(m1, m2, 3, 1).numpy(), ARR['combine/out_inc3_t1'])
    np.testing.assert_array_equal(so.combine_contacts(m1, m2, 2, 2).numpy(), ARR['combine/out_inc2_t2'])


def test_split_seq_hand_worked():
    """SURVEY 8f-1 worked examples: L=750 -> 0:500, 300:750 (600:750 dropped); L=1000 -> three windows."""
    from dctdomain_amd.embedding import Embedding
    for L, exp in ((750, [500, 450]), (1000, [500, 500, 400]), (200, []), (201, [201]), (501, [500, 201])):
        seq = 'A' * L
        assert [len(s) for s in Embedding(pid='x', seq=seq).split_seq(500, 200)] == exp
        assert [len(s) for s in so.split_seq(seq, 500, 200)] == exp


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=IDS)
def test_gpu_embed_seq_matches_reference_class(case):
    from dctdomain_amd.embedding import Embedding
    seq = make_sequence(case['L'], case['seq_seed'])
    model = SyntheticESM(dim=32, device='cuda')
    emb = Embedding(pid='p', seq=seq)
    emb.embed_seq(model, 'cuda', [15, 21], case['maxlen'])
    assert list(emb.embed[15].shape) == case['embed_shape']
    for l in (15, 21):
        assert emb.embed[l].is_cuda and emb.embed[l].dtype == torch.float32
        assert sha(emb.embed[l].cpu().numpy()) == case['embed_sha'][str(l)]
    assert sha(emb.contacts.cpu().numpy()) == case['contacts_sha']


@pytest.mark.gpu
def test_gpu_combine_contacts_and_batch():
    from dctdomain_amd.embedding import Embedding, stitch_contacts_batch, stitch_embeddings_batch
    e = Embedding(pid='c', seq='A' * 10)
    m1, m2 = torch.from_numpy(ARR['combine/m1']).cuda(), torch.from_numpy(ARR['combine/m2']).cuda()
    np.testing.assert_array_equal(e.combine_contacts(m1, m2, 3, 1).cpu().numpy(), ARR['combine/out_inc3_t1'])
    np.testing.assert_array_equal(e.combine_contacts(m1, m2, 2, 2).cpu().numpy(), ARR['combine/out_inc2_t2'])
    # a whole batch of sequences in one call (one launch per window index) equals per-sequence results
    sel = [c for c in CASES if c['maxlen'] == 500]
    ew, cw = [], []
    for c in sel:
        _, embs, cts = windows_of(c, device='cuda')
        ew.append([w[15] for w in embs])
        cw.append(cts)
    outs = stitch_embeddings_batch(ew)
    couts = stitch_contacts_batch(cw, 300)
    for c, o, co_ in zip(sel, outs, couts):
        assert sha(o.cpu().numpy()) == c['embed_sha']['15']
        assert sha(co_.cpu().numpy()) == c['contacts_sha']


@pytest.mark.gpu
def test_gpu_batch_one_launch_and_sequential_form():
    """Round 4: where every window overlaps its neighbours only (rows >= 2 * overlap for the windows in the middle), the
    whole batch goes out in ONE launch that averages shared rows from the two windows; otherwise one launch per window
    index.  Both against the reference's expression evaluated with torch (src/embedding.py:185-187), bit for bit: widths
    that are / are not a multiple of 4, row-strided windows, a middle window of 399 rows (three windows meet: the
    sequential form), a first window of exactly `overlap` rows, single-window sequences."""
    from dctdomain_amd.embedding import stitch_embeddings_batch
    g = torch.Generator(device='cuda')
    g.manual_seed(3)

    def ref(ws, step=200):
        run = ws[0].clone()
        for w in ws[1:]:
            run[-step:] = (run[-step:] + w[:step]) / 2
            run = torch.cat([run, w[step:]])
        return run

    for width, shapes in ((64, [[500, 500, 500, 300], [400, 400, 201], [500], [200, 350], [1022, 1022, 700]]),
                          (50, [[500, 450], [401, 400, 400, 400, 250]]),
                          (64, [[500, 399, 500], [500, 500]]),                 # 399 < 2 * 200: sequential form for the call
                          (33, [[300, 250, 260, 201]])):                        # the same, odd width
        batch = []
        for rows in shapes:
            ws = []
            for i, r in enumerate(rows):
                if i % 2:   # a row-strided view of a wider tensor
                    ws.append(torch.randn((r, width + 12), device='cuda', generator=g)[:, 4:4 + width])
                else:
                    ws.append(torch.randn((r, width), device='cuda', generator=g))
            batch.append(ws)
        outs = stitch_embeddings_batch(batch)
        for ws, o in zip(batch, outs):
            e = ref(ws)
            assert o.shape == e.shape
            assert torch.equal(o, e), (width, [w.shape[0] for w in ws])


@pytest.mark.gpu
def test_gpu_chunked_pipeline_end_to_end():
    """BASELINE config 3 flavour: long sequence -> windows -> stitch -> reccut -> quantize, all on the
    GPU, against the oracles chained on the CPU."""
    import dctdomain_amd as dd
    from dctdomain_amd.embedding import Embedding
    from oracle import contacts_oracle as co
    from oracle import dct_oracle as orc
    L, maxlen = 1035, 500
    seq = make_sequence(L, 99)
    emb = Embedding(pid='long', seq=seq)
    emb.embed_seq(SyntheticESM(dim=96, device='cuda'), 'cuda', [15, 21], maxlen)
    fp = dd.Fingerprint(pid='long', seq=seq, embed=emb.embed, contacts=emb.contacts)
    fp.reccut(2.6)
    fp.quantize([3, 80, 3, 80])
    # CPU chain
    model = SyntheticESM(dim=96)
    subs = so.split_seq(seq, maxlen, 200)
    embs, cts = [], []
    for s in subs:
        _, _, tok = model.esm_tokenizer([('p', s)])
        res = model.esm_encoder(tok)
        embs.append({l: res['representations'][l][0][1:-1] for l in (15, 21)})
        cts.append(res['contacts'][0])
    ed, ct = so.stitch(embs, cts, maxlen)
    ci, cj, cv = co.top_contacts(ct.numpy(), 2.6)
    if os.path.exists(co.REF_BIN):
        rc, out = co.run_ref_binary(co.ce_text('long', seq, ci, cj, cv), 'long')
        assert rc == 0
        doms = co.parse_reccut(out, L)
        q = orc.quantize([ed[15].numpy(), ed[21].numpy()], doms, [3, 80, 3, 80])
        assert fp.domains == list(q.keys())
        for k in q:
            np.testing.assert_array_equal(fp.quants[k], q[k])
    else:
        q = orc.quantize([ed[15].numpy(), ed[21].numpy()], fp.domains, [3, 80, 3, 80])
        for k in q:
            np.testing.assert_array_equal(fp.quants[k], q[k])
