"""CPU oracle for the chunk stitcher -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates mgtools/DCTdomain src/embedding.py with plain torch-CPU float32 operations:
* ``split_seq``         :83-100
* ``combine_contacts``  :123-150
* ``stitch``            the window loop of ``embed_seq`` :165-188
Pinned by tests/golden/stitch_golden.* which were produced by the reference's own ``Embedding``
class (see tests/golden/make_golden_stitch.py for how it is executed without the esm package)."""

import torch


def split_seq(seq: str, maxlen: int, overlap: int):
    out = []
    for i in range(0, len(seq), maxlen - overlap):
        sub = seq[i:i + maxlen]
        if len(sub) > overlap:
            out.append(sub)
    return out


def combine_contacts(mat1, mat2, inc, times):
    olp = inc * times
    n1, n2 = mat1.size(0), mat2.size(0)
    n3 = olp + n2
    new = torch.zeros((n3, n3))
    new[:n1, :n1] = mat1
    new[olp:n3, olp:n3] = new[olp:n3, olp:n3] + mat2
    new[olp:n1, olp:n1] = new[olp:n1, olp:n1] / 2
    return new


def stitch_embeddings(windows, olp=200):
    """One layer: ``run[-olp:] = (run[-olp:] + new[:olp]) / 2; run = cat(run, new[olp:])`` over the windows (:185-187),
    torch-CPU float32 as in the reference."""
    run = windows[0].clone()
    for emb in windows[1:]:
        run[-olp:] = (run[-olp:] + emb[:olp]) / 2
        run = torch.cat((run, emb[olp:]), axis=0)
    return run


def stitch(window_embeds, window_contacts, maxlen, olp=200):
    """window_embeds: list (per window) of {layer: (W, D) float32}; window_contacts: list of (W, W)."""
    edata = None
    for i, (embs, ct) in enumerate(zip(window_embeds, window_contacts)):
        if edata is None:
            edata = {k: v.clone() for k, v in embs.items()}
            edata['ct'] = ct.clone()
            continue
        for lay, emb in embs.items():
            edata[lay][-olp:] = (edata[lay][-olp:] + emb[:olp]) / 2
            edata[lay] = torch.cat((edata[lay], emb[olp:]), axis=0)
        edata['ct'] = combine_contacts(edata['ct'], ct, maxlen - olp, i)
    ct = edata.pop('ct')
    return edata, ct
