"""Parity soak of a database flush as make_db runs it (the two-phase flush: contact top-k -> the domain cutter on the GPU -> strings
+ pieces -> dctfp_quantize -> records) against the CPU chain on the host cores: oracle top-k -> THE REFERENCE'S RecCut BINARY
(oracle/_ref/RecCut) -> oracle quantize (scipy.fft, like the reference).  N random proteins with pfam-like lengths, contact maps
with block structure (so that the cutter finds domains) over a decaying band; counts mismatching domain lists and int8 values.
Checker use of oracle/ only (a test tool, not product code).
usage: python tools/parity_soak_flush.py [n_proteins] [procs] [D]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def make_protein(seed, D):
    rng = np.random.default_rng(seed)
    L = int(np.clip(rng.gamma(2.2, 170.0), 30, 1400))
    i = np.arange(L)
    blk = int(rng.integers(70, 180))
    near = 0.9 * np.exp(-np.abs(i[:, None] - i[None, :]) / 12.0)
    same = (i[:, None] // blk) == (i[None, :] // blk)
    cm = near + 0.3 * rng.random((L, L)) * same + 0.02 * rng.random((L, L))
    if rng.random() < 0.3 and L >= 3 * blk:   # head and tail fold onto each other (the flanks of a double cut: one discontinuous domain)
        head, tail = i < blk, i >= L - blk
        cross = (head[:, None] & tail[None, :]) | (tail[:, None] & head[None, :])
        cm = np.where(cross & (rng.random((L, L)) < 0.08), 0.65 + 0.3 * rng.random((L, L)), cm)
    if rng.random() < 0.25:         # plateaus: a map quantised to a hundred levels (ties at the selection's threshold)
        cm = np.round(cm * 97) / 97
    cm = np.clip(0.5 * (cm + cm.T), 0, 1).astype(np.float32)
    ls = [(rng.standard_normal((L, D)) * np.exp(rng.standard_normal(D)) + 5 * rng.standard_normal(D)).astype(np.float32) for _ in range(2)]
    return L, cm, ls


def _oracle_chunk(args):
    seed0, count, D = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import contacts_oracle as co
    from oracle import dct_oracle as orc
    out = []
    for k in range(count):
        L, cm, ls = make_protein(seed0 + k, D)
        ci, cj, cv = co.top_contacts(cm, 2.6)
        pid = f's{seed0 + k}'
        rc, txt = co.run_ref_binary(co.ce_text(pid, 'A' * L, ci, cj, cv), pid)
        assert rc == 0, (pid, rc)
        doms = co.parse_reccut(txt, L)
        q = orc.quantize(ls, doms, [3, 80, 3, 80])
        out.append((doms, np.stack([np.asarray(q[d]).astype(np.int8) for d in q]), list(q)))
    return out


if __name__ == '__main__':
    import multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    from oracle import contacts_oracle as co
    if not os.path.exists(co.REF_BIN):
        raise SystemExit('oracle/_ref/RecCut is not built (make -C oracle): this soak checks against the reference binary')
    t0 = time.time()
    B = 256
    jobs = [(70_000 + b0, min(B, n - b0), D) for b0 in range(0, n, B)]
    with mp.get_context('spawn').Pool(procs) as pool:
        async_res = pool.map_async(_oracle_chunk, [(s + i, 1, D) for s, c, _ in jobs for i in range(c)], chunksize=4)
        import torch
        import dctdomain_amd as dd
        from dctdomain_amd import make_db, reccut
        dev = torch.device('cuda', 0)
        got, redo, paths = [], 0, set()
        for seed0, count, _ in jobs:
            fps = []
            for k in range(count):
                L, cm, ls = make_protein(seed0 + k, D)
                fps.append(dd.Fingerprint(pid=f's{seed0 + k}', seq='A' * L, embed={15: torch.from_numpy(ls[0]).to(dev), 21: torch.from_numpy(ls[1]).to(dev)},
                                          contacts=torch.from_numpy(cm).to(dev)))
            got.extend(make_db.flush_records(fps, threads=4))
            paths.add(make_db.LAST_PATH[0])
            redo += len(reccut.LAST.host_redo)
        t_gpu = time.time() - t0
        want = [r[0] for r in async_res.get()]
    bad_doms = bad_rows = bad_values = n_fp = n_multi = n_disc = 0
    for (pid, doms, rows8), (wdoms, wrows, wkeys) in zip(got, want):
        n_fp += len(wdoms)
        n_multi += len(wdoms) > 1
        n_disc += any(',' in d for d in wdoms)
        if doms != wdoms or doms != wkeys:
            bad_doms += 1
            continue
        diff = rows8 != wrows
        bad_rows += int(diff.any(axis=1).sum())
        bad_values += int(diff.sum())
    print(f'{n} proteins at D = {D} (lengths 30-1400, {n_multi} with several domains, {n_disc} with a discontinuous one; a quarter of the maps '
          f'on 97 levels) through make_db.flush_records (path {sorted(paths)}, {redo} proteins redone by the host library) against oracle top-k -> '
          f'the reference\'s RecCut binary -> oracle quantize on {procs} processes: {n_fp} fingerprints, {n_fp * 480} int8 values; '
          f'mismatching domain lists {bad_doms}, mismatching fingerprints {bad_rows}, mismatching values {bad_values}; '
          f'GPU side {t_gpu:.1f} s, all {time.time() - t0:.1f} s')
    sys.exit(1 if bad_doms or bad_values else 0)
