"""Bigger one-off fuzz of libreccut against the compiled reference binary, incl. tie-rich graphs (few distinct weights)."""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from oracle import contacts_oracle as co
from dctdomain_amd import reccut
seed = int(sys.argv[1]); n = int(sys.argv[2])
rng = np.random.default_rng(seed)
n_multi = bad = 0
for it in range(n):
    L = int(rng.integers(22, 420))
    nb = int(rng.integers(1, 8))
    bounds = np.sort(rng.choice(np.arange(1, L), size=min(nb - 1, L - 1), replace=False)) if nb > 1 else []
    lab = np.zeros(L, int)
    for b in bounds: lab[b:] += 1
    for _m in range(int(rng.integers(0, 3))):
        if nb >= 3:
            a, b = sorted(rng.choice(nb, 2, replace=False)); lab[lab == b] = a
    same = lab[:, None] == lab[None, :]
    p = rng.random((L, L)) * (same * rng.uniform(0.5, 1.0) + (~same) * rng.uniform(0.0, 0.3))
    if it % 3 == 0:          # tie-rich: a handful of distinct probabilities -> many equal ratios in the scans
        p = np.round(p * 4) / 4
    if it % 7 == 0:          # periodic structure: exact symmetries between candidate cuts
        k = int(rng.integers(30, 80)); base = rng.random((k, k)); reps = L // k + 1
        p = np.tile(np.round(base * 5) / 5, (reps, reps))[:L, :L]
    mask = np.triu(rng.random((L, L)) < rng.uniform(0.02, 0.3), 5)
    ii, jj = np.nonzero(mask)
    t = int(2.6 * L)
    if len(ii) > t:
        order = np.argsort(-p[ii, jj], kind='stable')[:t]; ii, jj = ii[order], jj[order]
    pv = p[ii, jj].astype(np.float32)
    rc, out = co.run_ref_binary(co.ce_text('x', 'A' * L, ii, jj, pv))
    try:
        got = reccut.domains_from_contacts([L], [0, len(ii)], ii, jj, pv)[0]
    except RuntimeError as e:
        got = ['ERR']
    if rc != 0:
        continue
    exp = out.strip().split()[2].split(';')[:-1]
    if got != exp:
        bad += 1; print('MISMATCH', seed, it, L, exp, got)
    n_multi += len(exp) > 1
print(f'seed {seed}: {n} cases, {n_multi} multi-domain, {bad} mismatching')
