"""Seeded synthetic bipartite interaction graphs for the BASELINE.json configs.

users uniform; items Zipf-like (p_i ~ (i+1)^-0.8 on a random permutation of item ids) or uniform; exactly
`nnz` distinct (user, item) pairs, plus one extra edge for every node that would otherwise be isolated
(so the realised count can exceed `nnz` by a handful -- it is reported, never assumed).
"""
import numpy as np

CONFIGS = {
    # name: (n_users, n_items, nnz, d, n_layers)        BASELINE.json configs[1..3]
    'c2': (100_000, 50_000, 5_000_000, 64, 3),
    'c3': (180_000, 60_000, 1_600_000, 128, 4),
    'c4': (5_000_000, 2_000_000, 100_000_000, 64, 3),
    # cache experiment: same entry count as c2 on a 4 MB embedding table (L2-resident gathers)
    'l2': (12_000, 4_000, 5_000_000, 64, 3),
    # small shapes for tests / smoke
    'tiny': (300, 200, 4_000, 64, 3),
    'small': (5_000, 3_000, 150_000, 64, 3),
}


def interactions(n_users, n_items, nnz, seed=0, zipf=0.8):
    """Returns (u int64 [m], i int64 [m]) sorted by (u, i), m >= nnz (see module doc)."""
    if nnz > n_users * n_items:
        raise ValueError('nnz exceeds the number of possible pairs')
    rng = np.random.default_rng(seed)
    if zipf:
        p = (np.arange(n_items, dtype=np.float64) + 1.0) ** -float(zipf)
        cdf = np.cumsum(p / p.sum())
        cdf[-1] = 1.0
        perm = rng.permutation(n_items)
    keys = np.zeros(0, dtype=np.int64)
    need = nnz
    while need > 0:
        m = int(need * 1.05) + 16
        u = rng.integers(0, n_users, size=m, dtype=np.int64)
        if zipf:
            it = perm[np.searchsorted(cdf, rng.random(m), side='right').clip(max=n_items - 1)]
        else:
            it = rng.integers(0, n_items, size=m, dtype=np.int64)
        new = np.unique(u * np.int64(n_items) + it)
        if len(keys):
            new = new[~np.isin(new, keys, assume_unique=True)]
        if len(new) > need:  # drop a random surplus, not the largest keys
            new = np.sort(rng.choice(new, size=need, replace=False))
        keys = np.union1d(keys, new) if len(keys) else new
        need = nnz - len(keys)
    u, it = np.divmod(keys, np.int64(n_items))
    iso_u = np.setdiff1d(np.arange(n_users), u, assume_unique=False)
    iso_i = np.setdiff1d(np.arange(n_items), it, assume_unique=False)
    if len(iso_u) or len(iso_i):
        eu = np.concatenate([iso_u, rng.integers(0, n_users, size=len(iso_i), dtype=np.int64)])
        ei = np.concatenate([rng.integers(0, n_items, size=len(iso_u), dtype=np.int64), iso_i])
        keys = np.union1d(keys, eu * np.int64(n_items) + ei)
        u, it = np.divmod(keys, np.int64(n_items))
    return u, it


def embeddings(n, d, seed=0, std=0.1):
    """E0 ~ N(0, std^2) fp32 from a CPU torch generator (distribution of base_model.py:68-69)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    return torch.randn((n, d), generator=g, dtype=torch.float32) * std
