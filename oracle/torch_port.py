"""The reference's own torch calls for the hot path, restated -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

bench.py's cpu_baseline leg times these on the GPU box's host CPU ("kind": "port"); tests use them as a
second checker.  textgcn_amd never imports this module.  Each function cites the reference lines it repeats
(/root/reference paths).  Parity status: pinned through tests/test_oracle_golden.py (same golden vectors).
"""
import numpy as np
import torch


def norm_matrix(idx, val, n):
    """TextGCN/dataset.py:151-157 + :138 -- coalesced fp32 sparse COO with int64 indices."""
    t = torch.sparse_coo_tensor(torch.from_numpy(np.ascontiguousarray(idx)), torch.from_numpy(np.ascontiguousarray(val)),
                                (n, n))
    return t.coalesce()


@torch.no_grad()
def representation(a, e0, n_layers, single=False):
    """TextGCN/base_model.py:93-106: K x torch.sparse.mm (:148), then mean(stack) (:157) or last layer (:164)."""
    cur = e0
    cache = [cur]
    for _ in range(n_layers):
        cur = torch.sparse.mm(a, cur)
        cache.append(cur)
    return cache[-1] if single else torch.mean(torch.stack(cache), axis=0)


@torch.no_grad()
def score_mask_topk(users_emb, items_emb, mask_rowptr, mask_items, k):
    """TextGCN/base_model.py:254-263: matmul, -inf on train items, topk, round(4)."""
    rating = torch.matmul(users_emb, items_emb.t())
    rows = np.repeat(np.arange(len(mask_rowptr) - 1), np.diff(mask_rowptr))
    rating[torch.from_numpy(rows), torch.from_numpy(np.asarray(mask_items, dtype=np.int64))] = -np.inf
    probs, idx = torch.topk(rating, k=k)
    return probs.round(decimals=4), idx


@torch.no_grad()
def ltr_score_batchwise(users_emb, users_reviews, users_desc, items_emb, items_reviews, items_desc, weight, bias):
    """TextGCN/ltr_models.py:131-146 (five [B, I] products -> [B, I, 5]) + :200-204 (nn.Linear(5, 1), squeeze); the
    item-side `table[range(I)]` copies of :103-109 included (all_items is a range, dataset.py:108)."""
    all_items = range(items_emb.shape[0])
    i_desc, i_rev = items_desc[all_items], items_reviews[all_items]
    feats = torch.cat([(users_emb @ items_emb.T).unsqueeze(-1), (users_reviews @ i_rev.T).unsqueeze(-1),
                       (users_desc @ i_desc.T).unsqueeze(-1), (users_reviews @ i_desc.T).unsqueeze(-1),
                       (users_desc @ i_rev.T).unsqueeze(-1)], axis=-1)
    return torch.nn.functional.linear(feats, weight, bias).squeeze()
