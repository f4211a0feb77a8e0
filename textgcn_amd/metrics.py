"""Ranking metrics @k on top-k lists: recall, precision, hit, ndcg, f1 -- the numbers the reference's
`evaluate()` logs (TextGCN/utils.py:11-63, calculate_metrics).  Definitions (per user, then mean over users), exactly as
the reference computes them:
  y_true_len        = len(y_true)                      -- duplicate test rows count (utils.py:38)
  intersecting_len  = |np.intersect1d(pred[:k], y_true)|  -- distinct common items (utils.py:45-47)
  recall = intersecting_len / y_true_len   precision = intersecting_len / k   hit = [intersecting_len > 0]
  ndcg   = DCG(rel) / DCG(ideal): rel_j = [pred_j in intersection] for every position j (utils.py:33), ideal = ones in the
           first min(y_true_len, k) positions (utils.py:29-32), gain 2^rel - 1, discount log2(j + 2)
  f1     = 2 r p / (r + p), 0 when r + p = 0

Two implementations of the same arithmetic: `ranking_metrics` (numpy, lists in) and `ranking_metrics_device` (torch ops on
the top-k tensor `predict_tensors` leaves on the GPU, so `evaluate()` never converts [users, kmax] to Python lists).
"""
import numpy as np

METRICS = ('recall', 'precision', 'hit', 'ndcg', 'f1')


def ranking_metrics(y_true, y_pred, ks):
    """y_true: list (per user) of lists of relevant item ids; y_pred: [n_users, kmax] array of ranked ids."""
    y_pred = np.asarray(y_pred)
    n = len(y_true)
    if y_pred.shape[0] != n:
        raise ValueError('y_true and y_pred differ in the number of users')
    n_true = np.array([len(t) for t in y_true], dtype=np.float64)
    kmax = y_pred.shape[1]
    out = {m: [] for m in METRICS}
    for k in sorted(ks):
        if k > kmax:
            raise ValueError(f'k={k} exceeds the length of the prediction lists ({kmax})')
        disc = 1.0 / np.log2(np.arange(2, k + 2))
        hits = np.zeros(n)
        dcg = np.zeros(n)
        for r, (t, p) in enumerate(zip(y_true, y_pred)):
            inter = np.intersect1d(p[:k], np.asarray(list(t)))
            hits[r] = len(inter)
            dcg[r] = (np.isin(p[:k], inter) * disc).sum()
        rec = hits / n_true
        prec = hits / k
        ideal = np.array([disc[:int(min(t, k))].sum() for t in n_true])
        with np.errstate(invalid='ignore', divide='ignore'):
            f1 = np.where(rec + prec > 0, 2 * rec * prec / (rec + prec), 0.0)
        out['recall'].append(float(rec.mean()))
        out['precision'].append(float(prec.mean()))
        out['hit'].append(float((hits > 0).mean()))
        out['ndcg'].append(float((dcg / ideal).mean()))
        out['f1'].append(float(f1.mean()))
    return out


def true_lists_csr(y_true):
    """(ptr int64 [n+1], items int64) of the per-user relevant lists, duplicates kept"""
    ptr = np.zeros(len(y_true) + 1, dtype=np.int64)
    np.cumsum([len(t) for t in y_true], out=ptr[1:])
    items = np.concatenate([np.asarray(t, dtype=np.int64) for t in y_true]) if len(y_true) else np.zeros(0, dtype=np.int64)
    return ptr, items


def ranking_metrics_device(true_ptr, true_items, pred, ks, n_items):
    """The same numbers from device tensors: true_ptr int64 [n+1] / true_items int64 (CSR of the relevant lists, rows in
    the order of `pred`'s rows), pred int64 [n, kmax] ranked DISTINCT item ids per row (what a top-k returns; with distinct
    predictions the reference's intersect1d / isin pair reduces to a membership test per position); n_items bounds the ids, and
    an entry outside [0, n_items) means "no item" and counts as a miss.
    float64 on the device; only the 5 x len(ks) means cross to the host."""
    import torch
    n, kmax = pred.shape
    if true_ptr.numel() != n + 1:
        raise ValueError('true_ptr and pred differ in the number of users')
    dev = pred.device
    n_true = (true_ptr[1:] - true_ptr[:-1]).to(torch.float64)
    span = int(n_items)
    rows_true = torch.repeat_interleave(torch.arange(n, device=dev), true_ptr[1:] - true_ptr[:-1], output_size=int(true_items.numel()))
    keys_true = torch.sort(rows_true * span + true_items)[0]
    # an id outside [0, n_items) is "no item" (TGCN_NO_ITEM = 2^31 - 1 fills the list positions of a row with fewer than k
    # rankable scores, include/tgcn.h): it must never match -- as a key it would land in ANOTHER user's range
    keys_pred = torch.where((pred >= 0) & (pred < span), torch.arange(n, device=dev)[:, None] * span + pred, torch.full_like(pred, -1))
    if keys_true.numel():
        pos = torch.searchsorted(keys_true, keys_pred.reshape(-1)).clamp_(max=keys_true.numel() - 1)
        rel = (keys_true[pos] == keys_pred.reshape(-1)).reshape(n, kmax).to(torch.float64)
    else:
        rel = torch.zeros((n, kmax), dtype=torch.float64, device=dev)
    disc = 1.0 / torch.log2(torch.arange(2, kmax + 2, device=dev, dtype=torch.float64))
    ideal_cum = torch.cumsum(disc, 0)
    cols = []
    for k in sorted(ks):
        if k > kmax:
            raise ValueError(f'k={k} exceeds the length of the prediction lists ({kmax})')
        hits = rel[:, :k].sum(dim=1)
        rec = hits / n_true
        prec = hits / k
        dcg = (rel[:, :k] * disc[:k]).sum(dim=1)
        ideal = ideal_cum[(torch.clamp(n_true, max=k).to(torch.int64) - 1).clamp_(min=0)]
        den = rec + prec
        f1 = torch.where(den > 0, 2 * rec * prec / torch.where(den > 0, den, torch.ones_like(den)), torch.zeros_like(den))
        cols.append(torch.stack([rec.mean(), prec.mean(), (hits > 0).to(torch.float64).mean(), (dcg / ideal).mean(), f1.mean()]))
    vals = torch.stack(cols, dim=1).cpu().numpy()      # [5, len(ks)]: the one device-to-host copy
    return {m: [float(x) for x in vals[j]] for j, m in enumerate(METRICS)}


def early_stop(history):
    """TextGCN/utils.py:79-90: stop when the last three evaluations are within 1e-4 of each other on every
    metric, or strictly declining on every metric.  history: dict metric -> array [n_evals, n_k]."""
    if len(history['recall']) < 3:
        return False
    vals = list(history.values())
    declining = all((m[-1] < m[-2]).all() and (m[-2] < m[-3]).all() for m in vals)
    flat = all(np.allclose(m[-1], m[-2], atol=1e-4) and np.allclose(m[-1], m[-3], atol=1e-4) for m in vals)
    return flat or declining
