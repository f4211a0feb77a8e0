"""Ranking metrics @k on top-k lists: recall, precision, hit, ndcg, f1 -- the numbers the reference's
`evaluate()` logs (TextGCN/utils.py:11-63, calculate_metrics), computed with numpy arrays instead of a
pandas apply per user.  Definitions (per user, then mean over users):
  recall = |pred[:k] ∩ true| / |true|      precision = |pred[:k] ∩ true| / k      hit = [intersection non-empty]
  ndcg   = DCG(rel) / DCG(ideal), rel_j = [pred_j ∈ true], gain 2^rel - 1, discount log2(j + 2)
  f1     = 2 r p / (r + p), 0 when r + p = 0
"""
import numpy as np

METRICS = ('recall', 'precision', 'hit', 'ndcg', 'f1')


def ranking_metrics(y_true, y_pred, ks):
    """y_true: list (per user) of lists of relevant item ids; y_pred: [n_users, kmax] array of ranked ids."""
    y_pred = np.asarray(y_pred)
    n = len(y_true)
    if y_pred.shape[0] != n:
        raise ValueError('y_true and y_pred differ in the number of users')
    n_true = np.array([len(set(t)) for t in y_true], dtype=np.float64)
    kmax = y_pred.shape[1]
    # relevance matrix [n, kmax]
    rel = np.zeros((n, kmax), dtype=np.float64)
    for r, (t, p) in enumerate(zip(y_true, y_pred)):
        rel[r] = np.isin(p, np.asarray(list(t)))
        # a predicted id repeated in the list counts once (np.intersect1d semantics of the reference)
        _, first = np.unique(p, return_index=True)
        keep = np.zeros(kmax, dtype=bool)
        keep[first] = True
        rel[r] *= keep
    out = {m: [] for m in METRICS}
    for k in sorted(ks):
        if k > kmax:
            raise ValueError(f'k={k} exceeds the length of the prediction lists ({kmax})')
        hits = rel[:, :k].sum(axis=1)
        rec = hits / n_true
        prec = hits / k
        disc = 1.0 / np.log2(np.arange(2, k + 2))
        dcg = (rel[:, :k] * disc).sum(axis=1)
        ideal = np.array([disc[:int(min(t, k))].sum() for t in n_true])
        with np.errstate(invalid='ignore', divide='ignore'):
            f1 = np.where(rec + prec > 0, 2 * rec * prec / (rec + prec), 0.0)
        out['recall'].append(float(rec.mean()))
        out['precision'].append(float(prec.mean()))
        out['hit'].append(float((hits > 0).mean()))
        out['ndcg'].append(float((dcg / ideal).mean()))
        out['f1'].append(float(f1.mean()))
    return out


def early_stop(history):
    """TextGCN/utils.py:79-90: stop when the last three evaluations are within 1e-4 of each other on every
    metric, or strictly declining on every metric.  history: dict metric -> array [n_evals, n_k]."""
    if len(history['recall']) < 3:
        return False
    vals = list(history.values())
    declining = all((m[-1] < m[-2]).all() and (m[-2] < m[-3]).all() for m in vals)
    flat = all(np.allclose(m[-1], m[-2], atol=1e-4) and np.allclose(m[-1], m[-3], atol=1e-4) for m in vals)
    return flat or declining
