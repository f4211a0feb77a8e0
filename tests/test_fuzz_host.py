"""Seeded random sweeps of the host side against the oracle (CPU only): the graph builder on unsorted pair lists with repeated
pairs and isolated nodes, the train-mask CSR, the ranking metrics in both forms, and the plans that reorder a row's sum (long-row
split, XCD segments) replayed on the host."""
import numpy as np
import pytest

from conftest import bits

SEEDS = range(40)


def _pairs(seed):
    rng = np.random.default_rng(40_000 + seed)
    n_u, n_i = int(rng.choice([1, 2, 7, 100, 1500])), int(rng.choice([1, 3, 64, 900]))
    m = int(rng.choice([1, 5, 200, 6000]))
    u = rng.integers(0, n_u, size=m)
    i = rng.integers(0, n_i, size=m) if rng.random() < 0.5 else np.minimum((rng.pareto(1.0, size=m) * 3).astype(np.int64), n_i - 1)
    if rng.random() < 0.5 and m > 3:                   # repeated train rows: a_rc = 2, 3, ... (dataset.py:132)
        rep = rng.integers(0, m, size=max(1, m // 5))
        u, i = np.concatenate([u, u[rep]]), np.concatenate([i, i[rep]])
    order = rng.permutation(len(u))
    return u[order].astype(np.int64), i[order].astype(np.int64), n_u + int(rng.integers(0, 3)), n_i + int(rng.integers(0, 3)), rng


@pytest.mark.parametrize('seed', SEEDS)
def test_graph_builder_random_pairs_vs_oracle(oracle, seed):
    from textgcn_amd.graph import NormGraph, train_mask_csr
    u, i, n_u, n_i, _ = _pairs(seed)
    gr = NormGraph.from_pairs(u, i, n_u, n_i)
    idx, val = gr.to_coo()
    oidx, oval = oracle.norm_coo(u, i, n_u, n_i)
    assert np.array_equal(idx, oidx) and np.array_equal(bits(val), bits(oval))
    assert np.array_equal(gr.rowptr, oracle.coo_to_csr(oidx, n_u + n_i))
    perm = gr.transpose_perm()                          # entry e = (r, c)  ->  position of (c, r)
    assert np.array_equal(idx[0][perm], idx[1]) and np.array_equal(idx[1][perm], idx[0])
    rp, items = train_mask_csr(u, i, n_u)
    seen_u = int(u.max()) + 1                            # (the oracle sizes its table by the largest user id present)
    orp, oitems = oracle.train_mask_csr(u, i, np.arange(seen_u))
    got = [items[rp[j]:rp[j + 1]].tolist() for j in range(n_u)]
    assert got[:seen_u] == [oitems[orp[j]:orp[j + 1]].tolist() for j in range(seen_u)] and not any(got[seen_u:])
    mrp, mitems = gr.train_mask()                        # the graph's own form: distinct items
    assert [sorted(set(g)) for g in got] == [mitems[mrp[j]:mrp[j + 1]].tolist() for j in range(n_u)]


@pytest.mark.parametrize('seed', SEEDS)
def test_metrics_random_lists_vs_oracle(oracle, seed):
    import torch
    from textgcn_amd.metrics import ranking_metrics, ranking_metrics_device, true_lists_csr
    rng = np.random.default_rng(50_000 + seed)
    n, n_items = int(rng.choice([1, 3, 50, 400])), int(rng.choice([5, 40, 3000]))
    ks = sorted(set(int(k) for k in rng.choice(np.arange(1, min(n_items, 60) + 1), size=int(rng.integers(1, 4)))))
    y_pred = np.stack([rng.permutation(n_items)[:max(ks)] for _ in range(n)])
    y_true = [rng.integers(0, n_items, size=int(rng.integers(1, 12))).tolist() for _ in range(n)]     # duplicates happen
    want = oracle.metrics(y_true, y_pred, ks)
    got = ranking_metrics(y_true, y_pred, ks)
    ptr, items = true_lists_csr(y_true)
    dev = ranking_metrics_device(torch.from_numpy(ptr), torch.from_numpy(items), torch.from_numpy(y_pred), ks, n_items)
    for m in ('recall', 'precision', 'hit', 'ndcg', 'f1'):
        assert np.allclose(got[m], want[m], atol=1e-12), m
        assert np.allclose(dev[m], want[m], atol=1e-9), m


@pytest.mark.parametrize('seed', range(12))
def test_row_plans_replayed_on_the_host(oracle, seed):
    """split_plan_arrays / segment_plan_arrays only regroup a row's entries: every entry appears once, in column order inside
    its chunk / piece, and the chunks' sums in order give the oracle's row to rounding."""
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph, split_plan_arrays
    rng = np.random.default_rng(60_000 + seed)
    n_u, n_i = int(rng.choice([30, 400])), int(rng.choice([5, 120]))
    u, i = synth.interactions(n_u, n_i, int(min(n_u * n_i // 2, rng.choice([200, 5000]))), seed=seed, zipf=1.2)
    gr = NormGraph.from_pairs(u, i, n_u, n_i)
    thr = int(rng.choice([4, 16, 64]))
    plan = split_plan_arrays(gr.rowptr, thr)
    deg = gr.degrees()
    if plan is None:
        assert deg.max() <= thr
        return
    x = rng.standard_normal((gr.n, 8)).astype(np.float32)
    ref = oracle.spmm_csr(gr.rowptr, gr.colidx, gr.vals, x)
    seen = np.zeros(gr.nnz, dtype=np.int32)
    for b, e in zip(plan['chunk_beg'], plan['chunk_end']):
        assert 0 < e - b <= thr
        seen[b:e] += 1
    long_rows = np.flatnonzero(deg > thr)
    covered = np.zeros(gr.nnz, dtype=bool)
    for r in long_rows:
        covered[gr.rowptr[r]:gr.rowptr[r + 1]] = True
    assert np.array_equal(seen.astype(bool), covered) and seen.max() == 1
    # replay: per-chunk fp32 chains, then the chunk sums added in order
    cb, ce = np.asarray(plan['chunk_beg']), np.asarray(plan['chunk_end'])
    for r in long_rows[:20]:
        sel = np.flatnonzero((cb >= gr.rowptr[r]) & (ce <= gr.rowptr[r + 1]))
        total = np.zeros(8, dtype=np.float32)
        for c in sel:
            part = np.zeros(8, dtype=np.float64)
            for e in range(cb[c], ce[c]):
                part = np.float32(part + np.float64(gr.vals[e]) * np.float64(x[gr.colidx[e]])).astype(np.float64)
            total = (total + part.astype(np.float32)).astype(np.float32)
        assert np.abs(total - ref[r]).max() <= 1e-5 * max(np.abs(ref[r]).max(), 1e-6) + 1e-7


def test_row_groups_cover_every_short_row_once():
    """graph.row_groups (tgcn_spmm_groups_f32's work list): every covered row at or below the threshold lies in exactly one group
    of 1..max_rows consecutive rows; cut rows and uncovered rows in none; longest-first order is a permutation of the same groups."""
    from textgcn_amd.graph import row_groups
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.integers(1, 300))
        lens = rng.integers(0, rng.choice([2, 5, 40, 200, 3000]), size=n)
        rp = np.concatenate([[0], np.cumsum(lens)])
        thr = rng.choice([None, 64, 1024])
        rows = None if rng.random() < 0.5 else np.sort(rng.choice(n, size=int(rng.integers(0, n + 1)), replace=False))
        mr = int(rng.choice([4, 8]))
        g = row_groups(rp, rows, thr, mr, 64)
        cov = np.zeros(n, dtype=int)
        for f, c in g:
            assert 1 <= c <= mr
            cov[f:f + c] += 1
        exp = np.ones(n, dtype=bool) if rows is None else np.isin(np.arange(n), rows)
        if thr is not None:
            exp &= lens <= thr
        assert np.array_equal(cov, exp.astype(int)), trial
        gl = row_groups(rp, rows, thr, mr, 64, longest_first=True)
        assert sorted(map(tuple, gl)) == sorted(map(tuple, g))
        ent = rp[gl[:, 0] + gl[:, 1]] - rp[gl[:, 0]] if len(gl) else np.zeros(0)
        assert np.all(np.diff(ent) <= 0)
