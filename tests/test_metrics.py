"""Ranking metrics: the host form, the oracle and the device form against utils.calculate_metrics of the reference (golden
G9: duplicate test rows inside y_true, lists of different lengths, users without a hit)."""
import numpy as np
import pytest

METRICS = ('recall', 'precision', 'hit', 'ndcg', 'f1')


def _lists(g):
    ptr = g['true_ptr']
    return [g['true_items'][ptr[j]:ptr[j + 1]].tolist() for j in range(len(ptr) - 1)]


def test_host_metrics_match_reference_with_duplicate_test_rows(golden):
    from textgcn_amd.metrics import ranking_metrics, true_lists_csr
    g = golden('g9_metrics')
    y_true = _lists(g)
    assert any(len(t) != len(set(t)) for t in y_true)            # the fixture does hold duplicates
    res = ranking_metrics(y_true, g['y_pred'], g['ks'].tolist())
    for m in METRICS:
        assert np.allclose(res[m], g[f'metric_{m}'], atol=1e-12), m
    ptr, items = true_lists_csr(y_true)
    assert np.array_equal(ptr, g['true_ptr']) and np.array_equal(items, g['true_items'])


def test_oracle_metrics_match_reference(golden, oracle):
    g = golden('g9_metrics')
    res = oracle.metrics(_lists(g), g['y_pred'], g['ks'].tolist())
    for m in METRICS:
        assert np.allclose(res[m], g[f'metric_{m}'], atol=1e-12), m


def test_device_form_on_cpu_tensors_matches_reference(golden):
    """the device form is plain torch ops: run here on CPU tensors against the same golden"""
    import torch
    from textgcn_amd.metrics import ranking_metrics_device
    g = golden('g9_metrics')
    res = ranking_metrics_device(torch.from_numpy(g['true_ptr']), torch.from_numpy(g['true_items']), torch.from_numpy(g['y_pred']),
                                 g['ks'].tolist(), 60)
    for m in METRICS:
        assert np.allclose(res[m], g[f'metric_{m}'], atol=1e-12), m


@pytest.mark.gpu
def test_device_metrics_match_host_form(cuda, golden):
    import torch
    from textgcn_amd.metrics import ranking_metrics, ranking_metrics_device, true_lists_csr
    g = golden('g9_metrics')
    res = ranking_metrics_device(torch.from_numpy(g['true_ptr']).to(cuda), torch.from_numpy(g['true_items']).to(cuda),
                                 torch.from_numpy(g['y_pred']).to(cuda), g['ks'].tolist(), 60)
    for m in METRICS:
        assert np.allclose(res[m], g[f'metric_{m}'], atol=1e-12), m
    rng = np.random.default_rng(1)
    n, n_items, kmax = 5000, 3000, 40
    y_true = [rng.integers(0, n_items, rng.integers(1, 30)).tolist() for _ in range(n)]
    pred = np.stack([rng.permutation(n_items)[:kmax] for _ in range(n)])
    want = ranking_metrics(y_true, pred, [20, 40])
    ptr, items = true_lists_csr(y_true)
    got = ranking_metrics_device(torch.from_numpy(ptr).to(cuda), torch.from_numpy(items).to(cuda), torch.from_numpy(pred).to(cuda),
                                 [20, 40], n_items)
    for m in METRICS:
        assert np.allclose(got[m], want[m], atol=1e-12), m
