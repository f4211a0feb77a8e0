"""Pins the CPU oracle against golden vectors produced by RUNNING the reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import bits, normwise


def _coo(g, prefix=''):
    return g[prefix + 'norm_idx'], g[prefix + 'norm_val']


@pytest.mark.parametrize('name', ['g1_dummy', 'g2_synth60'])
def test_norm_matrix_bit_exact(golden, oracle, name):
    g = golden(name)
    idx, val = oracle.norm_coo(g['train_u'], g['train_i'], int(g['n_users']), int(g['n_items']))
    assert np.array_equal(idx, g['norm_idx'])
    assert np.array_equal(bits(val), bits(g['norm_val']))


@pytest.mark.parametrize('case', ['dup', 'star', 'rand'])
def test_norm_matrix_corner_cases(golden, oracle, case):
    g = golden('g6_builder')
    idx, val = oracle.norm_coo(g[f'{case}_train_u'], g[f'{case}_train_i'], int(g[f'{case}_n_users']), int(g[f'{case}_n_items']))
    assert np.array_equal(idx, g[f'{case}_norm_idx'])
    assert np.array_equal(bits(val), bits(g[f'{case}_norm_val']))


def test_norm_matrix_medium(golden, oracle):
    g = golden('g5_medium')
    idx, val = oracle.norm_coo(g['train_u'], g['train_i'], int(g['n_users']), int(g['n_items']))
    assert int(idx[0].sum()) == int(g['norm_row_sum'])
    assert int((idx[1] * (np.arange(idx.shape[1]) % 1009)).sum()) == int(g['norm_col_weighted'])
    assert np.array_equal(bits(val), bits(g['norm_val']))


def test_dummy_forward_bit_exact(golden, oracle):
    g = golden('g1_dummy')
    idx, val = _coo(g)
    e0 = np.concatenate([g['emb_user'], g['emb_item']])
    out, layers = oracle.propagate(idx, val, e0, 3)
    for k in range(4):
        assert np.array_equal(bits(layers[k]), bits(g[f'layer{k}'])), f'layer {k}'
    assert np.array_equal(bits(out), bits(np.concatenate([g['users_emb'], g['items_emb']])))


@pytest.mark.parametrize('variant', ['a', 'single', 'k4d128', 'd48'])
def test_synth60_forward_scores_topk(golden, oracle, variant):
    g = golden('g2_synth60')
    idx, val = _coo(g)
    K = int(g[f'{variant}_n_layers'])
    out, layers = oracle.propagate(idx, val, g[f'{variant}_layer0'], K, single=(variant == 'single'))
    for k in range(K + 1):
        assert np.array_equal(bits(layers[k]), bits(g[f'{variant}_layer{k}']))
    n_u = int(g['n_users'])
    assert np.array_equal(bits(out[:n_u]), bits(g[f'{variant}_users_emb']))
    assert np.array_equal(bits(out[n_u:]), bits(g[f'{variant}_items_emb']))
    s = oracle.score_dense(out[:n_u], out[n_u:])
    # BLAS order is unspecified in general -> normwise bar (it happens to be bit-equal at this size)
    assert normwise(s, g[f'{variant}_rating']) <= 1e-6
    mr, mi = oracle.train_mask_csr(g['train_u'], g['train_i'], np.arange(n_u))
    oracle.mask_train(s, mr, mi)
    assert np.array_equal(np.isneginf(s), np.isneginf(g[f'{variant}_masked']))
    v, i = oracle.topk(s, 10, round4=True)
    # every user here has >= 10 unmasked items with distinct scores -> fully determined order
    assert np.array_equal(i, g[f'{variant}_topk_idx'])
    assert np.array_equal(bits(v), bits(g[f'{variant}_topk_val']))


def test_dummy_topk_finite_prefix(golden, oracle):
    """-inf ties are implementation-defined in torch.topk: compare the finite prefix and the
    set of the rest."""
    g = golden('g1_dummy')
    out = np.concatenate([g['users_emb'], g['items_emb']])
    s = oracle.score_dense(out[:5], out[5:])
    assert np.array_equal(bits(s), bits(g['rating']))
    mr, mi = oracle.train_mask_csr(g['train_u'], g['train_i'], np.arange(5))
    oracle.mask_train(s, mr, mi)
    assert np.array_equal(bits(s), bits(g['masked']))
    v, i = oracle.topk(s, 3, round4=True)
    for b in range(5):
        fin = np.isfinite(g['topk_val'][b])
        assert np.array_equal(i[b][fin], g['topk_idx'][b][fin])
        assert np.array_equal(bits(v[b][fin]), bits(g['topk_val'][b][fin]))
        # which of the tied -inf (masked) items fill the tail is arbitrary in torch: they must be train items
        masked = set(np.nonzero(np.isneginf(g['masked'][b]))[0])
        assert set(i[b][~fin]) <= masked and set(g['topk_idx'][b][~fin]) <= masked
        assert np.all(np.isneginf(v[b][~fin]))


def test_dropout_matrix_and_forward(golden, oracle):
    g2, g3 = golden('g2_synth60'), golden('g3_dropout')
    idx, val = _coo(g2)
    di, dv = oracle.dropout_coo(idx, val, g3['rand'], float(g3['p']))
    assert np.array_equal(di, g3['drop_idx'])
    assert np.array_equal(bits(dv), bits(g3['drop_val']))
    out, _ = oracle.propagate(di, dv, g2['a_layer0'], 3)
    n_u = int(g2['n_users'])
    assert np.array_equal(bits(out[:n_u]), bits(g3['users_emb']))
    assert np.array_equal(bits(out[n_u:]), bits(g3['items_emb']))


@pytest.mark.parametrize('d,K', [(64, 3), (128, 4)])
def test_medium_forward_rows_and_checksum(golden, oracle, d, K):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    from golden_inputs import exact_embedding
    g = golden('g5_medium')
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    idx, val = oracle.norm_coo(g['train_u'], g['train_i'], n_u, n_i)
    e0 = np.concatenate([exact_embedding(n_u, d, 21), exact_embedding(n_i, d, 22)])
    out, layers = oracle.propagate(idx, val, e0, K)
    p = f'd{d}_'
    rows = g[p + 'rows']
    assert np.array_equal(bits(out[rows]), bits(g[p + 'repr_rows']))
    assert np.array_equal(bits(layers[-1][rows]), bits(g[p + 'lastlayer_rows']))
    assert int(bits(out).astype(np.uint64).sum()) == int(g[p + 'repr_bits_sum'])
    assert int(bits(layers[-1]).astype(np.uint64).sum()) == int(g[p + 'lastlayer_bits_sum'])
    users = g[p + 'pred_users']
    s = oracle.score_dense(out[users], out[n_u:])
    mr, mi = oracle.train_mask_csr(g['train_u'], g['train_i'], users)
    oracle.mask_train(s, mr, mi)
    v, i = oracle.topk(s, 40, round4=True)
    # torch's BLAS matmul may differ in the last bits at this size: compare where the reference's own
    # ordering is decided by a margin
    ref_v, ref_i = g[p + 'topk_val'], g[p + 'topk_idx']
    assert normwise(v, ref_v) <= 1e-3  # values are rounded to 1e-4 of ~0.05 magnitudes
    agree = (i == ref_i).mean()
    assert agree > 0.99, agree


def test_metrics_match_reference(golden, oracle):
    g = golden('g1_dummy')
    test_u, test_i = g['test_u'], g['test_i']
    users = g['test_users']
    y_true = [test_i[test_u == u].tolist() for u in users]
    res = oracle.metrics(y_true, g['topk_idx'][users], g['k'].tolist())
    for m in ('recall', 'precision', 'hit', 'ndcg', 'f1'):
        assert np.allclose(res[m], g[f'metric_{m}'], atol=1e-12), m
