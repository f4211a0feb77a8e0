"""Host-side pieces of the drop-in surface (no GPU): id mapping, metrics, sampler, registry, state_dict keys."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_interaction_data_matches_reference_ids(golden):
    from textgcn_amd.interactions import InteractionData
    g = golden('g1_dummy')
    ds = InteractionData(folder=os.path.join(GOLDEN, 'dummy'), k=[1, 2, 3])
    assert (ds.n_users, ds.n_items, ds.n_train, ds.n_test) == (5, 4, 13, 3)
    assert np.array_equal(ds.train_df.user_id.values, g['train_u']) and np.array_equal(ds.train_df.asin.values, g['train_i'])
    assert np.array_equal(ds.test_df.user_id.values, g['test_u']) and np.array_equal(ds.test_df.asin.values, g['test_i'])
    assert list(ds.user_mapping.org_id) == list(g['user_org']) and list(ds.item_mapping.org_id) == list(g['item_org'])
    idx = ds.norm_matrix.indices().numpy()
    assert np.array_equal(idx, g['norm_idx'])
    assert np.array_equal(ds.norm_matrix.values().numpy().view(np.uint32), g['norm_val'].view(np.uint32))


def test_interaction_data_rejects_k_too_large():
    from textgcn_amd.interactions import InteractionData
    with pytest.raises(AssertionError):
        InteractionData(folder=os.path.join(GOLDEN, 'dummy'), k=[20, 40])    # dataset.py:25 / SURVEY.md F5


def test_sampler_fails_loudly_where_reference_hangs():
    """users 0, 2, 3 of data/dummy have a single non-positive item but need 2 negatives."""
    from textgcn_amd.interactions import InteractionData
    ds = InteractionData(folder=os.path.join(GOLDEN, 'dummy'), k=[1])
    assert ds.bucket_len == 2
    with pytest.raises(ValueError):
        ds[0]


def test_sampler_contract(tmp_path):
    from textgcn_amd.interactions import InteractionData
    rng = np.random.default_rng(0)
    with open(tmp_path / 'train.tsv', 'w') as f:
        f.write('user_id\tasin\n')
        for u in range(30):
            for i in rng.choice(50, size=6, replace=False):
                f.write(f'u{u:02d}\ti{i:02d}\n')
    with open(tmp_path / 'test.tsv', 'w') as f:
        f.write('user_id\tasin\nu00\ti00\n')
    ds = InteractionData(folder=str(tmp_path), k=[5], neg_samples=2)
    assert len(ds) == ds.bucket_len * ds.n_users == 180
    rows = torch.stack([ds[j] for j in range(len(ds))]).numpy()
    assert rows.shape == (180, 4)
    pos_sets = [set(v) for v in ds.train_user_dict]
    for u, p, n1, n2 in rows:
        assert p in pos_sets[u] and n1 not in pos_sets[u] and n2 not in pos_sets[u]
    for u in range(ds.n_users):   # a user's negatives within an epoch are distinct (dataset.py:172-177)
        negs = rows[rows[:, 0] == u][:, 2:].ravel()
        assert len(set(negs)) == len(negs)


def test_metrics_match_reference_values(golden):
    """recall/precision/hit/ndcg/f1 @5,@10 of the reference's own evaluate() on its own predictions (G2)."""
    from textgcn_amd.metrics import ranking_metrics
    g = golden('g2_synth60')
    users = g['test_users']
    y_true = [g['test_i'][g['test_u'] == u].tolist() for u in users]
    for name in ('a', 'single', 'k4d128'):
        res = ranking_metrics(y_true, g[f'{name}_topk_idx'][users], [5, 10])
        for m in ('recall', 'precision', 'hit', 'ndcg', 'f1'):
            assert np.allclose(res[m], g[f'{name}_metric_{m}'], atol=1e-12), (name, m)


def test_early_stop_rule():
    from textgcn_amd.metrics import early_stop
    mk = lambda rows: {m: np.array(rows) for m in ('recall', 'precision')}   # noqa: E731
    assert not early_stop(mk([[.1], [.2]]))
    assert early_stop(mk([[.3], [.2], [.1]]))
    assert early_stop(mk([[.30001], [.30002], [.30003]]))
    assert not early_stop(mk([[.1], [.3], [.2]]))


def test_model_surface_and_state_dict_keys():
    from textgcn_amd.interactions import InteractionData
    from textgcn_amd.model import LightGCN, get_class
    ds = InteractionData(folder=os.path.join(GOLDEN, 'dummy'), k=[1, 2, 3])
    p = types.SimpleNamespace(k=[1, 2, 3], emb_size=64, n_layers=3, device='cpu', load=None)
    m = LightGCN(p, ds)
    assert list(m.state_dict().keys()) == ['embedding_user.weight', 'embedding_item.weight']
    for name in ('representation', 'layer_aggregation', 'layer_combination', 'score_pairwise', 'score_batchwise', 'predict',
                 'evaluate', 'fit', 'get_loss', 'bpr_loss', 'reg_loss', 'load_model', 'checkpoint', 'embedding_matrix'):
        assert hasattr(LightGCN, name), name
    assert m.training is False and m.k == [1, 2, 3] and m.batch_size == 2048
    assert get_class('lgcn')[1] is LightGCN
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m.representation        # no CPU fallback
