"""ltr_linear (BASELINE config 5): the oracle's LTR restatement against the reference's own outputs (CPU), and
the folded-GEMM HIP path against both (GPU)."""
import types

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import bits, normwise


def test_oracle_ltr_matches_reference(golden, oracle):
    g = golden('g4_ltr')
    n_u = int(g['n_users'])
    users = np.arange(n_u)
    s = oracle.ltr_score(g['users_emb'], g['users_as_avg_reviews'][users], g['users_as_avg_desc'][users], g['items_emb'],
                         g['items_as_avg_reviews'], g['items_as_desc'], g['w'], float(g['b'][0]))
    assert normwise(s, g['scores']) <= 1e-6          # BLAS / addmv order unspecified -> normwise
    f0 = oracle.score_dense(g['users_emb'], g['items_emb'])
    assert normwise(f0, g['features'][:, :, 0]) <= 1e-6
    f3 = oracle.score_dense(g['users_as_avg_reviews'], g['items_as_desc'])
    assert normwise(f3, g['features'][:, :, 3]) <= 1e-6
    p = oracle.ltr_pairwise(g['users_emb'][g['pairs_u']], g['users_as_avg_reviews'][g['pairs_u']], g['users_as_avg_desc'][g['pairs_u']],
                            g['items_emb'][g['pairs_i']], g['items_as_avg_reviews'][g['pairs_i']], g['items_as_desc'][g['pairs_i']],
                            g['w'], float(g['b'][0]))
    assert normwise(p, g['pair_scores'].reshape(-1)) <= 1e-6
    assert list(g['state_keys']) == ['embedding_item.weight', 'embedding_user.weight', 'layers.0.bias', 'layers.0.weight']


def test_oracle_ltr_pop_matches_reference(golden, oracle):
    """G7: ltr_pop scores of the reference (LTRLinearWPop) against the oracle's restatement."""
    g4, g = golden('g4_ltr'), golden('g7_ltr_pop')
    s = oracle.ltr_pop_score(g4['users_emb'], g4['users_as_avg_reviews'], g4['users_as_avg_desc'], g4['items_emb'],
                             g4['items_as_avg_reviews'], g4['items_as_desc'], g['popularity_users'], g['popularity_items'],
                             g['w'], float(g['b'][0]))
    assert normwise(s, g['scores']) <= 1e-6
    assert list(g['state_keys']) == ['embedding_item.weight', 'embedding_user.weight', 'layers.0.bias', 'layers.0.weight']
    assert list(g['feature_names'][-2:]) == ['user popularity', 'item popularity']


def _dataset(g):
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    from textgcn_amd.graph import NormGraph
    train = pd.DataFrame({'user_id': g['train_u'], 'asin': g['train_i']})
    test = pd.DataFrame({'user_id': g['test_u'], 'asin': g['test_i']})
    return types.SimpleNamespace(
        n_users=n_u, n_items=n_i, graph=NormGraph.from_pairs(g['train_u'], g['train_i'], n_u, n_i), norm_matrix=None,
        true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
        train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
        user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
        item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}), all_items=range(n_i),
        items_as_desc=torch.from_numpy(g['items_as_desc']), items_as_avg_reviews=torch.from_numpy(g['items_as_avg_reviews']),
        users_as_avg_reviews=torch.from_numpy(g['users_as_avg_reviews']), users_as_avg_desc=torch.from_numpy(g['users_as_avg_desc']))


@pytest.mark.gpu
def test_hip_ltr_matches_reference(golden, cuda, oracle, tmp_path):
    from golden_inputs import exact_embedding
    from textgcn_amd.ltr import LTRLinear
    g = golden('g4_ltr')
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    p = types.SimpleNamespace(k=[5, 10], emb_size=64, n_layers=3, device='cuda:0', load=None, load_base=None, freeze=True,
                              batch_size=32, quiet=True, exact=True, ltr_layers=[], save_path=str(tmp_path))
    m = LTRLinear(p, _dataset(g))
    with torch.no_grad():
        m.embedding_user.weight.copy_(torch.from_numpy(exact_embedding(n_u, 64, 11)))
        m.embedding_item.weight.copy_(torch.from_numpy(exact_embedding(n_i, 64, 12)))
        m.layers[0].weight.copy_(torch.from_numpy(g['w']))
        m.layers[0].bias.copy_(torch.from_numpy(g['b']))
    assert sorted(m.state_dict().keys()) == list(g['state_keys'])
    assert not m.embedding_user.weight.requires_grad
    with torch.no_grad():
        ue, ie = m.representation
        assert np.array_equal(bits(ue.cpu().numpy()), bits(g['users_emb']))
        users = torch.arange(n_u, device=cuda)
        s = m.score_batchwise(ue[users], ie, users)
        assert normwise(s.cpu().numpy(), g['scores']) <= 1e-5                # fp32 bar of the north star is 1e-4
        pu, pi = torch.from_numpy(g['pairs_u']).to(cuda), torch.from_numpy(g['pairs_i']).to(cuda)
        ps = m.score_pairwise(ue[pu], ie[pi], pu, pi)
        assert ps.shape == (200, 1) and normwise(ps.cpu().numpy(), g['pair_scores']) <= 1e-5
    pred, sc = m.predict(np.arange(n_u), with_scores=True)
    pred, sc = np.asarray(pred), np.asarray(sc, dtype=np.float32)
    # ranked top-10 identical wherever the reference's adjacent scores differ by more than the fp32 noise
    ref_i, ref_v = g['topk_idx'], g['topk_val']
    gaps_ok = np.abs(np.diff(ref_v, axis=1)) > 2e-4
    row_ok = gaps_ok.all(axis=1)
    assert row_ok.mean() > 0.8
    assert np.array_equal(pred[row_ok], ref_i[row_ok])
    assert np.abs(sc - ref_v).max() <= 1.01e-4                                 # both rounded to 4 decimals
    res = m.evaluate()
    for met in ('recall', 'precision', 'hit', 'ndcg', 'f1'):
        assert np.allclose(res[met], g[f'metric_{met}'], atol=0.02), met


@pytest.mark.gpu
def test_hip_ltr_hidden_layers_collapse_to_affine(golden, cuda, tmp_path):
    """--ltr_layers 4 3: still one affine map (no activation in the reference's Sequential)."""
    from textgcn_amd.ltr import LTRLinear
    g = golden('g4_ltr')
    p = types.SimpleNamespace(k=[5], emb_size=64, n_layers=2, device='cuda:0', load=None, load_base=None, freeze=True,
                              batch_size=64, quiet=True, ltr_layers=[4, 3], save_path=str(tmp_path))
    m = LTRLinear(p, _dataset(g))
    with torch.no_grad():
        ue, ie = m.representation
        users = torch.arange(int(g['n_users']), device=cuda)
        s = m.score_batchwise(ue[users], ie, users)
        # torch restatement of the reference's formula with the full layer stack
        uv = {'emb': ue, 'reviews': m.users_as_avg_reviews, 'desc': m.users_as_avg_desc}
        iv = {'emb': ie, 'reviews': m.items_as_avg_reviews, 'desc': m.items_as_desc}
        feats = torch.stack([uv['emb'] @ iv['emb'].T, uv['reviews'] @ iv['reviews'].T, uv['desc'] @ iv['desc'].T,
                             uv['reviews'] @ iv['desc'].T, uv['desc'] @ iv['reviews'].T], dim=-1)
        ref = m.layers(feats).squeeze(-1)
    assert normwise(s.cpu().numpy(), ref.cpu().numpy()) <= 1e-5


@pytest.mark.gpu
def test_hip_ltr_pop_matches_reference(golden, cuda, tmp_path):
    """ltr_pop (LTRLinearWPop, ltr_models.py:213-241) against the reference's own outputs (G7): the two popularity
    features ride in two more columns of the folded GEMM."""
    from golden_inputs import exact_embedding
    from textgcn_amd.ltr import LTRLinearWPop
    from textgcn_amd.model import get_class
    g4, g = golden('g4_ltr'), golden('g7_ltr_pop')
    assert get_class('ltr_pop')[1] is LTRLinearWPop
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    ds = _dataset(g4)
    ds.popularity_users, ds.popularity_items = torch.from_numpy(g['popularity_users']), torch.from_numpy(g['popularity_items'])
    p = types.SimpleNamespace(k=[5, 10], emb_size=64, n_layers=3, device='cuda:0', load=None, load_base=None, freeze=True,
                              batch_size=32, quiet=True, exact=True, ltr_layers=[], save_path=str(tmp_path))
    m = LTRLinearWPop(p, ds)
    assert m.feature_names == list(g['feature_names']) and m.layers[0].weight.shape == (1, 7)
    with torch.no_grad():
        m.embedding_user.weight.copy_(torch.from_numpy(exact_embedding(n_u, 64, 11)))
        m.embedding_item.weight.copy_(torch.from_numpy(exact_embedding(n_i, 64, 12)))
        m.layers[0].weight.copy_(torch.from_numpy(g['w']))
        m.layers[0].bias.copy_(torch.from_numpy(g['b']))
    assert sorted(m.state_dict().keys()) == list(g['state_keys'])
    with torch.no_grad():
        ue, ie = m.representation
        users = torch.arange(n_u, device=cuda)
        s = m.score_batchwise(ue[users], ie, users)
        assert normwise(s.cpu().numpy(), g['scores']) <= 1e-5
        assert normwise(s.cpu().numpy(), g4['scores']) > 1e-2                 # the popularity terms really count
        pu, pi = torch.from_numpy(g['pairs_u']).to(cuda), torch.from_numpy(g['pairs_i']).to(cuda)
        ps = m.score_pairwise(ue[pu], ie[pi], pu, pi)
        assert ps.shape == (200, 1) and normwise(ps.cpu().numpy(), g['pair_scores']) <= 1e-5
    pred, sc = m.predict(np.arange(n_u), with_scores=True)
    pred, sc = np.asarray(pred), np.asarray(sc, dtype=np.float32)
    ref_i, ref_v = g['topk_idx'], g['topk_val']
    row_ok = (np.abs(np.diff(ref_v, axis=1)) > 2e-4).all(axis=1)
    assert row_ok.mean() > 0.8
    assert np.array_equal(pred[row_ok], ref_i[row_ok])
    assert np.abs(sc - ref_v).max() <= 1.01e-4


@pytest.mark.gpu
def test_hip_ltr_scores_follow_in_place_embedding_updates(golden, cuda, tmp_path):
    """score_batchwise_ltr after the embedding tables changed IN PLACE (an unfrozen training step, load_state_dict): the
    item operand of the folded GEMM must be rebuilt -- the propagated tables are rewritten by HIP kernels at addresses the
    allocator reuses, so no pointer/version key can vouch for a cached copy.  Checked against the unfolded pairwise form."""
    from textgcn_amd.ltr import LTRLinear
    g = golden('g4_ltr')
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    p = types.SimpleNamespace(k=[5, 10], emb_size=64, n_layers=3, device='cuda:0', load=None, load_base=None, freeze=False,
                              batch_size=32, quiet=True, exact=True, ltr_layers=[], save_path=str(tmp_path))
    m = LTRLinear(p, _dataset(g))
    users = torch.arange(n_u, device=cuda)
    uu, ii = torch.meshgrid(users, torch.arange(n_i, device=cuda), indexing='ij')
    for step in range(3):
        with torch.no_grad():
            ue, ie = m.representation
            s = m.score_batchwise(ue[users], ie, users)
            ref = m.score_pairwise(ue[uu.reshape(-1)], ie[ii.reshape(-1)], uu.reshape(-1), ii.reshape(-1)).reshape(n_u, n_i)
            assert normwise(s.cpu().numpy(), ref.cpu().numpy()) <= 1e-5, step
            m.embedding_item.weight.mul_(1.5).add_(0.01 * (step + 1))      # in place: same storage, new values
            m.embedding_user.weight.add_(0.02)


@pytest.mark.gpu
@pytest.mark.parametrize('cls_name', ['LTRLinear', 'LightGCN'])
def test_predict_host_syncs_do_not_grow_with_chunks(golden, cuda, tmp_path, cls_name):
    """a predict call reads the head's weights once (LTR) and copies the user ids once; no chunk adds a
    host synchronisation (torch's sync debug mode counts every synchronising call), and non-contiguous user lists give the same
    rows as contiguous ones."""
    import warnings
    from textgcn_amd.ltr import LTRLinear
    from textgcn_amd.model import LightGCN
    g = golden('g4_ltr')
    n_u = int(g['n_users'])
    p = types.SimpleNamespace(k=[5, 10], emb_size=64, n_layers=3, device='cuda:0', load=None, load_base=None, freeze=True,
                              batch_size=8, quiet=True, ltr_layers=[], save_path=str(tmp_path))
    m = (LTRLinear if cls_name == 'LTRLinear' else LightGCN)(p, _dataset(g))
    users = np.arange(n_u, dtype=np.int64)

    def syncs(chunk):
        m.ltr_predict_chunk = m.predict_chunk = chunk
        m.predict_tensors(users)              # warm: streams, workspaces, plans
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode('warn')
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter('always')
                out = m.predict_tensors(users)
        finally:
            torch.cuda.set_sync_debug_mode('default')
        torch.cuda.synchronize()
        return len([x for x in w if 'synchroniz' in str(x.message).lower()]), out
    one, ref = syncs(1 << 20)                 # every user in one chunk
    many, got = syncs(8)                      # n_u / 8 chunks
    assert many == one, (one, many)
    assert torch.equal(ref[1], got[1]) and torch.equal(ref[0], got[0])
    perm = np.random.default_rng(0).permutation(n_u)
    pv, pi = m.predict_tensors(perm)          # a non-contiguous list: device gather of the mask rows
    assert torch.equal(pi, ref[1][perm]) and torch.equal(pv, ref[0][perm])


@pytest.mark.gpu
def test_pairwise_features_kernel_and_its_gradient(golden, cuda, tmp_path):
    """score_pairwise_ltr on the native feature kernel against the reference's composition (get_features_pairwise,
    ltr_models.py:148-166, restated with torch ops): values, the gradient into the head and -- unfrozen base model -- into the
    gathered embedding rows; int32 / host ids are taken, an id outside its table raises IndexError."""
    from textgcn_amd.ltr import LTRLinear
    g = golden('g4_ltr')
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    p = types.SimpleNamespace(k=[5, 10], emb_size=64, n_layers=3, device='cuda:0', load=None, load_base=None, freeze=False,
                              batch_size=32, quiet=True, ltr_layers=[7], save_path=str(tmp_path))
    m = LTRLinear(p, _dataset(g))
    rng = np.random.default_rng(0)
    pu = torch.from_numpy(rng.integers(0, n_u, 500)).to(cuda)
    pi = torch.from_numpy(rng.integers(0, n_i, 500)).to(cuda)
    res = []
    for native in (True, False):
        ue = torch.randn(n_u, 64, device=cuda, generator=torch.Generator(cuda).manual_seed(1)).requires_grad_()
        ie = torch.randn(n_i, 64, device=cuda, generator=torch.Generator(cuda).manual_seed(2)).requires_grad_()
        m.zero_grad()
        if native:
            s = m.score_pairwise(ue[pu], ie[pi], pu.to(torch.int32), pi.cpu())
        else:
            dot = lambda a, c: (a * c).sum(dim=1, keepdim=True)  # noqa: E731
            ru, du, ri, di = m.users_as_avg_reviews[pu], m.users_as_avg_desc[pu], m.items_as_avg_reviews[pi], m.items_as_desc[pi]
            s = m.layers(torch.cat([dot(ue[pu], ie[pi]), dot(ru, ri), dot(du, di), dot(ru, di), dot(du, ri)], dim=1))
        assert s.shape == (500, 1)
        (s * torch.linspace(-1, 1, 500, device=cuda)[:, None]).sum().backward()
        res.append([s.detach().cpu().numpy(), ue.grad.cpu().numpy(), ie.grad.cpu().numpy(), m.layers[0].weight.grad.cpu().numpy().copy()])
    for a, b in zip(*res):
        assert normwise(a, b) <= 1e-5
    with pytest.raises(IndexError):
        m.score_pairwise(torch.zeros(2, 64, device=cuda), torch.zeros(2, 64, device=cuda), torch.tensor([0, n_u]), torch.tensor([0, 1]))
    with pytest.raises(IndexError):      # device ids outside a training epoch: the range flag is read at once (round-3 advice)
        m.score_pairwise(torch.zeros(2, 64, device=cuda), torch.zeros(2, 64, device=cuda), torch.tensor([0, 1], device=cuda),
                         torch.tensor([0, -1], device=cuda))
