"""End-to-end drop-in checks on the GPU: the LightGCN class against the reference's own outputs (golden G1-G3)
and training-path gradients against a torch fp64 restatement."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bits, normwise

pytestmark = pytest.mark.gpu


def _params(**kw):
    base = dict(k=[1, 2, 3], emb_size=64, n_layers=3, device='cuda:0', load=None, batch_size=2048, quiet=True, save=False,
                dropout=0.4, single=False, exact=True, lr=0.001, epochs=1, reg_lambda=1e-4, evaluate_every=1, neg_samples=1,
                save_path='.', uid='t')
    base.update(kw)
    return types.SimpleNamespace(**base)


def _set_weights(model, eu, ei):
    with torch.no_grad():
        model.embedding_user.weight.copy_(torch.from_numpy(eu))
        model.embedding_item.weight.copy_(torch.from_numpy(ei))


def test_dummy_end_to_end_matches_reference(golden, cuda, tmp_path):
    """BASELINE config 1: data/dummy, lgcn, d=64, K=3: representation bit-identical, ranked top-k identical on
    the finite prefix, metrics identical, predictions.tsv identical where the reference's order is defined."""
    from textgcn_amd.interactions import InteractionData
    from textgcn_amd.model import LightGCN
    g = golden('g1_dummy')
    ds = InteractionData(folder=os.path.join(GOLDEN, 'dummy'), k=[1, 2, 3])
    m = LightGCN(_params(save_path=str(tmp_path)), ds)
    _set_weights(m, g['emb_user'], g['emb_item'])
    with torch.no_grad():
        ue, ie = m.representation
    assert np.array_equal(bits(ue.cpu().numpy()), bits(g['users_emb']))
    assert np.array_equal(bits(ie.cpu().numpy()), bits(g['items_emb']))
    pred, scores = m.predict(range(ds.n_users), with_scores=True, save=True)
    pred, scores = np.asarray(pred), np.asarray(scores, dtype=np.float32)
    for b in range(5):
        fin = np.isfinite(g['topk_val'][b])
        assert np.array_equal(pred[b][fin], g['topk_idx'][b][fin])
        assert np.array_equal(bits(scores[b][fin]), bits(g['topk_val'][b][fin]))
    res = m.evaluate()
    for name in ('recall', 'precision', 'hit', 'ndcg', 'f1'):
        assert np.allclose(res[name], g[f'metric_{name}'], atol=1e-12)
    ours = open(os.path.join(str(tmp_path), 'predictions.tsv')).read().splitlines()
    ref = bytes(g['predictions_tsv']).decode().splitlines()
    assert ours[0] == ref[0] and len(ours) == len(ref)
    for lo, lr in zip(ours[1:], ref[1:]):
        uo, po, so = lo.split('\t')
        ur, pr, sr = lr.split('\t')
        assert uo == ur
        n_fin = sum(1 for x in eval(sr, {'inf': float('inf')}) if np.isfinite(x))
        assert eval(po)[:n_fin] == eval(pr)[:n_fin]
        assert eval(so, {'inf': float('inf')})[:n_fin] == eval(sr, {'inf': float('inf')})[:n_fin]


@pytest.mark.parametrize('name,extra', [('a', {}), ('single', {'single': True}), ('k4d128', {'emb_size': 128, 'n_layers': 4}),
                                        ('d48', {'emb_size': 48, 'n_layers': 2})])
def test_synth60_model_vs_reference(golden, cuda, tmp_path, name, extra):
    """G2 through the model class built from a reference-style dataset object (norm_matrix COO + pandas dicts)."""
    import pandas as pd
    from textgcn_amd.model import LightGCN
    g = golden('g2_synth60')
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    train = pd.DataFrame({'user_id': g['train_u'], 'asin': g['train_i']})
    test = pd.DataFrame({'user_id': g['test_u'], 'asin': g['test_i']})
    ds = types.SimpleNamespace(   # what base_model.py:54-62 reads from a reference BaseDataset
        n_users=n_u, n_items=n_i,
        norm_matrix=torch.sparse_coo_tensor(torch.from_numpy(g['norm_idx']), torch.from_numpy(g['norm_val']), (n_u + n_i,) * 2).coalesce(),
        true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
        train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
        user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': [f'u{x}' for x in range(n_u)]}),
        item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': [f'i{x}' for x in range(n_i)]}))
    m = LightGCN(_params(k=[5, 10], batch_size=32, save_path=str(tmp_path), **extra), ds)
    d = int(g[f'{name}_d'])
    _set_weights(m, g[f'{name}_layer0'][:n_u], g[f'{name}_layer0'][n_u:])
    with torch.no_grad():
        ue, ie = m.representation
    assert np.array_equal(bits(ue.cpu().numpy()), bits(g[f'{name}_users_emb']))
    assert np.array_equal(bits(ie.cpu().numpy()), bits(g[f'{name}_items_emb']))
    pred, scores = m.predict(np.arange(n_u), with_scores=True)
    assert np.array_equal(np.asarray(pred), g[f'{name}_topk_idx'])
    assert np.array_equal(bits(np.asarray(scores, dtype=np.float32)), bits(g[f'{name}_topk_val']))
    res = m.evaluate()
    for met in ('recall', 'precision', 'hit', 'ndcg', 'f1'):
        assert np.allclose(res[met], g[f'{name}_metric_{met}'], atol=1e-12), met
    assert d == m.emb_size


def test_training_mode_dropout_forward_matches_reference(golden, cuda):
    """G3: same torch seed -> same dropped edges -> same training-mode representation as the reference."""
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.model import LightGCN
    import pandas as pd
    g, g3 = golden('g2_synth60'), golden('g3_dropout')
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    train = pd.DataFrame({'user_id': g['train_u'], 'asin': g['train_i']})
    test = pd.DataFrame({'user_id': g['test_u'], 'asin': g['test_i']})
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=NormGraph.from_pairs(g['train_u'], g['train_i'], n_u, n_i),
                               true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
                               train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
                               user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
                               item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}), norm_matrix=None)
    m = LightGCN(_params(k=[5], dropout=float(g3['p']), dropout_rng='cpu'), ds)
    _set_weights(m, g['a_layer0'][:n_u], g['a_layer0'][n_u:])
    m.training = True
    torch.manual_seed(123)
    with torch.no_grad():
        ue, ie = m.representation
    assert np.array_equal(bits(ue.cpu().numpy()), bits(g3['users_emb']))
    assert np.array_equal(bits(ie.cpu().numpy()), bits(g3['items_emb']))


@pytest.mark.parametrize('single', [False, True])
def test_backward_matches_fp64_autograd(golden, cuda, single):
    """d loss / d E0 through the HIP forward+backward vs torch autograd on a dense fp64 restatement, with a
    dropped (non-symmetric) matrix."""
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.model import LightGCN
    import pandas as pd
    g = golden('g2_synth60')
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    train = pd.DataFrame({'user_id': g['train_u'], 'asin': g['train_i']})
    test = pd.DataFrame({'user_id': g['test_u'], 'asin': g['test_i']})
    gr = NormGraph.from_pairs(g['train_u'], g['train_i'], n_u, n_i)
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=gr,
                               true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
                               train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
                               user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
                               item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}), norm_matrix=None)
    m = LightGCN(_params(k=[5], dropout=0.4, single=single, dropout_rng='cpu'), ds)
    m.training = True
    torch.manual_seed(7)
    keep = (torch.rand(gr.nnz) < 0.6)
    torch.manual_seed(7)
    ue, ie = m.representation
    w = torch.randn(n_u + n_i, 64, device=cuda, generator=torch.Generator(device=cuda).manual_seed(1))
    loss = (torch.cat([ue, ie]) * w).sum()
    loss.backward()
    gu, gi = m.embedding_user.weight.grad, m.embedding_item.weight.grad
    # fp64 dense restatement
    idx, val = gr.to_coo()
    a = torch.zeros(n_u + n_i, n_u + n_i, dtype=torch.float64)
    v = torch.from_numpy(val).double() / 0.6
    a[idx[0][keep.numpy()], idx[1][keep.numpy()]] = v[keep]
    e0 = torch.cat([m.embedding_user.weight, m.embedding_item.weight]).detach().cpu().double().requires_grad_(True)
    cur, layers = e0, [e0]
    for _ in range(3):
        cur = a @ cur
        layers.append(cur)
    out = layers[-1] if single else torch.stack(layers).mean(0)
    (out * w.cpu().double()).sum().backward()
    assert normwise(torch.cat([gu, gi]).cpu().numpy(), e0.grad.numpy()) <= 1e-5
    assert normwise(torch.cat([ue, ie]).detach().cpu().numpy(), out.detach().numpy()) <= 1e-5


def test_fit_runs_and_checkpoint_roundtrip(cuda, tmp_path):
    """A short training run on a synthetic TSV dataset: loss finite, parameters move, checkpoint keys are the
    reference's, reload reproduces predictions."""
    from textgcn_amd import synth
    from textgcn_amd.interactions import InteractionData
    from textgcn_amd.model import LightGCN
    u, i = synth.interactions(120, 80, 1500, seed=3)
    rng = np.random.default_rng(0)
    folder = tmp_path / 'data'
    folder.mkdir()
    te = rng.random(len(u)) < 0.1
    for name, sel in (('train.tsv', ~te), ('test.tsv', te)):
        with open(folder / name, 'w') as f:
            f.write('user_id\tasin\n')
            for a, b in zip(u[sel], i[sel]):
                f.write(f'u{a:04d}\ti{b:04d}\n')
    # every user / item must appear in train for the id maps
    p = _params(k=[5, 10], batch_size=256, epochs=2, evaluate_every=1, save=True, save_path=str(tmp_path), exact=False, data=str(folder))
    try:
        ds = InteractionData(p)
    except AssertionError:
        pytest.skip('random split left a test user without train rows')
    m = LightGCN(p, ds)
    before = m.embedding_user.weight.detach().clone()
    loader = torch.utils.data.DataLoader(ds, batch_size=256, shuffle=True)
    m.fit(loader)
    assert not torch.equal(before, m.embedding_user.weight.detach())
    assert os.path.exists(tmp_path / 'latest_checkpoint.pkl') and os.path.exists(tmp_path / 'best.pkl')
    sd = torch.load(tmp_path / 'latest_checkpoint.pkl')
    assert sorted(sd.keys()) == ['embedding_item.weight', 'embedding_user.weight']
    pred = m.predict(np.arange(ds.n_users))
    p2 = _params(k=[5, 10], batch_size=256, load=str(tmp_path), save_path=str(tmp_path), exact=False, data=str(folder))
    m2 = LightGCN(p2, ds)
    assert m2.predict(np.arange(ds.n_users)) == pred
    # a NaN loss stops the run with the reference's assertion (base_model.py:123), whichever step it appears in
    with torch.no_grad():
        m2.embedding_item.weight[3, 0] = float('nan')
    m2.epochs, m2.evaluate_every = 1, 1
    with pytest.raises(AssertionError, match='loss is NA'):
        m2.fit(loader)


def test_predict_edge_cases(golden, cuda, tmp_path):
    """empty user list, a single user, users in arbitrary order with repeats, batch size 1."""
    from textgcn_amd.interactions import InteractionData
    from textgcn_amd.model import LightGCN
    ds = InteractionData(folder=os.path.join(GOLDEN, 'dummy'), k=[1, 2, 3])
    m = LightGCN(_params(save_path=str(tmp_path), batch_size=1), ds)
    assert m.predict([]) == []
    assert m.predict([], with_scores=True) == ([], [])
    full = m.predict(range(ds.n_users))
    assert m.predict([3]) == [full[3]]
    assert m.predict(np.array([4, 0, 4, 2])) == [full[4], full[0], full[4], full[2]]


def test_predict_streams_do_not_change_results(cuda, tmp_path):
    """Many small chunks round-robin on three streams (shared inputs, per-stream scratch) vs one stream, and vs the
    unfused dense -> mask -> top-k path: identical lists."""
    import pandas as pd
    from textgcn_amd import scoring, synth
    from textgcn_amd.graph import NormGraph, train_mask_csr
    from textgcn_amd.model import LightGCN
    n_u, n_i = 3000, 9000
    u, i = synth.interactions(n_u, n_i, 60000, seed=8)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    rp, items = train_mask_csr(u, i, n_u)
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=g, norm_matrix=None, mask_rowptr=rp, mask_items=items,
                               true_test_lil=[[0]], train_user_dict=None, test_df=pd.DataFrame({'user_id': [0], 'asin': [0]}),
                               user_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['u']}),
                               item_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['i']}))
    m = LightGCN(_params(k=[20, 40], batch_size=64, save_path=str(tmp_path), exact=False), ds)
    m.predict_chunk = 128
    users = np.random.default_rng(0).permutation(n_u)
    m.predict_streams, m._streams = 4, None
    v3, i3 = m.predict_tensors(users)
    m.predict_streams, m._streams = 1, None
    v1, i1 = m.predict_tensors(users)
    assert torch.equal(i3, i1) and torch.equal(v3, v1)
    assert m.score_prefilter          # the class default: candidates from the bf16 pass ...
    m.score_prefilter = False         # ... and the fp32 MFMA filter: the same lists and scores
    vf, jf = m.predict_tensors(users)
    assert torch.equal(jf, i1) and torch.equal(vf, v1)
    with torch.no_grad():
        ue, ie = m.representation
    s = scoring.score_dense(ue.contiguous(), ie.contiguous(), user_ids=torch.from_numpy(users).to(cuda))
    cnt = rp[users + 1] - rp[users]
    brp = np.zeros(len(users) + 1, dtype=np.int32)
    np.cumsum(cnt, out=brp[1:])
    bit = np.concatenate([items[rp[x]:rp[x + 1]] for x in users])
    scoring.mask_train(s, torch.from_numpy(brp).to(cuda), torch.from_numpy(bit).to(cuda))
    rv, ri = scoring.topk(s, 40, round4=True)
    assert torch.equal(i3, ri) and torch.equal(v3, rv)


@pytest.mark.parametrize('d', [64, 128])
def test_ltr_predict_same_with_and_without_the_bf16_candidates(cuda, tmp_path, d):
    """LTRLinear.predict_tensors on a 9000-item catalogue (above the small-catalogue cut, so the fused entry points run): the folded
    operands are 896 / 960 wide -- k_score_prefilter_wide, k_refine and the list rescoring -- and the lists and scores must be the
    fp32 filter's, bit for bit, and the scores the pairwise feature path's within the fp32 bar."""
    import pandas as pd
    import types
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph, train_mask_csr
    from textgcn_amd.ltr import LTRLinear
    n_u, n_i, t = 700, 9000, 384
    u, i = synth.interactions(n_u, n_i, 20000, seed=4)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    rp, items = train_mask_csr(u, i, n_u)
    gen = torch.Generator().manual_seed(d)
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=g, norm_matrix=None, mask_rowptr=rp, mask_items=items,
                               true_test_lil=[[0]], train_user_dict=None, test_df=pd.DataFrame({'user_id': [0], 'asin': [0]}),
                               user_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['u']}),
                               item_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['i']}),
                               items_as_desc=torch.randn((n_i, t), generator=gen), items_as_avg_reviews=torch.randn((n_i, t), generator=gen),
                               users_as_avg_reviews=torch.randn((n_u, t), generator=gen), users_as_avg_desc=torch.randn((n_u, t), generator=gen),
                               all_items=range(n_i))
    p = _params(k=[20, 40], emb_size=d, batch_size=256, save_path=str(tmp_path), exact=False, load_base=None, freeze=True, ltr_layers=[])
    m = LTRLinear(p, ds)
    with torch.no_grad():
        m.layers[0].weight.copy_(torch.tensor([[0.9, 0.05, -0.03, 0.02, 0.04]]))
        m.layers[0].bias.fill_(0.1)
    m.ltr_predict_chunk = 256
    users = np.arange(n_u)
    assert m.score_prefilter and int(m._k()) == d + 2 * t + 64
    v1, i1 = m.predict_tensors(users)
    m.score_prefilter = False
    v0, i0 = m.predict_tensors(users)
    assert torch.equal(i1, i0) and torch.equal(v1, v0)
    with torch.no_grad():
        ue, ie = m.representation
        rows = torch.arange(0, n_u, 7, device=cuda)
        top = i1[rows, 0]
        pair = m.score_pairwise(ue[rows], ie[top], rows, top).reshape(-1)
    assert torch.allclose(pair, v1[rows, 0], atol=2e-4, rtol=1e-4)


def test_model_takes_segmented_kernels_where_they_pay(cuda):
    """A graph whose user table (10 MB) misses an XCD's L2 while an eighth fits: the model's default path segments
    the item rows (propagate.segment_blocks_auto) in inference, in dropout-free training and -- through the gathered
    value stream -- under edge dropout; results agree with the exact chain to rounding, gradients included."""
    import pandas as pd
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph, train_mask_csr
    from textgcn_amd.model import LightGCN
    n_u, n_i, nnz = 40000, 3000, 1300000
    u, i = synth.interactions(n_u, n_i, nnz, seed=7)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    rp, items = train_mask_csr(u, i, n_u)
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=g, norm_matrix=None, mask_rowptr=rp, mask_items=items,
                               true_test_lil=[[0]], train_user_dict=None, test_df=pd.DataFrame({'user_id': [0], 'asin': [0]}),
                               user_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['u']}),
                               item_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['i']}))
    torch.manual_seed(0)
    fast = LightGCN(_params(k=[20], exact=False, dropout=0.0), ds)
    exact = LightGCN(_params(k=[20], exact=True, dropout=0.0), ds)
    _set_weights(exact, fast.embedding_user.weight.detach().cpu().numpy(), fast.embedding_item.weight.detach().cpu().numpy())
    with torch.no_grad():
        fu, fi = fast.representation
        eu, ei = exact.representation
    assert fast._engine.csr.segment_blocks == [0, 8] and not exact._engine.csr.segment_blocks
    assert normwise(torch.cat([fu, fi]).cpu().numpy(), torch.cat([eu, ei]).cpu().numpy()) <= 5e-6
    assert not torch.equal(fi, ei)     # the item rows really took the other kernel
    # training forward + backward (no dropout: stored values; the backward propagates through the same kernels)
    w = torch.randn(g.n, 64, device=cuda)
    grads = []
    for m in (fast, exact):
        m.training = True
        m.zero_grad()
        mu, mi = m.representation
        (torch.cat([mu, mi]) * w).sum().backward()
        grads.append(torch.cat([m.embedding_user.weight.grad, m.embedding_item.weight.grad]).clone())
        m.training = False
    assert normwise(grads[0].cpu().numpy(), grads[1].cpu().numpy()) <= 5e-6
    # edge dropout: per-call values go through the plan's gathered stream; same mask for both models
    fast.dropout = exact.dropout = 0.4
    outs = []
    for m in (fast, exact):
        m.training = True
        torch.manual_seed(5)
        with torch.no_grad():
            mu, mi = m.representation
        outs.append(torch.cat([mu, mi]).clone())
        m.training = False
    assert normwise(outs[0].cpu().numpy(), outs[1].cpu().numpy()) <= 5e-6
    assert not torch.equal(outs[0], torch.cat([fu, fi]))


@pytest.mark.parametrize('device', [torch.device('cuda'), 'cuda', None])
def test_unindexed_device_runs_fit_predict_and_ltr(golden, cuda, tmp_path, device):
    """The reference's parser hands over torch.device('cuda') without an index (TextGCN/parser.py:174), which compares
    unequal to the cuda:0 that tensors report; the package resolves it where it enters.  `None`: the field is absent
    and the class default ('cuda') applies.  Runs the whole drop-in flow: fit, predict, ltr_linear on top."""
    import pandas as pd
    from textgcn_amd import synth
    from textgcn_amd.interactions import InteractionData
    from textgcn_amd.ltr import LTRLinear
    from textgcn_amd.model import LightGCN
    u, i = synth.interactions(90, 60, 1200, seed=5)
    folder = tmp_path / 'data'
    folder.mkdir()
    te = np.zeros(len(u), dtype=bool)
    second = np.unique(u, return_index=True)[1] + 1         # second interaction of every user -> test
    second = second[second < len(u)]
    te[second[u[second] == u[second - 1]]] = True
    for name, sel in (('train.tsv', ~te), ('test.tsv', te)):
        with open(folder / name, 'w') as f:
            f.write('user_id\tasin\n')
            for a, b in zip(u[sel], i[sel]):
                f.write(f'u{a:04d}\ti{b:04d}\n')
    p = _params(k=[5, 10], batch_size=128, epochs=1, evaluate_every=1, save_path=str(tmp_path), exact=False, data=str(folder))
    if device is None:
        del p.device
    else:
        p.device = device
    try:
        ds = InteractionData(p)
    except AssertionError:
        pytest.skip('split left a test user or item without train rows')
    m = LightGCN(p, ds)
    assert m.device == torch.device('cuda', torch.cuda.current_device())
    m.fit(torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True))
    pred = m.predict(np.arange(ds.n_users))
    assert len(pred) == ds.n_users and len(pred[0]) == 10
    # ltr_linear on top of the same (frozen) tables, same unindexed device
    g = torch.Generator().manual_seed(0)
    for name, rows in (('items_as_desc', ds.n_items), ('items_as_avg_reviews', ds.n_items), ('users_as_avg_reviews', ds.n_users),
                       ('users_as_avg_desc', ds.n_users)):
        setattr(ds, name, torch.randn((rows, 384), generator=g))
    ds.all_items = range(ds.n_items)
    p.load_base, p.freeze, p.ltr_layers = None, True, []
    lm = LTRLinear(p, ds)
    lp = lm.predict(np.arange(ds.n_users))
    assert len(lp) == ds.n_users and len(lp[0]) == 10
