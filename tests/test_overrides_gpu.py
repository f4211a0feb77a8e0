"""The reference's override points on the forward (TextGCN/base_model.py:93-106,141-164): `representation` calls
`self.layer_aggregation(norm_matrix, ...)` K times and `self.layer_combination(cache)` once, so a subclass or an instance that
replaces either changes the forward (rejected_models.py:27-39 shows the pattern).  Checked here by VALUE on the GPU:
`layer_aggregation` multiplies by the matrix it is GIVEN (reference goldens G2 / G3 bit for bit), anything that is not a matrix
is a TypeError, overrides are honoured by `representation` (forward and autograd), and `(-inf, TGCN_NO_ITEM)` fillers never reach
a consumer as an item id."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import bits

pytestmark = pytest.mark.gpu

TGCN_NO_ITEM = 2147483647


def _params(**kw):
    base = dict(k=[5, 10], emb_size=64, n_layers=3, device='cuda:0', load=None, batch_size=32, quiet=True, save=False,
                dropout=0.4, single=False, exact=True, lr=0.001, epochs=1, reg_lambda=1e-4, evaluate_every=1, neg_samples=1,
                save_path='.', uid='t')
    base.update(kw)
    return types.SimpleNamespace(**base)


def _dataset(g, reference_style=False):
    import pandas as pd
    from textgcn_amd.graph import NormGraph
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    train = pd.DataFrame({'user_id': g['train_u'], 'asin': g['train_i']})
    test = pd.DataFrame({'user_id': g['test_u'], 'asin': g['test_i']})
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i,
                               true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
                               train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
                               user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': [f'u{x}' for x in range(n_u)]}),
                               item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': [f'i{x}' for x in range(n_i)]}))
    if reference_style:     # what a reference BaseDataset hands over: the coalesced COO tensor, no CSR
        ds.norm_matrix = torch.sparse_coo_tensor(torch.from_numpy(g['norm_idx']), torch.from_numpy(g['norm_val']),
                                                 (n_u + n_i,) * 2).coalesce()
    else:
        ds.graph = NormGraph.from_pairs(g['train_u'], g['train_i'], n_u, n_i)
        ds.norm_matrix = None
    return ds


def _model(golden, cls=None, reference_style=False, **kw):
    from textgcn_amd.model import LightGCN
    g = golden('g2_synth60')
    m = (cls or LightGCN)(_params(**kw), _dataset(g, reference_style))
    n_u = int(g['n_users'])
    with torch.no_grad():
        m.embedding_user.weight.copy_(torch.from_numpy(g['a_layer0'][:n_u]))
        m.embedding_item.weight.copy_(torch.from_numpy(g['a_layer0'][n_u:]))
    return m, g


@pytest.mark.parametrize('reference_style', [False, True])
def test_layer_aggregation_on_the_models_matrix_is_g2_layer1(golden, cuda, reference_style):
    """layer_aggregation(model.norm_matrix, E0) == the reference's layer 1, bit for bit, and chained K times + layer_combination ==
    the reference's representation (golden G2: torch.sparse.mm on the CPU + mean(stack))."""
    m, g = _model(golden, reference_style=reference_style)
    nm = m.norm_matrix
    assert nm.layout == torch.sparse_coo and tuple(nm.shape) == (m.n_users + m.n_items,) * 2
    assert np.array_equal(nm.coalesce().indices().cpu().numpy(), g['norm_idx'])
    assert np.array_equal(bits(nm.coalesce().values().cpu().numpy()), bits(g['norm_val']))
    e0 = torch.from_numpy(g['a_layer0']).to(cuda)
    with torch.no_grad():
        cache = [e0]
        for k in range(1, 4):
            cache.append(m.layer_aggregation(nm, cache[-1]))
            assert np.array_equal(bits(cache[-1].cpu().numpy()), bits(g[f'a_layer{k}'])), k
        comb = m.layer_combination(cache)
    assert np.array_equal(bits(comb.cpu().numpy()), bits(np.concatenate([g['a_users_emb'], g['a_items_emb']])))
    assert m.layer_combination_single(cache) is cache[-1]


def test_layer_aggregation_uses_the_matrix_it_is_given(golden, cuda, oracle):
    """G3's DROPPED matrix handed over as a torch sparse tensor (CPU or GPU, COO or CSR): three products + the mean are the
    reference's training-mode representation bit for bit -- with the model's own matrix the result would be G2's.  A rectangular
    matrix and a DeviceCSR work too; a dense tensor, None or an array raise TypeError."""
    from textgcn_amd.propagate import DeviceCSR
    m, g = _model(golden)
    g3 = golden('g3_dropout')
    n = m.n_users + m.n_items
    dropped = torch.sparse_coo_tensor(torch.from_numpy(g3['drop_idx']), torch.from_numpy(g3['drop_val']), (n, n)).coalesce()
    want = np.concatenate([g3['users_emb'], g3['items_emb']])
    e0 = torch.from_numpy(g['a_layer0']).to(cuda)
    for matrix in (dropped, dropped.to(cuda), dropped.to_sparse_csr()):
        with torch.no_grad():
            cache = [e0]
            for _ in range(3):
                cache.append(m.layer_aggregation(matrix, cache[-1]))
            got = m.layer_combination(cache)
        assert np.array_equal(bits(got.cpu().numpy()), bits(want)), matrix.layout
        assert not np.array_equal(bits(cache[1].cpu().numpy()), bits(g['a_layer1']))
    # converted once, remembered by identity
    csr1 = m._resolve_matrix(dropped)[0]
    assert m._resolve_matrix(dropped)[0] is csr1
    # rectangular: the item rows of the dropped matrix only ([I, N] x [N, d])
    idx = g3['drop_idx']
    sel = idx[0] >= m.n_users
    rect = torch.sparse_coo_tensor(torch.from_numpy(np.stack([idx[0][sel] - m.n_users, idx[1][sel]])), torch.from_numpy(g3['drop_val'][sel]),
                                   (m.n_items, n)).coalesce()
    with torch.no_grad():
        part = m.layer_aggregation(rect, e0)
        full = m.layer_aggregation(dropped, e0)
    assert torch.equal(part.view(torch.int32), full[m.n_users:].view(torch.int32))
    # a DeviceCSR
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(idx[0], minlength=n), out=rp[1:])
    dc = DeviceCSR(rp, idx[1], g3['drop_val'], n, cuda)
    with torch.no_grad():
        assert torch.equal(m.layer_aggregation(dc, e0).view(torch.int32), full.view(torch.int32))
    for bad in (None, e0, g3['drop_val'], dropped.to(torch.float64)):
        with pytest.raises(TypeError):
            m.layer_aggregation(bad, e0)
    with pytest.raises(ValueError):
        m.layer_aggregation(dropped, e0[:-1])


def test_layer_aggregation_backward_through_a_foreign_matrix(golden, cuda):
    """d/dE of sum(w * (A_dropped E)) = A_dropped^T w: the transposed CSR of a caller's (non-symmetric) matrix, against torch's
    own sparse autograd on the CPU in float64."""
    m, g = _model(golden)
    g3 = golden('g3_dropout')
    n = m.n_users + m.n_items
    dropped = torch.sparse_coo_tensor(torch.from_numpy(g3['drop_idx']), torch.from_numpy(g3['drop_val']), (n, n)).coalesce()
    rng = np.random.default_rng(0)
    w = torch.from_numpy(rng.standard_normal((n, 64)).astype(np.float32))
    e = torch.from_numpy(g['a_layer0']).to(cuda).requires_grad_(True)
    y = m.layer_aggregation(dropped, e)
    (y * w.to(cuda)).sum().backward()
    e64 = torch.from_numpy(g['a_layer0']).double().requires_grad_(True)
    (torch.sparse.mm(dropped.double(), e64) * w.double()).sum().backward()
    err = (e.grad.cpu().double() - e64.grad).abs().max() / e64.grad.abs().max()
    assert float(err) <= 1e-6


def _torch_reference_forward(nm_cpu, e0, K, agg, comb):
    """the reference's loop (base_model.py:99-105) with the given members, torch CPU"""
    cur = e0
    cache = [cur]
    for _ in range(K):
        cur = agg(nm_cpu, cur)
        cache.append(cur)
    return comb(cache)


def test_subclass_layer_combination_changes_representation(golden, cuda):
    """A subclass with a WEIGHTED layer_combination: representation must be what the torch composition says, not the fused layer
    mean -- forward values (the layers come from the HIP kernel: bit-exact inputs to the same torch expression), predict, and the
    gradient through get_loss (generic composition: native node off)."""
    from textgcn_amd.model import LightGCN
    wts = [0.5, 0.25, 0.125, 0.125]

    class Weighted(LightGCN):
        def layer_combination(self, vectors):
            return sum(w * v for w, v in zip(wts, vectors))

    m, g = _model(golden, cls=Weighted)
    assert m._overridden('layer_combination') and not m._overridden('layer_aggregation') and not m._native_loss()
    layers = [g[f'a_layer{k}'] for k in range(4)]
    want = sum(np.float32(w) * torch.from_numpy(v) for w, v in zip(wts, layers)).numpy()
    with torch.no_grad():
        ue, ie = m.representation
    got = torch.cat([ue, ie]).cpu().numpy()
    assert np.array_equal(bits(got), bits(want))
    plain, _ = _model(golden)
    with torch.no_grad():
        assert not torch.equal(torch.cat(plain.representation), torch.cat([ue, ie]))
    # predict goes through the override too
    pred = np.asarray(m.predict(np.arange(m.n_users)))
    s = want[:m.n_users] @ want[m.n_users:].T
    for u, items in m.train_user_dict.items():
        s[u, items] = -np.inf
    top1 = s.argmax(axis=1)
    assert (pred[:, 0] == top1).mean() >= 0.95      # BLAS vs the k-ordered chain: near-ties may swap
    # training: autograd through K layer_aggregation calls + the override, against float64 torch on the CPU
    m.training = True
    m.dropout = 0.0
    batch = torch.from_numpy(np.stack([np.arange(20) % m.n_users, np.arange(20) % m.n_items, (np.arange(20) * 7 + 3) % m.n_items], axis=1))
    m.zero_grad()
    m.get_loss(batch).backward()
    nm = m.norm_matrix.cpu().double()
    wu = torch.from_numpy(g['a_layer0'][:m.n_users]).double().requires_grad_(True)
    wi = torch.from_numpy(g['a_layer0'][m.n_users:]).double().requires_grad_(True)
    out = _torch_reference_forward(nm, torch.cat([wu, wi]), 3, torch.sparse.mm, lambda vs: sum(w * v for w, v in zip(wts, vs)))
    au, ai = out[:m.n_users], out[m.n_users:]
    users, pos, neg = batch[:, 0], batch[:, 1], batch[:, 2]
    s_pos = (au[users] * ai[pos]).sum(1)
    s_neg = (au[users] * ai[neg]).sum(1)
    loss = torch.nn.functional.selu(s_neg - s_pos).mean()
    reg = (wu[users].norm(2).pow(2) + wi[pos].norm(2).pow(2) + wi[neg].norm(2).pow(2)) * (1e-4 / (2 * len(users)))
    (loss + reg).backward()
    for got_g, ref_g in ((m.embedding_user.weight.grad, wu.grad), (m.embedding_item.weight.grad, wi.grad)):
        err = (got_g.cpu().double() - ref_g).abs().max() / ref_g.abs().max()
        assert float(err) <= 1e-5


def test_overridden_layer_aggregation_gets_a_torch_sparse_matrix(golden, cuda):
    """An override of layer_aggregation (on the INSTANCE, as ltr_models.py:177-179 rebinds members; and on a subclass that calls
    super()) receives a torch sparse COO tensor on the model's device -- the model's matrix in eval mode, this step's dropped
    matrix in training -- and its result is what representation combines."""
    from textgcn_amd.model import LightGCN
    m, g = _model(golden, dropout_rng='cpu', dropout=0.4)
    seen = []

    def half(norm_matrix, emb):
        seen.append(norm_matrix)
        return torch.sparse.mm(norm_matrix, emb) * 0.5

    m.layer_aggregation = half
    assert m._overridden('layer_aggregation')
    with torch.no_grad():
        ue, ie = m.representation
    assert len(seen) == 3 and all(s.layout == torch.sparse_coo and s.device.type == 'cuda' for s in seen)
    layers = [torch.from_numpy(g['a_layer0'])]
    nm = m.norm_matrix.cpu()
    for _ in range(3):
        layers.append(torch.sparse.mm(nm, layers[-1]) * 0.5)
    want = torch.mean(torch.stack(layers), axis=0).numpy()
    got = torch.cat([ue, ie]).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
    # training mode: the override sees the dropped matrix of THIS step (reference mask stream), and super() multiplies by it
    g3 = golden('g3_dropout')

    class Sub(LightGCN):
        def layer_aggregation(self, norm_matrix, emb_matrix):
            self.seen.append(norm_matrix)
            return super().layer_aggregation(norm_matrix, emb_matrix)

    m2, _ = _model(golden, cls=Sub, dropout_rng='cpu', dropout=float(g3['p']))
    m2.seen = []
    m2.training = True
    torch.manual_seed(123)
    with torch.no_grad():
        ue, ie = m2.representation
    assert np.array_equal(bits(ue.cpu().numpy()), bits(g3['users_emb']))
    assert np.array_equal(bits(ie.cpu().numpy()), bits(g3['items_emb']))
    d = m2.seen[0].coalesce()
    assert np.array_equal(d.indices().cpu().numpy(), g3['drop_idx'])
    assert np.array_equal(bits(d.values().cpu().numpy()), bits(g3['drop_val']))


def test_single_is_not_an_override(golden, cuda):
    from textgcn_amd.model import LightGCN
    g = golden('g2_synth60')
    m = LightGCN(_params(single=True), _dataset(g))
    assert not m._overridden('layer_combination') and not m._overridden('layer_aggregation')


def test_no_item_fillers_never_reach_a_consumer(golden, cuda, tmp_path):
    """A user row with NaN scores has fewer than k rankable items: its list ends in (-inf, TGCN_NO_ITEM).  predict(save=True)
    must not look the filler up in the item mapping, and evaluate() must not count it as ANOTHER user's relevant item
    (row * n_items + 2^31 - 1 lands in a later user's key range)."""
    from textgcn_amd.metrics import ranking_metrics, ranking_metrics_device
    from textgcn_amd.model import LightGCN
    g = golden('g2_synth60')
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    m = LightGCN(_params(k=[5, 40], save_path=str(tmp_path), n_layers=0), _dataset(g))
    with torch.no_grad():
        m.embedding_user.weight.copy_(torch.from_numpy(g['a_layer0'][:n_u]))
        m.embedding_item.weight.copy_(torch.from_numpy(g['a_layer0'][n_u:]))
        m.embedding_user.weight[3] = float('nan')          # every score of user 3 is NaN
        m.embedding_item.weight[:5] = float('nan')         # and five items are NaN for everybody: 35 rankable items < k = 40
    val, idx = m.predict_tensors(np.arange(n_u))
    assert int((idx == TGCN_NO_ITEM).sum()) > 0
    pred, scores = m.predict(np.arange(n_u), save=True, with_scores=True)
    lines = open(os.path.join(str(tmp_path), 'predictions.tsv')).read().splitlines()
    assert len(lines) == n_u + 1
    row3 = lines[4].split('\t')
    # user 3: no rankable score at all -- only its masked train items (score -inf) can stand in the list, the rest are fillers
    mine = sorted(int(x) for x in m.train_user_dict[3])
    assert row3[0] == 'u3' and eval(row3[1]) == [f'i{x}' for x in mine]       # (a masked NaN item scores -inf like any train item)
    assert all(v == float('-inf') for v in eval(row3[2], {'inf': float('inf')}))
    for u, ln in enumerate(lines[1:]):      # every other user: the 35 finite items + those of the five NaN items it trained on (-inf)
        if u != 3:
            assert len(eval(ln.split('\t')[1])) == n_i - 5 + len(set(int(x) for x in m.train_user_dict.get(u, [])) & set(range(5))), u
    res = m.evaluate()       # no exception
    # the device metrics treat a filler as a miss: equal to the list form on the same predictions
    _, idx_t = m.predict_tensors(m.test_users)
    want = ranking_metrics(m.true_test_lil, idx_t.cpu().numpy(), m.k)
    for name in want:
        assert np.allclose(res[name], want[name], atol=1e-12), name
    # a crafted foreign hit: with 2^30 items, user 0's filler key 0 * 2^30 + (2^31 - 1) is user 1's item 2^30 - 1
    span = 1 << 30
    ptr = torch.tensor([0, 0, 1], dtype=torch.int64, device=cuda)
    out = ranking_metrics_device(ptr, torch.tensor([span - 1], device=cuda),
                                 torch.tensor([[TGCN_NO_ITEM, TGCN_NO_ITEM], [5, 6]], dtype=torch.int64, device=cuda), [2], span)
    assert out['hit'] == [0.0]
    ptr = torch.arange(0, 3, dtype=torch.int64, device=cuda)
    out = ranking_metrics_device(ptr, torch.tensor([7, 7], device=cuda), torch.tensor([[TGCN_NO_ITEM, 7], [TGCN_NO_ITEM - 1, -1]], device=cuda),
                                 [2], 10)
    assert out['hit'] == [0.5]
