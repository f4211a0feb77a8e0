"""Dynamic negative sampling (N4): candidate scores / positives filter / hard-negative selection on the HIP path."""
import types

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import bits

pytestmark = pytest.mark.gpu


def _model(cuda, n_u=300, n_i=500, nnz=6000):
    from textgcn_amd import synth
    from textgcn_amd.adv_sampling import AdvSamplModel
    from textgcn_amd.graph import NormGraph, train_mask_csr
    u, i = synth.interactions(n_u, n_i, nnz, seed=2)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    rp, items = train_mask_csr(u, i, n_u)
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=g, norm_matrix=None, mask_rowptr=rp, mask_items=items,
                               true_test_lil=[[0]], train_user_dict=None, test_df=pd.DataFrame({'user_id': [0], 'asin': [0]}),
                               user_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['u']}),
                               item_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['i']}), pos_samples=5, seed=0)
    p = types.SimpleNamespace(k=[5, 10], emb_size=64, n_layers=2, device='cuda:0', load=None, quiet=True, dropout=0.0, exact=True)
    return AdvSamplModel(p, ds), (u, i, rp, items)


def test_candidate_scores_and_hard_negatives(cuda, oracle):
    m, (u, i, rp, items) = _model(cuda)
    rng = np.random.default_rng(0)
    users = rng.integers(0, 300, 64).astype(np.int64)
    cand = np.stack([rng.choice(500, size=200, replace=False) for _ in users]).astype(np.int64)
    from textgcn_amd import scoring
    with torch.no_grad():
        ue, ie = m.representation
    s = scoring.score_candidates(ue.contiguous(), torch.from_numpy(users).to(cuda), ie.contiguous(), torch.from_numpy(cand).to(cuda))
    uen, ien = ue.cpu().numpy(), ie.cpu().numpy()
    ref = np.stack([oracle.score_pairwise(np.repeat(uen[x][None], 200, 0), ien[c]) for x, c in zip(users, cand)])
    assert np.array_equal(bits(s.cpu().numpy()), bits(ref))
    neg = m.hard_negatives(torch.from_numpy(users).to(cuda), torch.from_numpy(cand).to(cuda)).cpu().numpy()
    for r, (x, c) in enumerate(zip(users, cand)):
        pos = set(items[rp[x]:rp[x + 1]])
        sc = ref[r].copy()
        sc[[j for j, it in enumerate(c) if it in pos]] = -np.inf
        order = sorted(range(200), key=lambda j: (-sc[j], j))[:10]
        want = [c[j] for j in order if np.isfinite(sc[j])]
        got = [v for v in neg[r] if v >= 0]
        assert got == want
        assert not (set(got) & pos)


def test_adv_loss_and_step(cuda):
    m, _ = _model(cuda)
    rng = np.random.default_rng(1)
    users = rng.integers(0, 300, 32)
    data = torch.from_numpy(np.stack([np.concatenate([[x], rng.choice(500, size=100, replace=False)]) for x in users]).astype(np.int64))
    from collections import defaultdict
    m._loss_values = defaultdict(float)
    m.training = True
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    before = m.embedding_item.weight.detach().clone()
    loss = m.get_loss(data)
    assert torch.isfinite(loss)
    loss.backward()
    opt.step()
    assert not torch.equal(before, m.embedding_item.weight.detach())
    pos = m.sample_positives(torch.from_numpy(users).to(cuda))
    assert pos.is_cuda and tuple(pos.shape) == (32, 5)
    pos = pos.cpu().numpy()
    rp, items = m._mask_rowptr_host, m._mask_items_host
    for r, x in enumerate(users):
        mine = set(items[rp[x]:rp[x + 1]])
        got = [v for v in pos[r] if v >= 0]
        assert len(got) == min(5, len(mine)) == len(set(got)) and set(got) <= mine


def test_adv_get_loss_matches_reference(golden, cuda):
    """G11: AdvSamplModel.get_loss (advanced_sampling.py:46-69) run by the reference on the synth-60x40 data.  The hard
    negatives must be the reference's, in its order; with the reference's captured random positives injected, loss, its two
    parts and dE0 must match (normwise 1e-4, the north star's fp32 bar)."""
    from collections import defaultdict
    from conftest import normwise
    from textgcn_amd.adv_sampling import AdvSamplModel
    from textgcn_amd.graph import NormGraph
    g2, g = golden('g2_synth60'), golden('g11_adv_loss')
    n_u, n_i = int(g2['n_users']), int(g2['n_items'])
    train = pd.DataFrame({'user_id': g2['train_u'], 'asin': g2['train_i']})
    test = pd.DataFrame({'user_id': g2['test_u'], 'asin': g2['test_i']})
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=NormGraph.from_pairs(g2['train_u'], g2['train_i'], n_u, n_i), norm_matrix=None,
                               true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
                               train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
                               user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
                               item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}),
                               pos_samples=int(g['pos_samples']), seed=0)
    for exact in (True, False):
        p = types.SimpleNamespace(k=[int(x) for x in g['k']], emb_size=64, n_layers=3, device='cuda:0', load=None, quiet=True, dropout=0.0,
                                  exact=exact, reg_lambda=float(g['reg_lambda']))
        m = AdvSamplModel(p, ds)
        with torch.no_grad():
            m.embedding_user.weight.copy_(torch.from_numpy(g2['a_layer0'][:n_u]))
            m.embedding_item.weight.copy_(torch.from_numpy(g2['a_layer0'][n_u:]))
        batch = torch.from_numpy(g['batch'])
        neg = m.hard_negatives(batch[:, 0].contiguous().to(cuda), batch[:, 1:].contiguous().to(cuda)).cpu().numpy()
        assert np.array_equal(neg, g['negatives'])
        m.sample_positives = lambda users: torch.from_numpy(g['positives'])   # the reference's random.sample draws (not reproducible)
        m._loss_values = defaultdict(float)
        m.training = True
        loss = m.get_loss(batch)
        loss.backward()
        assert abs(float(loss) - float(g['loss'])) <= 1e-5 * abs(float(g['loss'])) + 1e-8
        assert abs(float(m._loss_values['bpr']) - float(g['bpr'])) <= 1e-5 * abs(float(g['bpr'])) + 1e-8
        assert abs(float(m._loss_values['reg']) - float(g['reg'])) <= 1e-5 * abs(float(g['reg']))
        assert normwise(m.embedding_user.weight.grad.cpu().numpy(), g['grad_user']) <= 1e-4
        assert normwise(m.embedding_item.weight.grad.cpu().numpy(), g['grad_item']) <= 1e-4


def test_adv_get_loss_never_waits_for_the_gpu(golden, cuda):
    """AdvSamplModel.get_loss + backward under torch's sync debug mode: the candidate ranking, the positives (drawn on the device
    from the device mask CSR), the pairing and the padded triple block make no host synchronisation (advanced_sampling.py:55-69 is
    a Python loop with a .to(device) per user).  And the device draw is a uniform sample without replacement: every train item of
    a user comes up, never twice in a row of the result."""
    from textgcn_amd.adv_sampling import AdvSamplModel
    from textgcn_amd.graph import NormGraph
    g2 = golden('g2_synth60')
    n_u, n_i = int(g2['n_users']), int(g2['n_items'])
    train = pd.DataFrame({'user_id': g2['train_u'], 'asin': g2['train_i']})
    test = pd.DataFrame({'user_id': g2['test_u'], 'asin': g2['test_i']})
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=NormGraph.from_pairs(g2['train_u'], g2['train_i'], n_u, n_i), norm_matrix=None,
                               true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
                               train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
                               user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
                               item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}), pos_samples=3, seed=0)
    p = types.SimpleNamespace(k=[5, 10], emb_size=64, n_layers=3, device='cuda:0', load=None, quiet=True, dropout=0.4)
    m = AdvSamplModel(p, ds)
    rng = np.random.default_rng(0)
    batch = torch.from_numpy(np.stack([np.concatenate([[u], rng.choice(n_i, 30, replace=False)]) for u in rng.integers(0, n_u, 24)]))
    m.training = True
    m.get_loss(batch).backward()          # warm-up: buffers, plans, lazily built device copies
    torch.cuda.synchronize()
    batch_dev = batch.to(cuda)
    m.zero_grad()
    torch.cuda.set_sync_debug_mode('error')
    try:
        loss = m.get_loss(batch_dev)
        loss.backward()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert torch.isfinite(loss) and torch.isfinite(m.embedding_item.weight.grad).all()
    # distribution of the device draw
    rp, items = m._mask_rowptr_host, m._mask_items_host
    u = int(np.argmax(np.diff(rp) >= 6))
    mine = items[rp[u]:rp[u + 1]]
    draws = m.sample_positives(torch.full((4000,), u, device=cuda)).cpu().numpy()
    assert all(len(set(r)) == 3 for r in draws) and set(draws.ravel()) == set(mine)
    freq = np.array([(draws == it).sum() for it in mine]) / draws.size
    assert np.abs(freq - 1.0 / len(mine)).max() < 0.25 / len(mine)
    with pytest.raises(IndexError):
        bad = batch.clone()
        bad[0, 5] = n_i
        m.get_loss(bad)
