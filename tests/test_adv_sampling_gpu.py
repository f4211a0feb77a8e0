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
    pos = m.sample_positives(users)
    assert pos.shape == (32, 5)
    rp, items = m._mask_rowptr_host, m._mask_items_host
    for r, x in enumerate(users):
        mine = set(items[rp[x]:rp[x + 1]])
        got = [v for v in pos[r] if v >= 0]
        assert len(got) == min(5, len(mine)) == len(set(got)) and set(got) <= mine
