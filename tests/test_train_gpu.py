"""N1: the training step on the native path -- get_loss (BPR + L2) and dE0 against the reference's own autograd (golden G8,
captured dropout mask), the fused step against the generic torch composition, and the dropout-values kernel."""
import types

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import normwise

pytestmark = pytest.mark.gpu


def _dataset(g):
    from textgcn_amd.graph import NormGraph
    n_u, n_i = int(g['n_users']), int(g['n_items'])
    train = pd.DataFrame({'user_id': g['train_u'], 'asin': g['train_i']})
    test = pd.DataFrame({'user_id': g['test_u'], 'asin': g['test_i']})
    return types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=NormGraph.from_pairs(g['train_u'], g['train_i'], n_u, n_i),
                                 true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
                                 train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
                                 user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
                                 item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}), norm_matrix=None)


def _model(g, **kw):
    from textgcn_amd.model import LightGCN
    base = dict(k=[5], emb_size=64, n_layers=3, device='cuda:0', load=None, batch_size=2048, quiet=True, save=False, dropout=0.4,
                single=False, exact=True, lr=0.001, epochs=1, reg_lambda=1e-4, evaluate_every=1, neg_samples=1, save_path='.',
                uid='t', dropout_rng='cpu')
    base.update(kw)
    m = LightGCN(types.SimpleNamespace(**base), _dataset(g))
    n_u = int(g['n_users'])
    with torch.no_grad():
        m.embedding_user.weight.copy_(torch.from_numpy(g['a_layer0'][:n_u]))
        m.embedding_item.weight.copy_(torch.from_numpy(g['a_layer0'][n_u:]))
    return m


@pytest.mark.parametrize('name', ['nodrop', 'drop', 'drop2'])
@pytest.mark.parametrize('exact', [True, False])
def test_get_loss_and_gradient_match_reference(golden, cuda, name, exact):
    """G8: loss value (bpr, reg) and dE0 of the reference's get_loss + autograd on a fixed batch; the dropout mask is the
    reference's own CPU draw (seed 123), reproduced through dropout_rng='cpu'."""
    g2, g8 = golden('g2_synth60'), golden('g8_loss')
    m = _model(g2, dropout=float(g8[f'{name}_p']), exact=exact, reg_lambda=float(g8[f'{name}_reg_lambda']))
    assert m._native_loss()
    m.training = True
    torch.manual_seed(123)
    loss = m.get_loss(torch.from_numpy(g8[f'{name}_batch']))
    loss.backward()
    assert abs(float(loss) - float(g8[f'{name}_loss'])) <= 1e-5 * abs(float(g8[f'{name}_loss'])) + 1e-8
    assert abs(float(m._loss_values['bpr']) - float(g8[f'{name}_bpr'])) <= 1e-5 * abs(float(g8[f'{name}_bpr'])) + 1e-8
    assert abs(float(m._loss_values['reg']) - float(g8[f'{name}_reg'])) <= 1e-5 * abs(float(g8[f'{name}_reg']))
    assert normwise(m.embedding_user.weight.grad.cpu().numpy(), g8[f'{name}_grad_user']) <= 1e-4
    assert normwise(m.embedding_item.weight.grad.cpu().numpy(), g8[f'{name}_grad_item']) <= 1e-4


@pytest.mark.parametrize('single', [False, True])
def test_fused_step_equals_generic_torch_composition(golden, cuda, single):
    """The one-node native step against the same loss composed from torch ops over the HIP propagation (the path subclasses
    with their own scoring keep): same mask, same batch -> same loss and gradient to fp32 rounding."""
    g2, g8 = golden('g2_synth60'), golden('g8_loss')
    batch = torch.from_numpy(g8['drop2_batch'])
    res = []
    for native in (True, False):
        m = _model(g2, single=single)
        if not native:
            m._native_loss = lambda: False
        m.training = True
        torch.manual_seed(7)
        loss = m.get_loss(batch)
        loss.backward()
        res.append((float(loss), m.embedding_user.weight.grad.cpu().numpy(), m.embedding_item.weight.grad.cpu().numpy()))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0]) + 1e-8      # --single: the loss is a ~1e-4 difference of sums
    assert normwise(res[0][1], res[1][1]) <= 1e-5 and normwise(res[0][2], res[1][2]) <= 1e-5


def test_dropout_values_kernel(cuda):
    """tgcn_dropout_values_f32: Bernoulli(1 - p) law on the device generator, values / transposed values / segment-stream
    copies consistent with each other, reproducible from the seed; the rand_u form equals the torch formula."""
    from textgcn_amd import _capi, synth
    from textgcn_amd.graph import NormGraph
    u, i = synth.interactions(3000, 1200, 90000, seed=2)
    g = NormGraph.from_pairs(u, i, 3000, 1200)
    nnz = g.nnz
    stored = torch.from_numpy(g.vals).to(cuda)
    scaled = torch.from_numpy((g.vals / np.float32(0.6)).astype(np.float32)).to(cuda)      # the reference's values / (1 - p)
    perm_h = g.transpose_perm()
    perm = torch.from_numpy(perm_h.astype(np.int32)).to(cuda)
    assert np.array_equal(g.vals[perm_h], g.vals)                # symmetric values: stored_vals_t may be NULL
    rng = np.random.default_rng(0)
    src_h = rng.integers(0, nnz, 50000)
    ent_src = torch.from_numpy(src_h.astype(np.int32)).to(cuda)
    ent_src_t = torch.from_numpy(perm_h[src_h].astype(np.int32)).to(cuda)
    ent_stored = stored[ent_src.long()].contiguous()
    lib = _capi.lib()

    def run(rand_u, seed, explicit_t=False):
        out = [torch.empty(nnz, device=cuda), torch.empty(nnz, device=cuda), torch.empty(50000, device=cuda), torch.empty(50000, device=cuda)]
        st = stored[perm.long()].contiguous() if explicit_t else None
        est = None if st is None else st[ent_src.long()].contiguous()
        rc = lib.tgcn_dropout_values_f32(_capi.ptr(stored), _capi.ptr(st), _capi.ptr(rand_u), seed, 0.6, _capi.ptr(perm), _capi.ptr(ent_stored),
                                         _capi.ptr(est), _capi.ptr(ent_src), _capi.ptr(ent_src_t), nnz, 50000,
                                         *[_capi.ptr(t) for t in out], _capi.current_stream(cuda))
        _capi.check(rc, 'tgcn_dropout_values_f32')
        return out
    vals, vals_t, ev, ev_t = run(None, 12345)
    kept = (vals != 0)
    frac = float(kept.float().mean())
    assert abs(frac - 0.6) < 5 * np.sqrt(0.24 / nnz)
    assert torch.equal(vals[kept], scaled[kept])
    assert torch.equal(vals_t, vals[perm.long()])
    assert torch.equal(ev, vals[ent_src.long()]) and torch.equal(ev_t, vals_t[ent_src.long()])
    again = run(None, 12345, explicit_t=True)       # the general (non-symmetric values) form gives the same result here
    assert all(torch.equal(a, b) for a, b in zip((vals, vals_t, ev, ev_t), again))
    other = run(None, 12346)[0]
    assert 0.3 < float(((other != 0) == kept).float().mean()) < 0.7        # a different seed is a different mask (agreement ~ 0.52)
    # entries of one row are not correlated with their neighbours: lag-1 agreement ~ p^2 + (1-p)^2 = 0.52
    lag = float((kept[1:] == kept[:-1]).float().mean())
    assert abs(lag - 0.52) < 0.01
    ru = torch.rand(nnz).to(cuda)
    v2, v2t, _, _ = run(ru, 0)
    want = torch.where(ru < 0.6, scaled, torch.zeros_like(scaled))
    assert torch.equal(v2, want) and torch.equal(v2t, want[perm.long()])


def test_c2_training_step_uses_one_regather_per_values(cuda):
    """the segment plan's value stream is produced once per step (by the dropout launch), not once per spmm call"""
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import EdgeValues, Propagator
    u, i = synth.interactions(20000, 12000, 1_500_000, seed=0, zipf=0.0)
    g = NormGraph.from_pairs(u, i, 20000, 12000)
    prop = Propagator(g, cuda, segment=[0, 8])
    e0 = synth.embeddings(g.n, 64, seed=0).to(cuda)
    ref = prop.forward(e0, 3).clone()
    ev = EdgeValues(prop.csr.vals.clone())
    out = prop.forward(e0, 3, vals=ev)
    assert ev.seg_vals is not None and torch.equal(out, ref)
    first = ev.seg_vals
    prop.forward(e0, 3, vals=ev)
    assert ev.seg_vals is first


@pytest.mark.parametrize('dtype', [torch.int32, torch.int16])
@pytest.mark.parametrize('where', ['cpu', 'cuda'])
def test_get_loss_accepts_any_integer_batch_dtype(golden, cuda, dtype, where):
    """the reference's `users_emb[users]` (base_model.py:189-193) indexes with any integer dtype; the native step reads
    raw int64 ids, so get_loss converts -- an int32 / int16 batch gives the int64 batch's loss and gradient (the gradient rows
    are added by float atomics, whose order differs from run to run: compared normwise)."""
    g2, g8 = golden('g2_synth60'), golden('g8_loss')
    batch = torch.from_numpy(g8['drop2_batch'])
    res = []
    for b in (batch, batch.to(dtype).to(where)):
        m = _model(g2, dropout=0.0)
        assert m._native_loss()
        m.training = True
        loss = m.get_loss(b)
        loss.backward()
        res.append((float(loss), m.embedding_user.weight.grad.cpu().numpy(), m.embedding_item.weight.grad.cpu().numpy()))
    assert res[0][0] == res[1][0] and normwise(res[1][1], res[0][1]) <= 1e-6 and normwise(res[1][2], res[0][2]) <= 1e-6


def test_get_loss_rejects_ids_outside_the_tables(golden, cuda):
    """a host batch is checked before anything is launched; a device batch is clamped (no kernel sees the bad address) and
    the step raises at its next flag read -- IndexError either way, as torch's own indexing in the reference"""
    g2, g8 = golden('g2_synth60'), golden('g8_loss')
    m = _model(g2, dropout=0.0)
    m.training = True
    bad = torch.from_numpy(g8['drop2_batch']).clone()
    bad[3, 1] = m.n_items            # one past the item table
    with pytest.raises(IndexError):
        m.get_loss(bad)
    neg = torch.from_numpy(g8['drop2_batch']).clone()
    neg[0, 0] = -1
    with pytest.raises(IndexError):
        m.get_loss(neg)
    with pytest.raises(TypeError):
        m.get_loss(bad.float())
    with pytest.raises(ValueError):
        m.get_loss(bad[:, :2])
    with pytest.raises(IndexError):          # a DEVICE batch outside fit(): nobody would read the flag later, so it is read here
        m.get_loss(bad.to(cuda))
    m.optimizer = torch.optim.Adam(m.parameters(), lr=1e-3)
    with pytest.raises(IndexError):
        m._train_epoch([bad.to(cuda)], 1)
    torch.cuda.synchronize()
    assert torch.isfinite(m.embedding_user.weight).all() and torch.isfinite(m.embedding_item.weight).all()


@pytest.mark.parametrize('deferred', [False, True])
def test_nan_loss_stops_fit_before_the_weights_take_it(golden, cuda, deferred):
    """base_model.py:123 asserts on the loss BEFORE backward(): with the default order a NaN loss leaves weights and Adam moments
    untouched; the opt-in deferred check raises the same AssertionError after the step"""
    g2, g8 = golden('g2_synth60'), golden('g8_loss')
    m = _model(g2, dropout=0.0)
    m.deferred_nan_check = deferred
    m.optimizer = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    with torch.no_grad():
        m.embedding_user.weight[int(g8['drop2_batch'][0, 0]), 0] = float('nan')
    before = m.embedding_item.weight.detach().clone()
    with pytest.raises(AssertionError, match='loss is NA'):
        m._train_epoch([torch.from_numpy(g8['drop2_batch'])], 1)
    torch.cuda.synchronize()
    if not deferred:
        assert torch.equal(m.embedding_item.weight.detach(), before)
        assert all(len(st) == 0 for st in m.optimizer.state.values())
