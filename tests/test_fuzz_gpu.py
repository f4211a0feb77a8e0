"""Seeded random sweeps over shapes and data styles the fixed cases do not name (tile boundaries +-1, k up to 128, widths 1..1024,
tied scores, non-finite rows, isolated nodes, rows longer than the split threshold).  TGCN_FUZZ_SCALE=n multiplies the number of
cases (the default counts keep the file at a few seconds); tools/fuzz_parity.py runs the scoring sweep by the clock."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, bits, normwise

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, 'tools'))
SCALE = max(1, int(os.environ.get('TGCN_FUZZ_SCALE', '1')))


@pytest.mark.parametrize('block', range(6))
def test_fused_scoring_random_shapes(cuda, block):
    """tgcn_score_topk_f32 / tgcn_score_topk_prefilter_f32 == dense -> mask -> top-k, bit for bit (fuzz_parity.draw_case)."""
    import fuzz_parity
    n = 50 * SCALE
    for seed in range(block * n, (block + 1) * n):
        desc, u, it, mask, ids = fuzz_parity.draw_case(seed, wide=(block == 5))
        err = fuzz_parity.run_case(cuda, desc, u, it, mask, ids)
        assert err is None, (desc, err)


def _draw_graph(seed):
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    rng = np.random.default_rng(10_000 + seed)
    n_u = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1000, 2500]))
    n_i = int(rng.choice([1, 3, 4, 64, 129, 500, 1500]))
    nnz = int(min(n_u * n_i, rng.choice([1, 10, 500, 5000, 40000])))
    zipf = float(rng.choice([0.0, 0.8, 1.2]))
    u, i = synth.interactions(n_u, n_i, nnz, seed=seed, zipf=zipf)
    iso_u, iso_i = int(rng.integers(0, 4)), int(rng.integers(0, 4))      # isolated nodes at the end of both blocks
    if rng.random() < 0.3 and len(u) > 4:                                # ... and in the middle: drop every edge of a few nodes
        drop_u = rng.choice(n_u, size=min(3, n_u), replace=False)
        keep = ~np.isin(u, drop_u)
        if keep.any():
            u, i = u[keep], i[keep]
    return NormGraph.from_pairs(u, i, n_u + iso_u, n_i + iso_i), rng


@pytest.mark.parametrize('block', range(4))
def test_spmm_random_graphs_vs_oracle(cuda, oracle, block):
    """One layer (every kernel variant, exact = bit for bit; the long-row split and the XCD segments to rounding, short / direct rows
    still bit for bit) and the K-layer forward with the fused running mean against the oracle's restatement."""
    from textgcn_amd.propagate import DeviceCSR, Propagator, spmm
    n = 6 * SCALE
    for seed in range(block * n, (block + 1) * n):
        gr, rng = _draw_graph(seed)
        d = int(rng.choice([1, 3, 8, 16, 32, 48, 64, 100, 128, 200, 256]))
        scale = np.exp2(rng.integers(-10, 6, size=(gr.n, 1))) if rng.random() < 0.3 else 0.1
        x = (rng.standard_normal((gr.n, d)) * scale).astype(np.float32)
        idx, val = gr.to_coo()
        ref = oracle.spmm_coo(idx, val, x, n_rows=gr.n)
        xd = torch.from_numpy(x).to(cuda)
        what = dict(seed=seed, n_users=gr.n_users, n_items=gr.n_items, nnz=int(gr.nnz), d=d)
        thr = int(rng.choice([4, 16, 100, 1024]))
        csr = DeviceCSR(gr.rowptr, gr.colidx, gr.vals, gr.n, cuda, split_threshold=thr)
        for variant in (0, 1):
            y = torch.full((gr.n, d), float('nan'), device=cuda)
            spmm(csr, xd, y=y, exact=True, variant=variant)
            assert np.array_equal(bits(y.cpu().numpy()), bits(ref)), (what, 'exact', variant)
        y = torch.full((gr.n, d), float('nan'), device=cuda)
        spmm(csr, xd, y=y)                                        # rows above thr are cut in chunks
        got = y.cpu().numpy()
        short = gr.degrees() <= thr
        assert np.array_equal(bits(got[short]), bits(ref[short])), (what, 'split: short rows', thr)
        assert normwise(got, ref) <= 1e-5, (what, 'split', thr)
        K = int(rng.integers(0, 5))
        single = bool(rng.random() < 0.3)
        e0 = x
        want, _ = oracle.propagate(idx, val, e0, K, single=single)
        prop = Propagator(gr, cuda, split_threshold=thr)
        out = prop.forward(xd, K, exact=True, single=single).cpu().numpy()
        assert np.array_equal(bits(out), bits(want)), (what, 'forward exact', K, single)
        if d in (64, 128, 256) and gr.nnz > 0:
            blocks = [[0, 8, 16, (4, 4)][int(rng.integers(0, 4))], int(rng.choice([0, 8, 24]))]      # (blocks, classes): two XCDs per block
            if any(blocks):
                prop.csr.configure_segments(blocks, tile_entries=int(rng.choice([64, 256])), min_row_len=int(rng.choice([0, 8, 48])))
                y = torch.full((gr.n, d), float('nan'), device=cuda)
                spmm(prop.csr, xd, y=y, segmented=True)
                assert normwise(y.cpu().numpy(), ref) <= 1e-5, (what, 'segmented', blocks)
        # the default (non-exact) forward against the path's bar, 1e-4 normwise: star graphs (one item, 2500 users: a row of 2500
        # entries summed in chunks instead of one chain) reach 2.3e-5 after three layers at 25x scale; everything else stays < 1e-5
        out = prop.forward(xd, K, single=single).cpu().numpy()
        assert normwise(out, want) <= 1e-4, (what, 'forward', K, single)


def _ltr_case(seed, tmp_path, pop):
    import types

    import pandas as pd
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.ltr import LTRLinear, LTRLinearWPop
    rng = np.random.default_rng(20_000 + seed)
    n_u = int(rng.choice([3, 33, 130, 700]))
    n_i = int(rng.choice([5, 64, 257, 2100]))
    d = int(rng.choice([8, 16, 48, 64, 128]))
    tw = int(rng.choice([4, 8, 20, 100, 384]))
    nnz = int(min(n_u * n_i // 2, rng.choice([10, 300, 4000])))
    u, i = synth.interactions(n_u, n_i, max(nnz, 1), seed=seed)
    test_u = np.arange(n_u)
    test_i = rng.integers(0, n_i, size=n_u)
    train = pd.DataFrame({'user_id': u, 'asin': i})
    test = pd.DataFrame({'user_id': test_u, 'asin': test_i})
    t = lambda r, c: torch.from_numpy(rng.standard_normal((r, c)).astype(np.float32))  # noqa: E731
    ds = types.SimpleNamespace(
        n_users=n_u, n_items=n_i, graph=NormGraph.from_pairs(u, i, n_u, n_i), norm_matrix=None,
        true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
        train_user_dict=train.groupby('user_id')['asin'].aggregate(list), test_df=test,
        user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
        item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}), all_items=range(n_i),
        items_as_desc=t(n_i, tw), items_as_avg_reviews=t(n_i, tw), users_as_avg_reviews=t(n_u, tw), users_as_avg_desc=t(n_u, tw),
        popularity_users=t(n_u, 1), popularity_items=t(n_i, 1))
    kmax = int(min(n_i - int(train.groupby('user_id').size().max()), rng.choice([1, 5, 40])))
    p = types.SimpleNamespace(k=[max(kmax, 1)], emb_size=d, n_layers=int(rng.integers(1, 4)), device='cuda:0', load=None, load_base=None,
                              freeze=True, batch_size=int(rng.choice([16, 2048])), quiet=True, ltr_layers=[[], [3]][int(rng.integers(0, 2))],
                              save_path=str(tmp_path))
    m = (LTRLinearWPop if pop else LTRLinear)(p, ds)
    return m, ds, dict(seed=seed, n_users=n_u, n_items=n_i, d=d, text=tw, k=p.k, layers=p.ltr_layers, pop=pop)


@pytest.mark.parametrize('pop', [False, True])
def test_ltr_folded_scores_random_widths(cuda, tmp_path, pop):
    """LTRLinear / LTRLinearWPop at table widths the goldens do not have: the folded one-GEMM score against the reference's
    five-feature formula restated in torch (ltr_models.py:131-146, 181-190, 233-241), and predict_tensors (fold -> fused top-k,
    bf16 candidates) against mask + top-k over the model's own [B, I] score matrix, bit for bit."""
    from textgcn_amd import scoring
    for seed in range(8 * SCALE):
        m, ds, what = _ltr_case(seed, tmp_path, pop)
        with torch.no_grad():
            ue, ie = m.representation
            users = torch.arange(m.n_users, device=cuda)
            s = m.score_batchwise(ue[users], ie, users)
            feats = [ue @ ie.T, m.users_as_avg_reviews @ m.items_as_avg_reviews.T, m.users_as_avg_desc @ m.items_as_desc.T,
                     m.users_as_avg_reviews @ m.items_as_desc.T, m.users_as_avg_desc @ m.items_as_avg_reviews.T]
            if pop:
                feats += [m.popularity_users.expand(-1, m.n_items), m.popularity_items.T.expand(m.n_users, -1)]
            ref = m.layers(torch.stack([f.double() for f in feats], dim=-1).float()).squeeze(-1)
            assert normwise(s.cpu().numpy(), ref.cpu().numpy()) <= 1e-4, what
            val, idx = m.predict_tensors(np.arange(m.n_users))
            rp, items = ds.graph.train_mask()
            sm = s.clone()
            scoring.mask_train(sm, torch.as_tensor(rp, dtype=torch.int32, device=cuda), torch.as_tensor(items, dtype=torch.int32, device=cuda))
            rv, ri = scoring.topk(sm, max(m.k), round4=True)
        assert torch.equal(idx, ri), (what, int((idx != ri).sum()))
        assert np.array_equal(bits(val.cpu().numpy()), bits(rv.cpu().numpy())), what


def test_training_step_random_shapes_vs_float64_autograd(cuda, tmp_path):
    """get_loss + backward of the fused native step (edge dropout as value masking, K-layer forward, BPR over 1-3 negatives, L2,
    the transposed propagation) against the reference's formulas (base_model.py:77-86, 93-106, 181-210) restated densely in float64
    with torch autograd on the CPU; the dropout draw is the reference's own CPU stream (dropout_rng='cpu')."""
    import types

    import pandas as pd
    import torch.nn.functional as F
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.model import LightGCN
    for seed in range(10 * SCALE):
        rng = np.random.default_rng(30_000 + seed)
        n_u, n_i = int(rng.choice([2, 40, 300])), int(rng.choice([3, 65, 200]))
        nnz = int(min(n_u * n_i // 2, rng.choice([5, 400, 3000])))
        u, i = synth.interactions(n_u, n_i, max(nnz, 1), seed=seed, zipf=float(rng.choice([0.0, 1.0])))
        gr = NormGraph.from_pairs(u, i, n_u, n_i)
        d, K = int(rng.choice([8, 32, 64, 100, 128, 256])), int(rng.integers(1, 5))
        n_neg, p = int(rng.integers(1, 4)), float(rng.choice([0.0, 0.3]))
        single, exact = bool(rng.random() < 0.3), bool(rng.random() < 0.5)
        b = int(rng.choice([1, 7, 64, 500]))
        what = dict(seed=seed, n_users=n_u, n_items=n_i, nnz=int(gr.nnz), d=d, K=K, n_neg=n_neg, p=p, single=single, exact=exact, b=b)
        train = pd.DataFrame({'user_id': u, 'asin': i})
        ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=gr, norm_matrix=None, test_df=train,
                                   true_test_lil=train.groupby('user_id')['asin'].aggregate(list).values.tolist(),
                                   train_user_dict=train.groupby('user_id')['asin'].aggregate(list),
                                   user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
                                   item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}))
        lam = float(rng.choice([0.0, 1e-4, 1e-2]))
        m = LightGCN(types.SimpleNamespace(k=[1], emb_size=d, n_layers=K, device='cuda:0', load=None, batch_size=2048, quiet=True, save=False,
                                           dropout=p, single=single, exact=exact, lr=1e-3, epochs=1, reg_lambda=lam, evaluate_every=1,
                                           neg_samples=n_neg, save_path=str(tmp_path), uid='f', dropout_rng='cpu'), ds)
        assert m._native_loss(), what
        batch = np.concatenate([rng.integers(0, n_u, size=(b, 1)), rng.integers(0, n_i, size=(b, 1 + n_neg))], axis=1)
        m.training = True
        torch.manual_seed(seed)
        loss = m.get_loss(torch.from_numpy(batch))
        loss.backward()
        # ---- the reference's step in float64 on the CPU
        torch.manual_seed(seed)
        idx, val = gr.to_coo()
        v = torch.from_numpy(val).double()
        if p > 0:
            keep = torch.rand(gr.nnz) < 1 - p                            # base_model.py:82-84 (the same CPU stream)
            v = torch.where(keep, v / (1 - p), torch.zeros_like(v))
        A = torch.zeros((gr.n, gr.n), dtype=torch.float64)
        A[torch.from_numpy(idx[0]), torch.from_numpy(idx[1])] = v
        eu = m.embedding_user.weight.detach().cpu().double().requires_grad_(True)
        ei = m.embedding_item.weight.detach().cpu().double().requires_grad_(True)
        layers = [torch.cat([eu, ei])]
        for _ in range(K):
            layers.append(A @ layers[-1])
        e = layers[-1] if single else torch.stack(layers).mean(0)
        au, ai = e[:n_u], e[n_u:]
        bt = torch.from_numpy(batch)
        us, pos, negs = bt[:, 0], bt[:, 1], [bt[:, 2 + j] for j in range(n_neg)]
        s_pos = (au[us] * ai[pos]).sum(1)
        bpr = torch.stack([F.selu((au[us] * ai[n]).sum(1) - s_pos).mean() for n in negs]).sum() / n_neg
        sq = lambda t: t.norm(2).pow(2)  # noqa: E731
        reg = (sq(eu[us]) + sq(ei[pos]) + sq(ei[torch.stack(negs)]).mean()) * (lam / (2 * b))
        (bpr + reg).backward()
        ref = float((bpr + reg).detach())
        assert abs(float(loss.detach()) - ref) <= 2e-5 * abs(ref) + 1e-7, (what, float(loss.detach()), ref)
        # (a batch whose negative IS its positive has an exactly zero float64 gradient and fp32 rounding residue around it:
        # the bar is relative to the largest gradient entry with a floor at the rounding of one term, eps * max|E0|)
        floor = 1e-7 * float(max(eu.detach().abs().max(), ei.detach().abs().max()))
        for got, want in ((m.embedding_user.weight.grad, eu.grad), (m.embedding_item.weight.grad, ei.grad)):
            err = float((got.cpu().double() - want).abs().max())
            assert err <= 1e-4 * float(want.abs().max()) + floor, (what, err, float(want.abs().max()))


def test_predict_and_evaluate_random_datasets_vs_oracle(cuda, oracle, tmp_path):
    """The whole inference path of the model class (BaseModel.predict / evaluate, base_model.py:212-276) in exact mode against the
    oracle chained end to end: propagate -> dense scores -> train mask -> top-k (+ round) -> metrics.  Lists and scores bit for
    bit (finite prefix: the order among masked -inf items is undefined in the reference too), metrics to 1e-12; the users asked
    for come as a shuffled subset, in chunks smaller and larger than the batch, with and without the bf16 candidate pass."""
    import types

    import pandas as pd
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.model import LightGCN
    for seed in range(8 * SCALE):
        rng = np.random.default_rng(70_000 + seed)
        n_u, n_i = int(rng.choice([3, 90, 700, 2300])), int(rng.choice([6, 130, 1100, 9000]))
        nnz = int(min(n_u * n_i // 3, rng.choice([20, 900, 12000])))
        u, i = synth.interactions(n_u, n_i, max(nnz, 1), seed=seed, zipf=float(rng.choice([0.0, 0.8])))
        gr = NormGraph.from_pairs(u, i, n_u, n_i)
        max_train = int(np.bincount(u, minlength=n_u).max())
        ks = sorted(set(int(k) for k in rng.choice(np.arange(1, max(2, min(n_i - max_train, 130 if seed % 3 == 0 else 50))), size=2)))      # (above 64: the multi-pass wrapper)
        d, K, single = int(rng.choice([16, 64, 100, 128])), int(rng.integers(1, 5)), bool(rng.random() < 0.3)
        test_u = rng.integers(0, n_u, size=max(1, n_u // 2))
        test = pd.DataFrame({'user_id': test_u, 'asin': rng.integers(0, n_i, size=len(test_u))}).sort_values('user_id')
        train = pd.DataFrame({'user_id': u, 'asin': i})
        ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=gr, norm_matrix=None, test_df=test,
                                   true_test_lil=test.groupby('user_id')['asin'].aggregate(list).values.tolist(),
                                   train_user_dict=train.groupby('user_id')['asin'].aggregate(list),
                                   user_mapping=pd.DataFrame({'remap_id': range(n_u), 'org_id': range(n_u)}),
                                   item_mapping=pd.DataFrame({'remap_id': range(n_i), 'org_id': range(n_i)}))
        m = LightGCN(types.SimpleNamespace(k=ks, emb_size=d, n_layers=K, device='cuda:0', load=None, batch_size=int(rng.choice([7, 256, 2048])),
                                           quiet=True, save=False, dropout=0.4, single=single, exact=True, lr=1e-3, epochs=1, reg_lambda=1e-4,
                                           evaluate_every=1, neg_samples=1, save_path=str(tmp_path), uid='p'), ds)
        m.score_prefilter = bool(rng.random() < 0.6)
        what = dict(seed=seed, n_users=n_u, n_items=n_i, nnz=int(gr.nnz), d=d, K=K, single=single, ks=ks, batch=m.batch_size,
                    prefilter=m.score_prefilter)
        e0 = np.concatenate([m.embedding_user.weight.detach().cpu().numpy(), m.embedding_item.weight.detach().cpu().numpy()])
        idx, val = gr.to_coo()
        rep, _ = oracle.propagate(idx, val, e0, K, single=single)
        users = rng.permutation(n_u)[:max(1, int(n_u * rng.random()))]
        s = oracle.score_dense(rep[:n_u][users], rep[n_u:])
        rp, items = oracle.train_mask_csr(u, i, users)
        oracle.mask_train(s, rp, items)
        want_v, want_i = oracle.topk(s, max(ks), round4=True)
        pred, scores = m.predict(users, with_scores=True)
        pred, scores = np.asarray(pred), np.asarray(scores, dtype=np.float32)
        fin = np.isfinite(want_v)
        assert np.array_equal(pred[fin], want_i[fin]), (what, int((pred[fin] != want_i[fin]).sum()))
        assert np.array_equal(bits(scores[fin]), bits(want_v[fin])), what
        # evaluate(): the test users, metrics of the reference's calculate_metrics
        tu = np.sort(test['user_id'].unique())
        s = oracle.score_dense(rep[:n_u][tu], rep[n_u:])
        rp, items = oracle.train_mask_csr(u, i, tu)
        oracle.mask_train(s, rp, items)
        _, ti = oracle.topk(s, max(ks), round4=True)
        want = oracle.metrics(ds.true_test_lil, ti, ks)
        got = m.evaluate()
        for name in ('recall', 'precision', 'hit', 'ndcg', 'f1'):
            assert np.allclose(got[name], want[name], atol=1e-12), (what, name)
