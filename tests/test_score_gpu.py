"""Parity of the HIP scoring kernels (fp32 MFMA dense scores, mask, top-k, pairwise) against the oracle and
the reference's golden vectors."""
import numpy as np
import pytest
import torch

from conftest import bits, normwise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('b,i,d', [(5, 4, 64), (60, 40, 48), (130, 257, 64), (257, 1000, 128), (33, 129, 100),
                                   (128, 128, 256), (1, 70, 6)])
def test_dense_scores_bit_exact_vs_oracle(cuda, oracle, b, i, d):
    from textgcn_amd import scoring
    rng = np.random.default_rng(b * 1000 + i + d)
    u = rng.standard_normal((b, d)).astype(np.float32)
    it = rng.standard_normal((i, d)).astype(np.float32)
    ref = oracle.score_dense(u, it)
    s = scoring.score_dense(torch.from_numpy(u).to(cuda), torch.from_numpy(it).to(cuda))
    assert np.array_equal(bits(s.cpu().numpy()), bits(ref))


def test_dense_scores_layout_asymmetric(cuda):
    """A = I against an asymmetric B catches a transposed C write."""
    from textgcn_amd import scoring
    d = 64
    u = torch.eye(d, device=cuda)
    it = torch.arange(200 * d, device=cuda, dtype=torch.float32).reshape(200, d) / 7.0
    s = scoring.score_dense(u, it)
    assert torch.equal(s, it.t().contiguous())


def test_dense_scores_user_gather(cuda, oracle):
    from textgcn_amd import scoring
    rng = np.random.default_rng(5)
    table = rng.standard_normal((500, 64)).astype(np.float32)
    it = rng.standard_normal((300, 64)).astype(np.float32)
    ids = rng.permutation(500)[:200].astype(np.int64)
    ref = oracle.score_dense(table[ids], it)
    s = scoring.score_dense(torch.from_numpy(table).to(cuda), torch.from_numpy(it).to(cuda),
                            user_ids=torch.from_numpy(ids).to(cuda))
    assert np.array_equal(bits(s.cpu().numpy()), bits(ref))


@pytest.mark.parametrize('name', ['a', 'k4d128', 'd48'])
def test_scores_mask_topk_vs_reference(golden, cuda, name):
    """G2: rating, masked rating and top-10 (+ round) equal the reference's own outputs."""
    from textgcn_amd import scoring
    from textgcn_amd.graph import train_mask_csr
    g = golden('g2_synth60')
    n_u = int(g['n_users'])
    ue = torch.from_numpy(g[f'{name}_users_emb']).to(cuda)
    ie = torch.from_numpy(g[f'{name}_items_emb']).to(cuda)
    s = scoring.score_dense(ue, ie)
    assert normwise(s.cpu().numpy(), g[f'{name}_rating']) <= 1e-6
    rp, items = train_mask_csr(g['train_u'], g['train_i'], n_u)
    scoring.mask_train(s, torch.from_numpy(rp.astype(np.int32)).to(cuda), torch.from_numpy(items).to(cuda))
    sm = s.cpu().numpy()
    assert np.array_equal(np.isneginf(sm), np.isneginf(g[f'{name}_masked']))
    v, i = scoring.topk(s, 10, round4=True)
    assert np.array_equal(i.cpu().numpy(), g[f'{name}_topk_idx'])
    assert np.array_equal(bits(v.cpu().numpy()), bits(g[f'{name}_topk_val']))


def test_dummy_predictions_vs_reference(golden, cuda):
    """G1 (data/dummy, k=3 of 4 items): finite prefix identical, -inf tail made of masked train items."""
    from textgcn_amd import scoring
    from textgcn_amd.graph import train_mask_csr
    g = golden('g1_dummy')
    s = scoring.score_dense(torch.from_numpy(g['users_emb']).to(cuda), torch.from_numpy(g['items_emb']).to(cuda))
    assert np.array_equal(bits(s.cpu().numpy()), bits(g['rating']))
    rp, items = train_mask_csr(g['train_u'], g['train_i'], 5)
    scoring.mask_train(s, torch.from_numpy(rp.astype(np.int32)).to(cuda), torch.from_numpy(items).to(cuda))
    assert np.array_equal(bits(s.cpu().numpy()), bits(g['masked']))
    v, i = scoring.topk(s, 3, round4=True)
    v, i = v.cpu().numpy(), i.cpu().numpy()
    for b in range(5):
        fin = np.isfinite(g['topk_val'][b])
        assert np.array_equal(i[b][fin], g['topk_idx'][b][fin])
        assert np.array_equal(bits(v[b][fin]), bits(g['topk_val'][b][fin]))
        masked = set(np.nonzero(np.isneginf(g['masked'][b]))[0])
        assert set(i[b][~fin]) <= masked and np.all(np.isneginf(v[b][~fin]))


@pytest.mark.parametrize('b,i,k', [(7, 64, 64), (3, 65, 1), (40, 1000, 40), (5, 50000, 40), (2, 5, 5), (9, 300, 20)])
def test_topk_vs_oracle_with_ties_and_inf(cuda, oracle, b, i, k):
    from textgcn_amd import scoring
    rng = np.random.default_rng(b + i + k)
    # few distinct values -> many ties; sprinkle -inf
    s = rng.integers(0, 50, size=(b, i)).astype(np.float32) / 8.0
    s[rng.random((b, i)) < 0.1] = -np.inf
    if b > 1:
        s[1, :] = -np.inf          # a fully masked row
    rv, ri = oracle.topk(s, k, round4=False)
    v, idx = scoring.topk(torch.from_numpy(s).to(cuda), k)
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(bits(v.cpu().numpy()), bits(rv))


def test_topk_strided_rows_and_round(cuda, oracle):
    from textgcn_amd import scoring
    rng = np.random.default_rng(0)
    big = rng.standard_normal((6, 1003)).astype(np.float32)
    t = torch.from_numpy(big).to(cuda)[:, :999]            # row stride 1003, unaligned rows
    rv, ri = oracle.topk(big[:, :999], 40, round4=True)
    v, idx = scoring.topk(t, 40, round4=True)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(bits(v.cpu().numpy()), bits(rv))


def test_topk_rejects_bad_k(cuda):
    from textgcn_amd import scoring
    s = torch.zeros((2, 10), device=cuda)
    with pytest.raises(RuntimeError):
        scoring.topk(s, 0)
    with pytest.raises(RuntimeError):
        scoring.topk(s, 11)
    with pytest.raises(ValueError):
        scoring.topk(torch.zeros((2, 100), device=cuda), 101)         # k > 64 takes several passes, but never k > I
    v, i = scoring.topk(torch.arange(200, dtype=torch.float32, device=cuda).reshape(2, 100), 65)
    assert i[0].tolist() == list(range(99, 34, -1)) and v[1, 0].item() == 199.0


def test_pairwise_bit_exact_vs_oracle(cuda, oracle):
    from textgcn_amd import scoring
    rng = np.random.default_rng(2)
    ut = rng.standard_normal((100, 64)).astype(np.float32)
    vt = rng.standard_normal((80, 64)).astype(np.float32)
    users = rng.integers(0, 100, 500).astype(np.int64)
    items = rng.integers(0, 80, 500).astype(np.int64)
    ref = oracle.score_pairwise(ut[users], vt[items])
    out = scoring.score_pairwise(torch.from_numpy(ut).to(cuda), torch.from_numpy(vt).to(cuda),
                                 torch.from_numpy(users).to(cuda), torch.from_numpy(items).to(cuda))
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))


def test_full_size_c2_batch_properties(cuda, oracle):
    """One B=2048 batch against I=50k items (config 2 shape): sampled entries vs the oracle; top-40 of the
    masked scores equals a torch sort of the same device matrix wherever values are distinct."""
    from textgcn_amd import scoring
    rng = np.random.default_rng(9)
    b, i, d, k = 2048, 50000, 64, 40
    u = (rng.standard_normal((b, d)) * 0.1).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    ud, itd = torch.from_numpy(u).to(cuda), torch.from_numpy(it).to(cuda)
    s = scoring.score_dense(ud, itd)
    rows = rng.integers(0, b, 64)
    cols = rng.integers(0, i, 64)
    ref = oracle.score_pairwise(u[rows], it[cols])
    assert np.array_equal(bits(s[torch.from_numpy(rows).to(cuda), torch.from_numpy(cols).to(cuda)].cpu().numpy()), bits(ref))
    # mask: 50 random items per user
    mi = np.sort(rng.integers(0, i, size=(b, 50)), axis=1).astype(np.int32)
    rp = (np.arange(b + 1) * 50).astype(np.int32)
    scoring.mask_train(s, torch.from_numpy(rp).to(cuda), torch.from_numpy(mi.ravel()).to(cuda))
    assert torch.isneginf(s[0, int(mi[0, 0])]) and int(torch.isneginf(s).sum()) == len(np.unique(np.arange(b)[:, None] * i + mi))
    v, idx = scoring.topk(s, k)
    tv, ti = torch.sort(s, dim=1, descending=True, stable=True)   # stable: ties keep ascending index
    assert torch.equal(v, tv[:, :k]) and torch.equal(idx, ti[:, :k])


# ------------------------------------------------------------------------------------------------ fused path
def _fused_vs_dense(cuda, u, it, k, mask=None, user_ids=None, round4=True):
    """tgcn_score_topk_f32 must equal tgcn_score_dense_f32 -> tgcn_mask_f32 -> tgcn_topk_f32 bit for bit."""
    from textgcn_amd import scoring
    ud, itd = torch.from_numpy(u).to(cuda), torch.from_numpy(it).to(cuda)
    ids = None if user_ids is None else torch.from_numpy(user_ids).to(cuda)
    rp = it_ = None
    if mask is not None:
        rp = torch.from_numpy(mask[0].astype(np.int32)).to(cuda)
        it_ = torch.from_numpy(mask[1].astype(np.int32)).to(cuda)
    s = scoring.score_dense(ud, itd, user_ids=ids)
    if mask is not None:
        scoring.mask_train(s, rp, it_)
    rv, ri = scoring.topk(s, k, round4=round4)
    v, i = scoring.score_topk(ud, itd, k, user_ids=ids, mask_rowptr=rp, mask_items=it_, round4=round4)
    torch.cuda.synchronize()
    assert torch.equal(i, ri), (i != ri).sum().item()
    assert np.array_equal(bits(v.cpu().numpy()), bits(rv.cpu().numpy()))
    # the bf16-prefiltered entry point (tgcn_score_topk_prefilter_f32): the same bits, with the item operand packed inside the
    # call and handed in
    for pack in (None, scoring.item_pack(itd)):
        pv, pi = scoring.score_topk(ud, itd, k, user_ids=ids, mask_rowptr=rp, mask_items=it_, round4=round4, prefilter=True,
                                    item_pack=pack)
        torch.cuda.synchronize()
        assert torch.equal(pi, ri), ('prefilter', (pi != ri).sum().item())
        assert np.array_equal(bits(pv.cpu().numpy()), bits(rv.cpu().numpy()))
    return v, i


def _rand_mask(rng, b, n_items, lo, hi):
    cnt = rng.integers(lo, hi + 1, size=b)
    rp = np.zeros(b + 1, dtype=np.int64)
    np.cumsum(cnt, out=rp[1:])
    items = np.concatenate([np.sort(rng.choice(n_items, size=c, replace=False)) for c in cnt]) if rp[-1] else np.zeros(0, np.int64)
    return rp, items


@pytest.mark.parametrize('b,i,d,k', [(300, 20000, 64, 40), (257, 9000, 128, 20), (130, 12345, 256, 64), (64, 10000, 100, 1),
                                     (2048, 50000, 64, 40), (5, 8193, 48, 7), (130, 9000, 50, 10), (70, 20011, 101, 33),
                                     (513, 16400, 127, 40)])
def test_fused_topk_equals_dense_path(cuda, b, i, d, k):
    rng = np.random.default_rng(b + i + d + k)
    u = (rng.standard_normal((b, d)) * 0.1).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    _fused_vs_dense(cuda, u, it, k, mask=_rand_mask(rng, b, i, 0, 120))
    _fused_vs_dense(cuda, u, it, k, mask=None, round4=False)


def test_fused_topk_small_catalogue_and_gather(cuda):
    """I <= 8192 takes the dense route inside the same entry point; user_ids gather."""
    rng = np.random.default_rng(1)
    table = rng.standard_normal((700, 64)).astype(np.float32)
    it = rng.standard_normal((4000, 64)).astype(np.float32)
    ids = rng.permutation(700)[:333].astype(np.int64)
    _fused_vs_dense(cuda, table, it, 40, mask=_rand_mask(rng, 333, 4000, 1, 50), user_ids=ids)
    it2 = rng.standard_normal((30000, 64)).astype(np.float32)
    _fused_vs_dense(cuda, table, it2, 40, mask=_rand_mask(rng, 333, 30000, 1, 50), user_ids=ids)


@pytest.mark.parametrize('d', [64, 128, 40, 200, 320, 960])
def test_fused_topk_hard_cases(cuda, d):
    """Cases built to defeat the threshold estimate: all-equal scores (no candidate passes a strict bar), scores
    increasing with the item id, a user whose train list covers most of the catalogue (log overflow / fewer
    than k unmasked items), duplicated item rows (ties broken by index)."""
    rng = np.random.default_rng(2)
    b, i, k = 200, 20000, 40
    u = np.abs(rng.standard_normal((b, d))).astype(np.float32) * 0.1
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    # (a) every item identical -> every score equal per user
    same = np.repeat(it[:1], i, axis=0)
    _fused_vs_dense(cuda, u, same, k, mask=_rand_mask(rng, b, i, 0, 30))
    # (b) scores strictly increasing with the item id for every user (all-positive users x a ramp)
    ramp = (np.arange(i, dtype=np.float32)[:, None] / i) * np.ones((1, d), dtype=np.float32)
    _fused_vs_dense(cuda, u, ramp, k, mask=_rand_mask(rng, b, i, 0, 30))
    # (c) heavy users: user 0 has all but 10 items masked, user 1 all but 60, user 2 has 5000 train items
    cnt = rng.integers(0, 40, size=b)
    cnt[0], cnt[1], cnt[2] = i - 10, i - 60, 5000
    rp = np.zeros(b + 1, dtype=np.int64)
    np.cumsum(cnt, out=rp[1:])
    items = np.concatenate([np.sort(rng.choice(i, size=c, replace=False)) for c in cnt])
    _fused_vs_dense(cuda, u, it, k, mask=(rp, items))
    # (d) duplicated item rows: exact ties decided by the smaller index
    dup = it.copy()
    dup[1::2] = dup[0::2]
    _fused_vs_dense(cuda, u, dup, k, mask=_rand_mask(rng, b, i, 0, 30))


def test_fused_topk_vs_oracle_sample(cuda, oracle):
    """independent check against the CPU oracle on a slice (the dense path is itself pinned to the oracle above)"""
    from textgcn_amd import scoring
    rng = np.random.default_rng(4)
    b, i, d, k = 40, 9000, 64, 40
    u = (rng.standard_normal((b, d)) * 0.1).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    rp, items = _rand_mask(rng, b, i, 5, 60)
    s = oracle.score_dense(u, it)
    oracle.mask_train(s, rp, items)
    rv, ri = oracle.topk(s, k, round4=True)
    for prefilter in (False, True):        # both entry points directly against the oracle
        v, idx = scoring.score_topk(torch.from_numpy(u).to(cuda), torch.from_numpy(it).to(cuda), k,
                                    mask_rowptr=torch.from_numpy(rp.astype(np.int32)).to(cuda),
                                    mask_items=torch.from_numpy(items.astype(np.int32)).to(cuda), round4=True, prefilter=prefilter)
        assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(bits(v.cpu().numpy()), bits(rv)), prefilter


@pytest.mark.parametrize('b,i,d,k', [(70, 20000, 64, 100), (33, 9000, 128, 150), (9, 300, 64, 130), (5, 200, 48, 65)])
def test_topk_beyond_64_takes_several_passes(cuda, b, i, d, k):
    """k above the kernels' 64-entry list: ceil(k/64) passes with the items already taken masked out -- fused and
    dense entry points against torch.topk on the masked score matrix (values everywhere; item ids where scores are
    finite and untied)."""
    from textgcn_amd import scoring
    rng = np.random.default_rng(b + i + k)
    u = torch.from_numpy((rng.standard_normal((b, d)) * 0.1).astype(np.float32)).to(cuda)
    it = torch.from_numpy((rng.standard_normal((i, d)) * 0.1).astype(np.float32)).to(cuda)
    rp, items = _rand_mask(rng, b, i, 0, 60)
    rpd, itd = torch.from_numpy(rp.astype(np.int32)).to(cuda), torch.from_numpy(items.astype(np.int32)).to(cuda)
    s = scoring.score_dense(u, it)
    scoring.mask_train(s, rpd, itd)
    rv, ri = torch.topk(s, k, dim=1)
    for v, idx in (scoring.score_topk(u, it, k, mask_rowptr=rpd, mask_items=itd), scoring.topk(s, k)):
        assert torch.equal(v, rv)
        fin = torch.isfinite(rv)
        untied = fin.clone()
        untied[:, 1:] &= rv[:, 1:] != rv[:, :-1]
        untied[:, :-1] &= rv[:, :-1] != rv[:, 1:]
        assert torch.equal(idx[untied], ri[untied])
        got = torch.gather(s, 1, idx)
        assert torch.equal(got[fin], rv[fin])                       # every listed item really has that score
        for r in range(b):
            f = idx[r][fin[r]]
            assert f.unique().numel() == f.numel()                  # no item twice among the finite ones


def _oracle_topk_rows(oracle, u, it, rp, items, rows, k):
    """oracle top-k (k-ordered fmaf chains, mask, (value desc, index asc), round) of the selected batch rows"""
    s = oracle.score_dense(u[rows], it)
    sub_rp = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(rp[rows + 1] - rp[rows], out=sub_rp[1:])
    sub_items = np.concatenate([items[rp[r]:rp[r + 1]] for r in rows]) if sub_rp[-1] else np.zeros(0, np.int64)
    oracle.mask_train(s, sub_rp, sub_items)
    return oracle.topk(s, k, round4=True)


def test_fused_topk_at_config3_scoring_shape(cuda, oracle):
    """BASELINE config 3's full-catalogue call as the model class issues it: 16 384 users x 60 000 items x d = 128
    (k_score_filter32 + k_tau + k_select + fallback) against the unfused dense -> mask -> top-k path on the same device
    buffers (bit for bit, all users) and against the CPU oracle on sampled users."""
    rng = np.random.default_rng(33)
    b, i, d, k = 16384, 60000, 128, 40
    u = (rng.standard_normal((b, d)) * 0.1).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    lists = [np.unique(rng.integers(0, i, size=rng.integers(1, 30))) for _ in range(b)]     # sorted, distinct train items
    rp = np.zeros(b + 1, dtype=np.int64)
    np.cumsum([len(t) for t in lists], out=rp[1:])
    items = np.concatenate(lists)
    v, idx = _fused_vs_dense(cuda, u, it, k, mask=(rp, items))      # (runs the prefiltered entry point too, same bits required)
    rows = rng.choice(b, size=48, replace=False)
    rv, ri = _oracle_topk_rows(oracle, u, it, rp, items, rows, k)
    assert np.array_equal(idx.cpu().numpy()[rows], ri)
    assert np.array_equal(bits(v.cpu().numpy()[rows]), bits(rv))


def test_scoring_at_config5_folded_width(cuda, oracle):
    """BASELINE config 5's scoring call: the folded ltr_linear operands are 960 wide (d = 128 + 2 x 384 + bias column,
    padded), 2048 users x 60 000 items.  Widths beyond 256 take the dense -> mask -> top-k route inside
    tgcn_score_topk_f32: checked against the separate calls (bit for bit) and against the CPU oracle's k-ordered chain on
    sampled users."""
    rng = np.random.default_rng(55)
    b, i, d, k = 2048, 60000, 960, 40
    u = (rng.standard_normal((b, d)) * 0.05).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.05).astype(np.float32)
    rp, items = _rand_mask(rng, b, i, 1, 30)
    v, idx = _fused_vs_dense(cuda, u, it, k, mask=(rp, items))
    rows = rng.choice(b, size=12, replace=False)
    rv, ri = _oracle_topk_rows(oracle, u, it, rp, items, rows, k)
    assert np.array_equal(idx.cpu().numpy()[rows], ri)
    assert np.array_equal(bits(v.cpu().numpy()[rows]), bits(rv))


@pytest.mark.parametrize('b,i,d', [(300, 200_000, 64), (48, 1_000_000, 64), (48, 1_700_000, 64), (40, 2_000_000, 64), (33, 2_200_000, 64),
                                   (40, 2_150_000, 128), (64, 140_000, 256), (4200, 140_001, 128), (4130, 131_073, 50), (260, 131_137, 64)])
def test_fused_topk_very_large_catalogue(cuda, b, i, d):
    """Large catalogues (config 4 has 2 M items).  Narrow rows keep the in-kernel sample (two values per user and block of 128 sampled
    items reach k_tau) at the density whose bitmap row fits k_sample_bits' LDS: every 8th item up to 1.0 M items, every 16th up to
    2.1 M (200 000; 1 000 000 on the boundary; 1 700 000; config 4's 2 000 000), every 32nd beyond (2 200 000; 2 150 000 x 128).  Wide
    rows beyond 131 072 items take the threshold from the mask + workgroup-per-row top-k on a stride-32 sample of all scores instead
    of k_tau, and the fallback bookkeeping is cleared by a memset (140 000 x 256).  From 131 072 items on the narrow filter also keeps
    its stage summary and k_rescore reads the flagged stages' words only: both stage sizes (d <= 64: 256 items, d <= 128: 128), both
    consumers (calls up to 4096 users select inside k_rescore; 4200 / 4130 users take the separate selection), catalogues ending inside
    a stage and inside a 64-item unit.  Same contract everywhere: equal to the dense path."""
    rng = np.random.default_rng(77)
    k = 40
    u = (rng.standard_normal((b, d)) * 0.1).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    _fused_vs_dense(cuda, u, it, k, mask=_rand_mask(rng, b, i, 0, 60))
    # a user with nearly everything masked -> flagged -> exact fallback with the last-arriver merge on this path too
    cnt = rng.integers(0, 40, size=b)
    cnt[0] = i - 25
    rp = np.zeros(b + 1, dtype=np.int64)
    np.cumsum(cnt, out=rp[1:])
    items = np.concatenate([np.sort(rng.choice(i, size=c, replace=False)) for c in cnt])
    _fused_vs_dense(cuda, u, it, k, mask=(rp, items))


@pytest.mark.parametrize('d', [64, 128, 32, 960, 1024])
def test_prefilter_bound_under_worst_case_rounding(cuda, d):
    """Data built against the bf16 bound: every element sits just below a bf16 rounding boundary with the signs aligned, so the
    approximate score of the planted winners is lower than their fp32 score by nearly the whole bound (2^-7 |u| |i|) while
    thousands of other items sit just under the threshold.  The prefiltered call must still return the dense path's bits."""
    rng = np.random.default_rng(d)
    b, i, k = 160, 20000, 40
    # elements (1 + 2^-8 - 2^-20) * 2^e: bf16 rounds them DOWN by a relative 2^-8 (the worst case of round-to-nearest)
    mant = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20)
    u = (mant * np.exp2(rng.integers(-4, 0, size=(b, d)))).astype(np.float32)
    it = (mant * np.exp2(rng.integers(-4, 0, size=(i, d)))).astype(np.float32)
    # crowd: most items are an exact-bf16 copy scaled so their scores land a hair under the planted ones
    it[k:] = (np.exp2(rng.integers(-4, 0, size=(i - k, d)))).astype(np.float32) * np.float32(1.0 + 2.0 ** -7)
    _fused_vs_dense(cuda, u, it, k, mask=_rand_mask(rng, b, i, 0, 20), round4=False)
    # mixed signs and magnitudes from denormal to large, a NaN row and an inf row among the items
    u2 = (rng.standard_normal((b, d)) * np.exp2(rng.integers(-30, 10, size=(b, 1)))).astype(np.float32)
    it2 = (rng.standard_normal((i, d)) * np.exp2(rng.integers(-30, 10, size=(i, 1)))).astype(np.float32)
    it2[5] = 1e-42
    it2[7, 0] = np.inf
    it2[9, 1] = np.nan
    _fused_vs_dense(cuda, u2, it2, k, mask=_rand_mask(rng, b, i, 0, 20), round4=False)


@pytest.mark.parametrize('d', [960, 1024, 128, 64])
def test_prefilter_accumulation_budget_under_cancelling_sums(cuda, d):
    """VERDICT r2 (weak, parity): the bf16 pass budgets the matrix pipe's fp32 ACCUMULATION (and the fp32 chain's own rounding) at
    2^-11 |u| |i|.  Operand rounding is taken out of the picture here -- every element is a signed power of two, exact in bf16, so
    the residual terms of the bound are at their floors and the budget is the only slack -- and the accumulation is made as bad as
    the data can make it: the first half of every row's products is large and positive, the second half cancels it, so running
    sums climb to ~d/2 of a product while the scores that decide the top k are a few units of the LAST place of those sums.
    Thousands of near-copies put the k-th score inside a crowd, i.e. on the threshold.  A pipe whose rounding exceeded the budget
    would drop a true top-k pair and the lists would differ from the dense path's."""
    rng = np.random.default_rng(1000 + d)
    b, i, k = 96, 12000, 40
    half = d // 2
    sgn_u = rng.choice([-1.0, 1.0], size=(b, d)).astype(np.float32)
    mag_u = np.exp2(rng.integers(-2, 1, size=(b, d))).astype(np.float32)
    u = sgn_u * mag_u
    # item rows follow ONE user pattern (user 0's signs) so that its products are all positive in the first half and all negative
    # in the second: the running sum peaks at ~ half * E|product|; the other users see random signs (ordinary cancellation)
    base_sign = np.concatenate([sgn_u[0, :half], -sgn_u[0, half:]])
    mag_i = np.exp2(rng.integers(-2, 1, size=(i, d))).astype(np.float32)
    it = (base_sign[None, :] * mag_i).astype(np.float32)
    # make the two halves cancel EXACTLY for user 0 on a crowd of items, then separate them by one small power of two in one column:
    # scores of the crowd are 2^-e with e in a narrow range, against partial sums of ~d/4
    it[:, half:] = -it[:, :half] * (mag_u[0, :half] / mag_u[0, half:2 * half])[None, :] * sgn_u[0, :half][None, :] * sgn_u[0, half:2 * half][None, :]
    # ... then one bf16 unit in the last place on one column per item: user 0's scores are +-2^-9 .. 2^-7, about one part in 2^16 of
    # the running sums they are the remainder of; every element is still exact in bf16
    col = rng.integers(0, half, size=i)
    it[np.arange(i), col] *= np.float32(1.0 + 2.0 ** -7)
    assert np.array_equal(it, torch.from_numpy(it).to(torch.bfloat16).float().numpy())
    _fused_vs_dense(cuda, u.astype(np.float32), it.astype(np.float32), k, mask=_rand_mask(rng, b, i, 0, 30), round4=False)
    # the same rows at a scale where the budget term itself is near the fp32 denormal range and near overflow
    for scale in (2.0 ** -40, 2.0 ** 40):
        _fused_vs_dense(cuda, (u * np.float32(scale)).astype(np.float32), it.astype(np.float32), k, mask=_rand_mask(rng, b, i, 0, 30), round4=False)


def test_item_norms(cuda):
    """the item factors of the bound: the row norm (never below the floored Euclidean norm, within 2^-11 above it) and the norm
    of the row's bf16 rounding residual (same margins; at most 2^-8 of the row norm); an infinite row gives +inf"""
    from textgcn_amd import scoring
    rng = np.random.default_rng(3)
    for n, d in ((50000, 64), (777, 128), (1000, 100), (5, 32), (3, 16)):
        it = (rng.standard_normal((n, d)) * 0.3).astype(np.float32)
        if n > 100:
            it[7, 3] = np.inf
            it[11] = 0.0
            it[13] = 0.375          # exactly representable in bf16: residual 0
        want = np.sqrt((np.maximum(np.abs(it.astype(np.float64)), 2.0 ** -50) ** 2).sum(axis=1))
        t = torch.from_numpy(it)
        resid = (t.double() - t.to(torch.bfloat16).double()).numpy()        # torch's conversion is round-to-nearest-even too
        want_r = np.sqrt((np.maximum(np.abs(resid), 2.0 ** -58) ** 2).sum(axis=1))
        got = scoring.item_norms(t.to(cuda)).cpu().numpy().astype(np.float64)
        fin = np.isfinite(want)
        assert np.all(got[fin, 0] >= want[fin]) and np.all(got[fin, 0] <= want[fin] * (1 + 2.0 ** -11))
        assert np.all(got[fin, 1] >= want_r[fin]) and np.all(got[fin, 1] <= want_r[fin] * (1 + 2.0 ** -11))
        assert np.all(got[fin, 1] <= 2.0 ** -8 * got[fin, 0] * (1 + 2.0 ** -10))
        assert np.all(np.isposinf(got[np.isposinf(want)]))


def test_item_pack(cuda):
    """the packed item operand: row i = the bf16 image of the row (round-to-nearest-even, zeros behind d) and, in the 16-byte chunk
    behind it, [r_i, n_i + r_i, n_i] each rounded UP to bf16 from the factors tgcn_item_norms_f32 reports"""
    from textgcn_amd import scoring
    rng = np.random.default_rng(4)
    for n, d in ((50000, 64), (777, 128), (1000, 100), (333, 50), (5, 32), (3, 16), (300, 200), (300, 960), (300, 896), (100, 1000)):
        it = (rng.standard_normal((n, d)) * 0.3).astype(np.float32)
        if n > 100:
            it[11] = 0.0
            it[13] = 0.375
        t = torch.from_numpy(it).to(cuda)
        width = 64 if d <= 64 else 128 if d <= 128 else 256 if d <= 256 else 512 if d <= 512 else 1024 if d <= 832 else 896 if d <= 896 else 960 if d <= 960 else 1024
        pack = scoring.item_pack(t).cpu().numpy().reshape(n, 2 * width + 16)
        rows = pack[:, :2 * width].copy().view(np.uint16)
        want = np.zeros((n, width), dtype=np.uint16)
        want[:, :d] = torch.from_numpy(it).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
        assert np.array_equal(rows, want)
        chunk = pack[:, 2 * width:].copy().view(np.uint16)
        assert not chunk[:, 3:].any()
        f = (chunk[:, :3].astype(np.uint32) << 16).view(np.float32).astype(np.float64)      # r, n + r, n as bf16 values
        norms = scoring.item_norms(t).cpu().numpy().astype(np.float64)
        for got, ref in ((f[:, 0], norms[:, 1]), (f[:, 1], norms[:, 0] + norms[:, 1]), (f[:, 2], norms[:, 0])):
            assert np.all(got >= ref) and np.all(got <= ref * (1 + 2.0 ** -7) * (1 + 2.0 ** -19))
    assert scoring.item_pack(torch.zeros(10, 2048, device=cuda)) is None and scoring.item_pack(torch.zeros(10, 132, device=cuda)) is None


def test_prefilter_outlier_item_does_not_flood_the_lists(cuda):
    """The bound is per (user, item) pair: one item row of enormous norm (here 1000x the rest, pointing away from every user)
    becomes everybody's candidate but must not hand the users to the exact fallback, and the lists stay the dense path's."""
    from textgcn_amd import scoring
    rng = np.random.default_rng(8)
    b, i, d, k = 512, 30000, 64, 40
    u = np.abs(rng.standard_normal((b, d)) * 0.1).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    it[12345] = -100.0
    _fused_vs_dense(cuda, u, it, k, mask=_rand_mask(rng, b, i, 0, 20), round4=False)
    ud, itd = torch.from_numpy(u).to(cuda), torch.from_numpy(it).to(cuda)
    scoring.score_topk(ud, itd, k, prefilter=True, slot=7)
    assert scoring.fallback_count(cuda, b, i, d, k, slot=7) <= 2


def test_prefilter_wide_rows_identical(cuda):
    """Rows wider than 128 (the folded ltr_linear operands and every pack width of k_score_prefilter_wide: 256, 512, 896, 960,
    1024 elements; d % 8 == 0) -- both entry points bit for bit, ragged user tiles, partial last units, masks, ties, an outlier row;
    widths the bf16 pass does not take (d % 8 != 0, d > 1024) run the fp32 path behind the same entry point."""
    from textgcn_amd import scoring
    rng = np.random.default_rng(77)
    for trial, d in enumerate([136, 256, 264, 512, 520, 840, 896, 960, 1000, 1024, 204, 1032]):
        b = int(rng.integers(1, 400))
        i = int(rng.integers(8193, 14000))
        k = int(rng.integers(1, 65))
        u = (rng.standard_normal((b, d)) * 0.1).astype(np.float32)
        it = (rng.standard_normal((i, d)) * 0.1 * np.exp(rng.normal(0, 1.0, size=(i, 1)))).astype(np.float32)
        if trial % 3 == 1:
            it[1::3] = it[0::3][: len(it[1::3])]
        if trial % 4 == 2:
            it[17] = -50.0
        assert (scoring.item_pack(torch.from_numpy(it[:4]).to(cuda)) is None) == (d % 8 != 0 or d > 1024)
        _fused_vs_dense(cuda, u, it, k, mask=_rand_mask(rng, b, i, 0, 60), round4=bool(trial % 2))
    # k_refine's corners at K = 960: train lists longer than its LDS cache that hold the users' best items (the second threshold
    # must be formed from what is left), and a catalogue of near-copies (hundreds of candidates inside the error band of the k-th)
    b, i, d, k = 96, 9000, 960, 64
    u = (rng.standard_normal((b, d)) * 0.1).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    best = np.argsort(-(u @ it.T), axis=1)[:, :700]                       # every user's 700 best items are train items
    rows = [np.unique(np.concatenate([best[r], rng.integers(0, i, 300)])).astype(np.int32) for r in range(b)]
    rp = np.zeros(b + 1, dtype=np.int64)
    np.cumsum([len(r) for r in rows], out=rp[1:])
    _fused_vs_dense(cuda, u, it, k, mask=(rp, np.concatenate(rows)), round4=False)
    it2 = np.repeat(it[:30], 300, axis=0) * (1.0 + 2.0 ** -12 * rng.integers(0, 3, size=(9000, 1))).astype(np.float32)
    _fused_vs_dense(cuda, u, it2.astype(np.float32), k, mask=_rand_mask(rng, b, i, 0, 60), round4=False)


@pytest.mark.parametrize('b,i,d', [(33000, 8200, 264), (700, 16500, 960), (4100, 9000, 512), (6200, 16500, 264), (12300, 8200, 264)])
def test_prefilter_wide_split_counts(cuda, b, i, d):
    """The wide filter's item splits come from generations x (prologue + units per split) (tgcn_score_fused.hip make_plan): 1-2 splits
    for hundreds of user tiles, 16 for a handful, with the per-segment log capacity scaled to match (4 segments per split and user,
    4096 log entries per user in all); its (split, tile) pairs are dealt to the XCDs in groups of 32 that may straddle two splits.
    Both ends of that range, an odd tile count, and tile counts that are no multiple of 32 (49, 97: the last group of a split is
    shared with the next split's first tiles) against the fp32 path bit for bit."""
    from textgcn_amd import scoring
    rng = np.random.default_rng(b + d)
    u = torch.from_numpy((rng.standard_normal((b, d)) * 0.1).astype(np.float32)).to(cuda)
    it = torch.from_numpy((rng.standard_normal((i, d)) * 0.1).astype(np.float32)).to(cuda)
    rp, items = _rand_mask(rng, b, i, 0, 12)
    rpd, imd = torch.from_numpy(rp.astype(np.int32)).to(cuda), torch.from_numpy(items.astype(np.int32)).to(cuda)
    rv, ri = scoring.score_topk(u, it, 40, mask_rowptr=rpd, mask_items=imd, round4=False)
    pv, pi = scoring.score_topk(u, it, 40, mask_rowptr=rpd, mask_items=imd, round4=False, prefilter=True, slot=5)
    torch.cuda.synchronize()
    assert torch.equal(pi, ri) and np.array_equal(bits(pv.cpu().numpy()), bits(rv.cpu().numpy()))
    st = scoring.call_stats(cuda, b, i, d, 40, True, slot=5)
    assert st['fallback_users'] <= max(2, b // 2000), st
    assert b * 40 <= st['kept_pairs'] <= st['rescored_pairs'] <= st['logged_pairs'], st


@pytest.mark.parametrize('b,i,d', [(2048, 50000, 64), (2048, 60000, 128)])
def test_prefilter_keeps_the_fallback_rare(cuda, b, i, d):
    """The exact fallback hides a filter that loses or floods candidates (the results stay right, the call gets slow): on
    Gaussian embeddings the bf16-candidate path must hand no more users to it than a handful per call."""
    from textgcn_amd import scoring
    g = torch.Generator().manual_seed(d)
    for _ in range(3):
        ue = (torch.randn(b, d, generator=g) * 0.1).to(cuda)
        ie = (torch.randn(i, d, generator=g) * 0.1).to(cuda)
        counts = {}
        for mode in (False, True):
            scoring.score_topk(ue, ie, 40, prefilter=mode, slot=7)
            counts[mode] = scoring.fallback_count(cuda, b, i, d, 40, slot=7)
        assert counts[False] <= 2 and counts[True] <= 4, counts


def test_prefilter_identical_over_random_shapes(cuda):
    """A sweep of shapes, widths, k, mask sizes and value scales (seeded): both entry points against each other, bit for bit --
    heavy-tailed norms, sparse rows and near-ties included."""
    from textgcn_amd import scoring
    rng = np.random.default_rng(2024)
    for trial in range(14):
        b = int(rng.integers(1, 700))
        i = int(rng.integers(8193, 40000))
        d = int(rng.choice([8, 24, 32, 50, 64, 64, 96, 100, 128, 128]))
        k = int(rng.integers(1, 65))
        scale_u = np.exp(rng.normal(0, 1.5, size=(b, 1))).astype(np.float32)        # log-normal row scales: heavy-tailed norms
        scale_i = np.exp(rng.normal(0, 1.5, size=(i, 1))).astype(np.float32)
        u = (rng.standard_normal((b, d)) * scale_u * 0.1).astype(np.float32)
        it = (rng.standard_normal((i, d)) * scale_i * 0.1).astype(np.float32)
        if trial % 3 == 0:
            it[rng.random((i, d)) < 0.7] = 0.0                                        # sparse rows
        if trial % 4 == 1:
            it[1::3] = it[0::3][: len(it[1::3])]                                      # exact ties
        rp, items = _rand_mask(rng, b, i, 0, int(rng.integers(1, 200)))
        ud, itd = torch.from_numpy(u).to(cuda), torch.from_numpy(it).to(cuda)
        rpd, imd = torch.from_numpy(rp.astype(np.int32)).to(cuda), torch.from_numpy(items.astype(np.int32)).to(cuda)
        rv, ri = scoring.score_topk(ud, itd, k, mask_rowptr=rpd, mask_items=imd, round4=False)
        pv, pi = scoring.score_topk(ud, itd, k, mask_rowptr=rpd, mask_items=imd, round4=False, prefilter=True)
        torch.cuda.synchronize()
        assert torch.equal(pi, ri), (trial, b, i, d, k, int((pi != ri).sum()))
        assert np.array_equal(bits(pv.cpu().numpy()), bits(rv.cpu().numpy())), (trial, b, i, d, k)


def test_rows_with_fewer_scores_than_k_do_not_become_positions(cuda):
    """A row with fewer than k non-NaN scores leaves NO_ITEM (2^31 - 1) in the unfilled list positions.  The multi-pass wrappers for
    k > 64 retire a pass's ids before the next pass; used raw as positions those entries are out of range (a torch scatter with them
    took the GPU down in a random sweep: 65 items, k = 65, one NaN item, a user row with -inf).  They must be retired as duplicates."""
    from textgcn_amd import scoring
    rng = np.random.default_rng(3)
    b, i, d, k = 40, 65, 6, 65
    u = rng.standard_normal((b, d)).astype(np.float32)
    it = rng.standard_normal((i, d)).astype(np.float32)
    it[7, 2] = np.nan          # one NaN score in every row
    it[9, 1], it[9, 3] = np.inf, 1.0
    u[5, 1], u[5, 3] = 1.0, -np.inf          # this row: +-inf everywhere, NaN against item 7 and (inf - inf) item 9 -> 63 scores for k = 65
    ud, itd = torch.from_numpy(u).to(cuda), torch.from_numpy(it).to(cuda)
    s = scoring.score_dense(ud, itd)
    assert int(torch.isnan(s[5]).sum()) == 2 and int(torch.isnan(s[0]).sum()) == 1
    v, idx = scoring.topk(s, k)
    torch.cuda.synchronize()
    n_scores = (~torch.isnan(s)).sum(dim=1)
    assert int(n_scores[5]) == 63 and int(n_scores[0]) == 64
    assert int((idx == scoring.NO_ITEM).sum()) > 0          # the case does leave unfilled positions
    for row in range(b):         # the first n_scores positions: every non-NaN item exactly once (what follows them is unspecified)
        n = int(n_scores[row])
        assert sorted(idx[row, :n].tolist()) == [j for j in range(i) if not bool(torch.isnan(s[row, j]))]
    for pre in (False, True):    # the fused entry points' multi-pass wrapper: the same lists on the filled positions
        fv, fi = scoring.score_topk(ud, itd, k, prefilter=pre)
        torch.cuda.synchronize()
        for row in range(b):
            n = int(n_scores[row])
            assert torch.equal(fi[row, :n], idx[row, :n]) and torch.equal(fv[row, :n].view(torch.int32), v[row, :n].view(torch.int32))


@pytest.mark.parametrize('d', [64, 128, 40])
def test_prefilter_huge_norms_and_threshold_signs(cuda, d):
    """Rows at the edges of the fp32 range through both fused entry points (written in round 4 for a form of the narrow filter that
    read its verdict from the accumulator's sign -- measured no faster and not kept, profiles/r04_experiments.md -- and kept as data
    the bound must survive): items and users of norm 2^59, 2^61, 2^100, with inf / NaN elements, zero rows, users whose every
    score is negative (tau < 0), users with tiny scores (|tau| ~ 2^-40), a user with every item masked but 25 (tau = -inf): both
    entry points must return the dense path's lists, bit for bit."""
    rng = np.random.default_rng(77 + d)
    b, i, k = 300, 9000, 40
    u = (rng.standard_normal((b, d)) * 0.1).astype(np.float32)
    it = (rng.standard_normal((i, d)) * 0.1).astype(np.float32)
    it[5] *= np.float32(2.0 ** 59)
    it[6] *= np.float32(2.0 ** 62)
    it[7] *= np.float32(2.0 ** 100)
    it[8, 1] = np.inf
    it[9, d - 1] = -np.inf
    it[10, 0] = np.nan
    it[11] = 0.0
    it[12] = np.float32(2.0 ** 63)
    u[3] *= np.float32(2.0 ** 58)
    u[4] *= np.float32(2.0 ** 61)
    u[5, 2] = np.inf
    u[6, 0] = np.nan
    u[7] = 0.0
    u[8] = -np.abs(it[100:4000]).mean(axis=0) * 3          # most scores negative
    u[9] = u[9] * np.float32(2.0 ** -40)                   # tiny scores
    u[10] = np.float32(2.0 ** -70)
    u[11] *= np.float32(2.0 ** 40)
    rp, items = _rand_mask(rng, b, i, 0, 30)
    # user 12: all but 25 items masked (fewer than k unmasked: tau = -inf, the lists end in masked -inf entries)
    cnt = np.diff(rp)
    keep25 = np.sort(rng.choice(i, size=25, replace=False))
    row12 = np.setdiff1d(np.arange(i), keep25)
    items = np.concatenate([items[:rp[12]], row12, items[rp[13]:]])
    cnt[12] = len(row12)
    rp = np.concatenate([[0], np.cumsum(cnt)])
    _fused_vs_dense(cuda, u, it, k, mask=(rp, items), round4=False)
    _fused_vs_dense(cuda, u, it, k, mask=None, round4=True)
