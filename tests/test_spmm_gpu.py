"""Parity of the HIP SpMM / propagation path against the CPU oracle and the reference's golden vectors.
Everything here calls through the C ABI (textgcn_amd._capi -> libtgcn.so)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import bits, normwise

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))


def _graph(golden, name, prefix=''):
    from textgcn_amd.graph import NormGraph
    g = golden(name)
    return NormGraph.from_pairs(g[prefix + 'train_u'], g[prefix + 'train_i'], int(g[prefix + 'n_users']), int(g[prefix + 'n_items']))


def _random_graph(n_u, n_i, nnz, seed, zipf=0.8):
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    u, i = synth.interactions(n_u, n_i, nnz, seed=seed, zipf=zipf)
    return NormGraph.from_pairs(u, i, n_u, n_i)


VARIANTS = [(0, 0), (1, 4), (1, 8), (1, 16)]  # (variant, row gathers in flight)


@pytest.mark.parametrize('variant,unroll', VARIANTS)
def test_dummy_layers_bit_exact_vs_reference(golden, cuda, variant, unroll):
    """G1: every layer and the combined representation equal the reference's CPU output bit for bit."""
    from textgcn_amd.propagate import Propagator
    g = golden('g1_dummy')
    gr = _graph(golden, 'g1_dummy')
    prop = Propagator(gr, cuda)
    e0 = torch.from_numpy(np.concatenate([g['emb_user'], g['emb_item']])).to(cuda)
    out, layers = prop.forward(e0, 3, exact=True, keep_layers=True, variant=variant, unroll=unroll)
    for k in range(4):
        assert np.array_equal(bits(layers[k].cpu().numpy()), bits(g[f'layer{k}'])), f'layer {k}'
    full = np.concatenate([g['users_emb'], g['items_emb']])
    assert np.array_equal(bits(out.cpu().numpy()), bits(full))
    out2 = prop.forward(e0, 3, exact=True, variant=variant, unroll=unroll)   # fused path without stored layers
    assert np.array_equal(bits(out2.cpu().numpy()), bits(full))


@pytest.mark.parametrize('name', ['a', 'single', 'k4d128', 'd48'])
def test_synth60_variants_bit_exact_vs_reference(golden, cuda, name):
    """G2: d=64/K=3, --single, d=128/K=4 and an odd width (d=48) against the reference's outputs."""
    from textgcn_amd.propagate import Propagator
    g = golden('g2_synth60')
    gr = _graph(golden, 'g2_synth60')
    prop = Propagator(gr, cuda)
    K = int(g[f'{name}_n_layers'])
    e0 = torch.from_numpy(g[f'{name}_layer0']).to(cuda)
    full = np.concatenate([g[f'{name}_users_emb'], g[f'{name}_items_emb']])
    for variant in (0, 1):
        out, layers = prop.forward(e0, K, single=(name == 'single'), exact=True, keep_layers=True, variant=variant)
        for k in range(K + 1):
            assert np.array_equal(bits(layers[k].cpu().numpy()), bits(g[f'{name}_layer{k}'])), (variant, k)
        assert np.array_equal(bits(out.cpu().numpy()), bits(full)), variant
        out2 = prop.forward(e0, K, single=(name == 'single'), exact=True, variant=variant)
        assert np.array_equal(bits(out2.cpu().numpy()), bits(full)), variant


@pytest.mark.parametrize('d,K', [(64, 3), (128, 4)])
def test_medium_rows_and_checksum_vs_reference(golden, cuda, d, K):
    """G5: 2200-node Zipf graph; sampled rows + a checksum over every output bit."""
    from golden_inputs import exact_embedding
    from textgcn_amd.propagate import Propagator
    g = golden('g5_medium')
    gr = _graph(golden, 'g5_medium')
    e0 = np.concatenate([exact_embedding(gr.n_users, d, 21), exact_embedding(gr.n_items, d, 22)])
    prop = Propagator(gr, cuda, split_threshold=256)
    p = f'd{d}_'
    out = prop.forward(torch.from_numpy(e0).to(cuda), K, exact=True).cpu().numpy()
    assert np.array_equal(bits(out[g[p + 'rows']]), bits(g[p + 'repr_rows']))
    assert int(bits(out).astype(np.uint64).sum()) == int(g[p + 'repr_bits_sum'])
    # split mode: long rows (> 256 entries) are chained per chunk -> rounding-level differences only there
    assert prop.csr.n_chunks > 0
    out_s = prop.forward(torch.from_numpy(e0).to(cuda), K, exact=False).cpu().numpy()
    assert normwise(out_s[g[p + 'rows']], g[p + 'repr_rows']) <= 1e-6


def test_dropout_forward_vs_reference(golden, cuda, oracle):
    """G3: the training-mode forward on the reference's dropped matrix (captured mask)."""
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import Propagator
    g2, g3 = golden('g2_synth60'), golden('g3_dropout')
    gr = NormGraph.from_coo(g3['drop_idx'], g3['drop_val'], int(g2['n_users']), int(g2['n_items']))
    out = Propagator(gr, cuda).forward(torch.from_numpy(g2['a_layer0']).to(cuda), 3, exact=True).cpu().numpy()
    assert np.array_equal(bits(out), bits(np.concatenate([g3['users_emb'], g3['items_emb']])))


@pytest.mark.parametrize('d', [64, 128, 256, 32, 48, 100])
@pytest.mark.parametrize('variant', [0, 1])
def test_random_graph_bit_exact_vs_oracle(cuda, oracle, d, variant):
    """Seeded Zipf graph with empty rows, rows of every small length and a few long ones."""
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import DeviceCSR, spmm
    rng = np.random.default_rng(d * 7 + variant)
    gr = _random_graph(700, 300, 9000, seed=d)
    # add isolated nodes at the end of both blocks
    gr = NormGraph.from_pairs(*_pairs_of(gr), 705, 303)
    x = rng.standard_normal((gr.n, d)).astype(np.float32)
    idx, val = gr.to_coo()
    ref = oracle.spmm_coo(idx, val, x)
    csr = DeviceCSR(gr.rowptr, gr.colidx, gr.vals, gr.n, cuda)
    xd = torch.from_numpy(x).to(cuda)
    y = torch.full((gr.n, d), float('nan'), device=cuda)
    spmm(csr, xd, y=y, exact=True, variant=variant)
    assert np.array_equal(bits(y.cpu().numpy()), bits(ref))
    # fused epilogue: acc_out = (acc_in + y) / 5, with and without storing y
    acc_in = rng.standard_normal((gr.n, d)).astype(np.float32)
    want = ((acc_in + ref) / np.float32(5.0)).astype(np.float32)
    acc = torch.from_numpy(acc_in).to(cuda)
    out = torch.empty_like(acc)
    spmm(csr, xd, y=None, acc_in=acc, acc_out=out, acc_div=5.0, exact=True, variant=variant)
    assert np.array_equal(bits(out.cpu().numpy()), bits(want))
    spmm(csr, xd, y=y, acc_in=acc, acc_out=acc, acc_div=1.0, exact=True, variant=variant)   # in place, no division
    assert np.array_equal(bits(acc.cpu().numpy()), bits((acc_in + ref).astype(np.float32)))


def _pairs_of(gr):
    idx, _ = gr.to_coo()
    m = idx[0] < gr.n_users
    return idx[0][m], idx[1][m] - gr.n_users


@pytest.mark.parametrize('d', [64, 128, 256])
@pytest.mark.parametrize('threshold', [16, 100, 1024])
def test_split_rows_deterministic_and_close(cuda, oracle, d, threshold):
    """Long-row split: rows at or under the threshold stay bit-exact, split rows agree to rounding and are
    identical run to run (no atomics)."""
    from textgcn_amd.propagate import DeviceCSR, spmm
    gr = _random_graph(3000, 400, 40000, seed=3, zipf=1.1)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((gr.n, d)).astype(np.float32)
    idx, val = gr.to_coo()
    ref = oracle.spmm_coo(idx, val, x)
    csr = DeviceCSR(gr.rowptr, gr.colidx, gr.vals, gr.n, cuda, split_threshold=threshold)
    deg = gr.degrees()
    assert (deg > threshold).any() or threshold == 1024
    xd = torch.from_numpy(x).to(cuda)
    outs = []
    for variant in (1, 0, 1):
        y = torch.empty((gr.n, d), device=cuda)
        spmm(csr, xd, y=y, variant=variant)
        outs.append(y.cpu().numpy())
    short = deg <= threshold
    assert np.array_equal(bits(outs[0][short]), bits(ref[short]))
    assert normwise(outs[0], ref) <= 1e-6
    assert np.array_equal(bits(outs[0]), bits(outs[1])) and np.array_equal(bits(outs[0]), bits(outs[2]))


def test_full_size_c2_properties(cuda, oracle):
    """BASELINE config 2 (U=100k, I=50k, nnz=5M, d=64, K=3) at full size: checked through properties that
    do not need a full CPU run -- sampled rows against the oracle, linearity, and exact == split on short rows."""
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import Propagator
    n_u, n_i, nnz, d, K = synth.CONFIGS['c2']
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    gr = NormGraph.from_pairs(u, i, n_u, n_i)
    e0 = synth.embeddings(gr.n, d, seed=0)
    prop = Propagator(gr, cuda)
    e0d = e0.to(cuda)
    out_exact, layers = prop.forward(e0d, K, exact=True, keep_layers=True)
    out_split = prop.forward(e0d, K, exact=False)
    # (1) sampled rows of layer 1 against the oracle (CSR rows cut out on the host)
    rng = np.random.default_rng(0)
    rows = np.unique(np.concatenate([rng.integers(0, gr.n, 400), np.argsort(-gr.degrees())[:4]]))
    l1 = layers[1].cpu().numpy()
    x0 = e0.numpy()
    for r in rows:
        a, b = gr.rowptr[r], gr.rowptr[r + 1]
        sub = oracle.spmm_csr(np.array([0, b - a]), gr.colidx[a:b], gr.vals[a:b], x0)
        assert np.array_equal(bits(l1[r]), bits(sub[0])), r
    # (2) split vs exact: normwise tiny, identical on rows no layer of which saw a split row is too strong
    #     (split rows feed their neighbours) -> normwise bar only
    assert normwise(out_split.cpu().numpy(), out_exact.cpu().numpy()) <= 1e-5
    # (3) linearity: A(2x) == 2 A(x) exactly in fp32 (power-of-two scaling commutes with rounding)
    out2 = prop.forward(e0d * 2.0, K, exact=True)
    assert torch.equal(out2, out_exact * 2.0)
    # (4) mean of layers == explicit sequential sum / (K+1)
    s = layers[0].clone()
    for k in range(1, K + 1):
        s = s + layers[k]
    assert torch.equal(out_exact, s / float(K + 1))


def test_argument_validation(cuda):
    from textgcn_amd.propagate import DeviceCSR, spmm
    gr = _random_graph(50, 40, 300, seed=1)
    csr = DeviceCSR(gr.rowptr, gr.colidx, gr.vals, gr.n, cuda)
    x = torch.zeros((gr.n, 64), device=cuda)
    with pytest.raises(ValueError):
        spmm(csr, x[:, :32].contiguous()[:10], y=torch.zeros((gr.n, 32), device=cuda))
    with pytest.raises(ValueError):
        spmm(csr, x, y=x)
    with pytest.raises(TypeError):
        spmm(csr, x.double(), y=torch.zeros((gr.n, 64), device=cuda))
    with pytest.raises(RuntimeError):
        spmm(csr, x, y=None, acc_in=None, acc_out=None)   # nothing to compute -> C ABI error surfaces


def test_full_size_c3_properties(cuda, oracle):
    """BASELINE config 3 shape (U=180k, I=60k, nnz=1.6M, d=128, K=4): sampled rows of every layer against the
    oracle chain, exact == split on short rows of layer 1, linearity, layer mean identity."""
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import Propagator
    n_u, n_i, nnz, d, K = synth.CONFIGS['c3']
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    gr = NormGraph.from_pairs(u, i, n_u, n_i)
    e0 = synth.embeddings(gr.n, d, seed=0)
    prop = Propagator(gr, cuda)
    e0d = e0.to(cuda)
    out, layers = prop.forward(e0d, K, exact=True, keep_layers=True)
    rng = np.random.default_rng(3)
    rows = np.unique(np.concatenate([rng.integers(0, gr.n, 200), np.argsort(-gr.degrees())[:3]]))
    for k in range(1, K + 1):
        src = layers[k - 1].cpu().numpy()
        got = layers[k].cpu().numpy()
        for r in rows:
            a, b = gr.rowptr[r], gr.rowptr[r + 1]
            sub = oracle.spmm_csr(np.array([0, b - a]), gr.colidx[a:b], gr.vals[a:b], src)
            assert np.array_equal(bits(got[r]), bits(sub[0])), (k, r)
    s = layers[0].clone()
    for k in range(1, K + 1):
        s = s + layers[k]
    assert torch.equal(out, s / torch.tensor(float(K + 1), device=cuda))      # /5: a true division
    out_split, layers_split = prop.forward(e0d, K, exact=False, keep_layers=True)
    short = torch.from_numpy(gr.degrees() <= 1024).to(cuda)
    assert torch.equal(layers_split[1][short], layers[1][short])
    assert normwise(out_split.cpu().numpy(), out.cpu().numpy()) <= 1e-5
    assert torch.equal(prop.forward(e0d * 4.0, K, exact=True), out * 4.0)


def test_edge_cases_empty_and_isolated(cuda, oracle):
    """No interactions at all, isolated nodes, a single edge, n_layers = 0."""
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import Propagator
    e = np.zeros(0, dtype=np.int64)
    g0 = NormGraph.from_pairs(e, e, 7, 5)                      # 12 isolated nodes, nnz = 0
    x = torch.randn((12, 64), device=cuda)
    out = Propagator(g0, cuda).forward(x, 3)
    assert torch.equal(out, x / 4.0)                            # every layer is zero: mean = E0 / (K+1)
    g1 = NormGraph.from_pairs(np.array([2]), np.array([3]), 7, 5)
    out1 = Propagator(g1, cuda).forward(x, 2, exact=True).cpu().numpy()
    idx, val = g1.to_coo()
    ref, _ = oracle.propagate(idx, val, x.cpu().numpy(), 2)
    assert np.array_equal(bits(out1), bits(ref))
    assert torch.equal(Propagator(g1, cuda).forward(x, 0), x)   # K = 0: representation is E0


@pytest.mark.parametrize('d', [64, 128, 256])
@pytest.mark.parametrize('blocks', [(0, 8), (8, 16), (24, 0)])
@pytest.mark.parametrize('tile,min_len', [(256, 0), (64, 32)])
def test_segmented_xcd_affine_kernel(cuda, oracle, d, blocks, tile, min_len):
    """tgcn_spmm_segmented_f32: rows cut at column-block boundaries, piece sums added in column order.  Direct rows
    are bit-exact, segmented rows agree with the one-chain result to rounding, runs are identical (no atomics), and the
    fused epilogue (acc_in/acc_out/acc_div, Y optional) matches the oracle's layer sum."""
    from textgcn_amd.propagate import Propagator, spmm
    gr = _random_graph(1500, 700, 30000, seed=5, zipf=1.0)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((gr.n, d)).astype(np.float32)
    e0 = rng.standard_normal((gr.n, d)).astype(np.float32)
    idx, val = gr.to_coo()
    ref = oracle.spmm_coo(idx, val, x)
    prop = Propagator(gr, cuda)
    prop.csr.configure_segments(list(blocks), tile_entries=tile, min_row_len=min_len)    # specs: (user rows, item rows)
    xd, e0d = torch.from_numpy(x).to(cuda), torch.from_numpy(e0).to(cuda)
    outs = []
    for unroll in (0, 8, 0):
        y = torch.full((gr.n, d), float('nan'), device=cuda)
        spmm(prop.csr, xd, y=y, segmented=True, unroll=unroll)
        outs.append(y.cpu().numpy())
    assert normwise(outs[0], ref) <= 5e-6    # rounding only (the path's tolerance is 1e-4)
    assert np.array_equal(bits(outs[0]), bits(outs[1])) and np.array_equal(bits(outs[0]), bits(outs[2]))
    direct = np.zeros(gr.n, dtype=bool)
    if blocks[0] == 0:
        direct[:gr.n_users] = True
    if blocks[1] == 0:
        direct[gr.n_users:] = True
    direct |= gr.degrees() < max(min_len, 1)
    assert not direct.all()
    assert np.array_equal(bits(outs[0][direct]), bits(ref[direct]))
    # fused epilogue on top of the segmented sums
    acc = torch.full((gr.n, d), float('nan'), device=cuda)
    spmm(prop.csr, xd, y=None, acc_in=e0d, acc_out=acc, acc_div=4.0, segmented=True)
    want = (e0 + outs[0]) / np.float32(4.0)
    assert np.array_equal(bits(acc.cpu().numpy()), bits(want.astype(np.float32)))
    # per-call values (edge dropout, base_model.py:77-86): gathered into the plan's streams
    torch.manual_seed(0)
    v2 = (prop.csr.vals * (torch.rand_like(prop.csr.vals) > 0.4)).contiguous()
    ya, yb = torch.empty((gr.n, d), device=cuda), torch.empty((gr.n, d), device=cuda)
    spmm(prop.csr, xd, y=ya, vals=v2, segmented=True)
    spmm(prop.csr, xd, y=yb, vals=v2, segmented=False, exact=True)
    assert normwise(ya.cpu().numpy(), yb.cpu().numpy()) <= 5e-6
    assert np.array_equal(bits(ya.cpu().numpy()[direct]), bits(yb.cpu().numpy()[direct]))
    assert not torch.equal(ya, torch.from_numpy(outs[0]).to(cuda))
    # whole forward through the segmented path
    full = prop.forward(e0d, 3, segmented=True).cpu().numpy()
    exact = prop.forward(e0d, 3, exact=True).cpu().numpy()
    assert normwise(full, exact) <= 5e-6


@pytest.mark.parametrize('d', [64, 128, 256])
@pytest.mark.parametrize('unroll', [0, 8, 16, 32])
def test_row_groups_bit_identical_to_one_wave_per_row(cuda, oracle, d, unroll):
    """tgcn_spmm_groups_f32 (round 4): consecutive rows share a wave.  Rows without entries at the start, middle and end of a
    group, runs of one-entry rows (eight rows per group), rows of exactly 63 / 64 / 65 / 128 entries (slab boundaries), a row
    that ends on a slab's last entry, long rows beside short ones, with and without the split plan: the same bits as the
    one-wave-per-row kernel and the oracle, for Y, the fused epilogue and the in-place form."""
    from textgcn_amd import _capi
    from textgcn_amd.propagate import DeviceCSR, spmm
    rng = np.random.default_rng(1000 + d + unroll)
    n_src = 900
    lens = np.concatenate([
        [0, 0, 0, 5, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 63, 1, 64, 65, 0, 128, 3, 61, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 7],
        rng.integers(0, 4, 300), rng.integers(0, 40, 200), [700, 2, 0, 1500, 1, 1], rng.integers(0, 130, 150), [0, 0, 0]])
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cols = np.concatenate([np.sort(rng.choice(n_src, size=min(l, n_src), replace=False)) if l <= n_src else
                           np.sort(rng.integers(0, n_src, l)) for l in lens]).astype(np.int32)
    vals = rng.standard_normal(len(cols)).astype(np.float32)
    n = len(lens)
    x = rng.standard_normal((n_src, d)).astype(np.float32)
    idx = np.stack([np.repeat(np.arange(n), lens), cols.astype(np.int64)])
    # (the oracle's COO product wants a square shape: pad)
    size = max(n, n_src)
    xp = np.zeros((size, d), dtype=np.float32)
    xp[:n_src] = x
    ref = oracle.spmm_coo(idx, vals, xp)[:n]
    xd = torch.from_numpy(x).to(cuda)
    acc_in = rng.standard_normal((n, d)).astype(np.float32)
    for thr in (None, 256):
        csr = DeviceCSR(rowptr, cols, vals, n_src, cuda, split_threshold=thr)
        grp = csr.groups(d, exact=thr is None)
        assert grp is not None and grp[1] > 0
        g = grp[0].cpu().numpy()[:grp[1]]
        assert g[:, 1].max() == (4 if d == 256 else 8)
        exact = thr is None
        y = torch.full((n, d), float('nan'), device=cuda)
        spmm(csr, xd, y=y, exact=exact, unroll=unroll)
        y1 = torch.full((n, d), float('nan'), device=cuda)
        spmm(csr, xd, y=y1, exact=exact, variant=_capi.SPMM_WAVE_PER_ROW)
        assert torch.equal(y.view(torch.int32), y1.view(torch.int32)), thr
        short = lens <= (thr or (1 << 30))
        assert np.array_equal(bits(y.cpu().numpy()[short]), bits(ref[short])), thr
        if thr is not None:
            assert normwise(y.cpu().numpy(), ref) <= 1e-5
        acc = torch.from_numpy(acc_in).to(cuda)
        out = torch.empty_like(acc)
        spmm(csr, xd, y=None, acc_in=acc, acc_out=out, acc_div=4.0, exact=exact, unroll=unroll)
        out1 = torch.empty_like(acc)
        spmm(csr, xd, y=None, acc_in=acc, acc_out=out1, acc_div=4.0, exact=exact, variant=_capi.SPMM_WAVE_PER_ROW)
        assert torch.equal(out.view(torch.int32), out1.view(torch.int32)), thr
        want = ((acc_in + ref) / np.float32(4.0)).astype(np.float32)
        assert np.array_equal(bits(out.cpu().numpy()[short]), bits(want[short])), thr
        y2 = torch.empty_like(y)
        spmm(csr, xd, y=y2, acc_in=acc, acc_out=acc, acc_div=1.0, exact=exact, unroll=unroll)     # in place + Y
        assert torch.equal(y2.view(torch.int32), y.view(torch.int32))
        assert np.array_equal(bits(acc.cpu().numpy()[short]), bits((acc_in + ref).astype(np.float32)[short])), thr

