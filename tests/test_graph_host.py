"""Host logic of the product (no GPU): graph builder bit-exact vs the reference's norm_matrix, CSR views,
partition, split plans, mask CSR."""
import numpy as np
import pytest

from conftest import bits
from textgcn_amd.graph import NormGraph, split_plan_arrays, train_mask_csr


@pytest.mark.parametrize('name,prefix', [('g1_dummy', ''), ('g2_synth60', ''), ('g6_builder', 'dup_'),
                                         ('g6_builder', 'star_'), ('g6_builder', 'rand_')])
def test_from_pairs_matches_reference(golden, name, prefix):
    g = golden(name)
    gr = NormGraph.from_pairs(g[prefix + 'train_u'], g[prefix + 'train_i'], int(g[prefix + 'n_users']), int(g[prefix + 'n_items']))
    idx, val = gr.to_coo()
    assert np.array_equal(idx, g[prefix + 'norm_idx'])
    assert np.array_equal(bits(val), bits(g[prefix + 'norm_val']))
    assert gr.nnz == len(val) and gr.n == int(g[prefix + 'n_users']) + int(g[prefix + 'n_items'])


def test_from_pairs_medium_checksums(golden):
    g = golden('g5_medium')
    gr = NormGraph.from_pairs(g['train_u'], g['train_i'], int(g['n_users']), int(g['n_items']))
    idx, val = gr.to_coo()
    assert int(idx[0].sum()) == int(g['norm_row_sum'])
    assert int((idx[1] * (np.arange(idx.shape[1]) % 1009)).sum()) == int(g['norm_col_weighted'])
    assert np.array_equal(bits(val), bits(g['norm_val']))


def test_from_pairs_order_independent_and_isolated_nodes():
    rng = np.random.default_rng(0)
    u = rng.integers(0, 30, 200)
    i = rng.integers(0, 20, 200)
    a = NormGraph.from_pairs(u, i, 33, 25)          # users 30..32 and items 20..24 isolated
    p = rng.permutation(200)
    b = NormGraph.from_pairs(u[p], i[p], 33, 25)
    assert np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.colidx, b.colidx)
    assert np.array_equal(bits(a.vals), bits(b.vals))
    deg = a.degrees()
    assert np.all(deg[30:33] == 0) and np.all(deg[33 + 20:] == 0)
    assert np.all(np.isfinite(a.vals))


def test_from_pairs_rejects_bad_input():
    with pytest.raises(ValueError):
        NormGraph.from_pairs([0, 5], [0, 1], 3, 3)
    with pytest.raises(ValueError):
        NormGraph.from_pairs([0, 1], [0], 3, 3)
    with pytest.raises(TypeError):
        NormGraph.from_pairs(np.array([0.5]), np.array([1]), 3, 3)


def test_empty_graph():
    g = NormGraph.from_pairs(np.zeros(0, np.int64), np.zeros(0, np.int64), 4, 3)
    assert g.nnz == 0 and len(g.rowptr) == 8


def test_from_coo_roundtrip_and_validation(golden):
    g = golden('g2_synth60')
    gr = NormGraph.from_coo(g['norm_idx'], g['norm_val'], int(g['n_users']), int(g['n_items']))
    ref = NormGraph.from_pairs(g['train_u'], g['train_i'], int(g['n_users']), int(g['n_items']))
    assert np.array_equal(gr.rowptr, ref.rowptr) and np.array_equal(gr.colidx, ref.colidx)
    bad = g['norm_idx'][:, ::-1]
    with pytest.raises(ValueError):
        NormGraph.from_coo(bad, g['norm_val'], int(g['n_users']), int(g['n_items']))


def test_from_coo_accepts_torch_sparse(golden):
    import torch
    g = golden('g1_dummy')
    t = torch.sparse_coo_tensor(torch.from_numpy(g['norm_idx']), torch.from_numpy(g['norm_val']), (9, 9)).coalesce()
    gr = NormGraph.from_coo(t, None, 5, 4)
    assert np.array_equal(gr.to_coo()[0], g['norm_idx'])


def test_transpose_perm_symmetric(golden):
    g = golden('g2_synth60')
    gr = NormGraph.from_pairs(g['train_u'], g['train_i'], int(g['n_users']), int(g['n_items']))
    perm = gr.transpose_perm()
    assert np.array_equal(bits(gr.vals[perm]), bits(gr.vals))   # A is symmetric
    v = np.arange(gr.nnz, dtype=np.float32)                     # a non-symmetric valuation on the structure
    idx, _ = gr.to_coo()
    import scipy.sparse as sp
    a = sp.coo_matrix((v, (idx[0], idx[1])), shape=(gr.n, gr.n)).tocsr()
    at = a.T.tocsr()
    at.sort_indices()
    assert np.array_equal(at.data, v[perm])


@pytest.mark.parametrize('world', [1, 2, 3, 8])
def test_partition_covers_and_balances(golden, world):
    g = golden('g5_medium')
    gr = NormGraph.from_pairs(g['train_u'], g['train_i'], int(g['n_users']), int(g['n_items']))
    ub, ib = gr.partition(world)
    assert ub[0] == 0 and ub[-1] == gr.n_users and ib[0] == gr.n_users and ib[-1] == gr.n
    assert np.all(np.diff(ub) >= 0) and np.all(np.diff(ib) >= 0)
    nnz_u = [gr.rowptr[ub[p + 1]] - gr.rowptr[ub[p]] for p in range(world)]
    nnz_i = [gr.rowptr[ib[p + 1]] - gr.rowptr[ib[p]] for p in range(world)]
    assert sum(nnz_u) + sum(nnz_i) == gr.nnz
    if world > 1:
        assert max(nnz_u) <= 1.5 * (sum(nnz_u) / world) + gr.degrees()[:gr.n_users].max()
        assert max(nnz_i) <= 1.5 * (sum(nnz_i) / world) + gr.degrees()[gr.n_users:].max()


def test_row_block_views(golden):
    g = golden('g2_synth60')
    gr = NormGraph.from_pairs(g['train_u'], g['train_i'], int(g['n_users']), int(g['n_items']))
    rp, ci, va = gr.row_block(10, 37)
    assert rp[0] == 0 and rp[-1] == len(ci) == len(va)
    assert np.array_equal(np.diff(rp), gr.degrees()[10:37])


def test_split_plan_arrays():
    rowptr = np.array([0, 3, 3, 13, 14, 40])
    assert split_plan_arrays(rowptr, 100) is None
    p = split_plan_arrays(rowptr, 8)
    assert p['long_rows'].tolist() == [2, 4]
    assert p['long_chunk_ptr'].tolist() == [0, 2, 6]
    assert p['chunk_beg'].tolist() == [3, 11, 14, 22, 30, 38]
    assert p['chunk_end'].tolist() == [11, 13, 22, 30, 38, 40]
    # exactly at the threshold: not long
    assert split_plan_arrays(np.array([0, 8]), 8) is None


def test_train_mask_csr_matches_oracle(golden, oracle):
    g = golden('g2_synth60')
    rp, items = train_mask_csr(g['train_u'], g['train_i'], int(g['n_users']))
    orp, oitems = oracle.train_mask_csr(g['train_u'], g['train_i'], np.arange(int(g['n_users'])))
    assert np.array_equal(rp, orp) and np.array_equal(items, oitems)


def test_block_plan_arrays_segments_partition_rows(golden):
    from textgcn_amd.graph import block_plan_arrays
    g = golden('g5_medium')
    gr = NormGraph.from_pairs(g['train_u'], g['train_i'], int(g['n_users']), int(g['n_items']))
    u, n = gr.n_users, gr.n
    for (r0, r1, c0, c1) in ((0, u, u, n), (u, n, 0, u)):
        bp, nb = block_plan_arrays(gr.rowptr, gr.colidx, r0, r1, c0, c1, 100, long_threshold=64)
        assert bp.shape == (nb + 1, r1 - r0) and nb >= 2
        lens = np.diff(gr.rowptr[r0:r1 + 1])
        assert np.all(np.diff(bp.astype(np.int64), axis=0) >= 0)
        short = lens <= 64
        assert np.array_equal(bp[0][short], gr.rowptr[r0:r1][short]) and np.array_equal(bp[nb][short], gr.rowptr[r0 + 1:r1 + 1][short])
        assert np.all(bp[:, ~short] == gr.rowptr[r0:r1][~short])            # long rows: empty segments
        width = -(-(c1 - c0) // nb)
        for r in np.nonzero(short)[0][::37]:
            for b in range(nb):
                seg = gr.colidx[bp[b, r]:bp[b + 1, r]]
                assert np.all((seg >= c0 + b * width) & (seg < c0 + (b + 1) * width))
    with pytest.raises(ValueError):
        block_plan_arrays(gr.rowptr, gr.colidx, 0, u, 0, u, 100)                # wrong column range


def test_segment_plan_covers_every_entry_once_in_order():
    """XCD-affine segment plan (graph.segment_plan_arrays): every stored entry of a phase row lies in exactly one
    segment, a row's slots follow its entry order, positions of one workgroup share a column-block class."""
    from textgcn_amd.graph import NormGraph, segment_plan_arrays
    from textgcn_amd import synth
    u, i = synth.interactions(300, 170, 6000, seed=3)
    g = NormGraph.from_pairs(u, i, 300, 170)
    U, N = g.n_users, g.n
    for nb_items, nb_users, S, max_len in ((8, 4, 16, 128), (16, 0, 8, 5), (2, 8, 4, 7)):
        phases = [(U, N, 0, U, nb_items)] + ([(0, U, U, N, nb_users)] if nb_users else [])
        p = segment_plan_arrays(g.rowptr, g.colidx, phases, S, max_len)
        assert len(p['seg_beg']) % S == 0
        live = p['seg_end'] > p['seg_beg']
        assert (p['seg_end'] - p['seg_beg'])[live].max() <= max_len
        # coverage: entries of segment rows exactly once
        mark = np.zeros(g.nnz, dtype=np.int32)
        for b, e in zip(p['seg_beg'][live], p['seg_end'][live]):
            mark[b:e] += 1
        seg_row_mask = np.zeros(N, dtype=bool)
        seg_row_mask[p['seg_rows']] = True
        ent_row = np.repeat(np.arange(N), np.diff(g.rowptr))
        assert np.array_equal(mark == 1, seg_row_mask[ent_row]) and mark.max() <= 1
        assert sorted(np.concatenate([p['seg_rows'], p['direct_rows']]).tolist()) == list(range(N))
        # slots: unique, a row's slots ascending with the entry offset
        slots = p['seg_slot'][live]
        assert len(np.unique(slots)) == len(slots) == p['n_slots']
        beg_of_slot = np.empty(p['n_slots'], dtype=np.int64)
        beg_of_slot[slots] = p['seg_beg'][live]
        for k, r in enumerate(p['seg_rows']):
            s0, s1 = p['seg_row_ptr'][k], p['seg_row_ptr'][k + 1]
            b = beg_of_slot[s0:s1]
            assert s1 > s0 and np.all(np.diff(b) > 0) and g.rowptr[r] <= b[0] and b[-1] < g.rowptr[r + 1]
        # class affinity: all live segments of a workgroup read columns of one block class
        first_col = g.colidx[np.minimum(p['seg_beg'], g.nnz - 1)]
        for (r0, r1, c0, c1, nb) in phases:
            width = -(-(c1 - c0) // nb)
            in_phase = live & (first_col >= c0) & (first_col < c1)
            blk = (first_col - c0) // width
            wg = np.arange(len(live)) // S
            for w in np.unique(wg[in_phase]):
                cls = np.unique(blk[in_phase & (wg == w)] % 8) if nb >= 8 else np.unique(blk[in_phase & (wg == w)])
                assert len(cls) == 1
