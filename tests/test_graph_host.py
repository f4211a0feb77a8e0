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


def test_segment_plan_streams_reproduce_the_product():
    """XCD-affine segment plan (graph.segment_plan_arrays): walking the tiles as the kernel does and adding a row's
    pieces in row_slots order reproduces A.x; every tile stays in one column-block class and the 4 tiles of a
    workgroup share it; a row's pieces are in ascending column order."""
    from textgcn_amd.graph import NormGraph, segment_plan_arrays
    from textgcn_amd import synth
    u, i = synth.interactions(300, 170, 6000, seed=3)
    g = NormGraph.from_pairs(u, i, 300, 170)
    U, N = g.n_users, g.n
    x = np.random.default_rng(0).standard_normal((N, 8))
    ref = np.zeros((N, 8))
    np.add.at(ref, np.repeat(np.arange(N), np.diff(g.rowptr)), g.vals[:, None].astype(np.float64) * x[g.colidx])
    for nb_items, nb_users, T in ((8, 0, 64), (16, 8, 128), (8, 24, 64)):
        phases = [(U, N, 0, U, nb_items)] + ([(0, U, U, N, nb_users)] if nb_users else [])
        p = segment_plan_arrays(g.rowptr, g.colidx, g.vals, phases, tile_entries=T)
        n_tiles = len(p['tile_meta'])
        assert n_tiles % 4 == 0 and len(p['ent_col']) == n_tiles * T == len(p['ent_val']) == 64 * len(p['ent_flags'])
        ws = np.zeros((p['n_slots'], 8))
        written = np.zeros(p['n_slots'], dtype=np.int32)
        tile_class = np.full(n_tiles, -1)
        slot_first_col = np.zeros(p['n_slots'], dtype=np.int64)
        for t, (slot, n) in enumerate(p['tile_meta']):
            acc, first_col = np.zeros(8), None
            for e in range(t * T, t * T + n):
                c = int(p['ent_col'][e])
                flag = (int(p['ent_flags'][e // 64]) >> (e % 64)) & 1
                first_col = c if first_col is None else first_col
                acc = acc + np.float64(p['ent_val'][e]) * x[c]
                for (r0, r1, c0, c1, nb) in phases:
                    if c0 <= c < c1:
                        cls = ((c - c0) // -(-(c1 - c0) // nb)) % 8
                        assert tile_class[t] in (-1, cls)
                        tile_class[t] = cls
                if flag:
                    ws[slot], written[slot], slot_first_col[slot] = acc, written[slot] + 1, first_col
                    slot, acc, first_col = slot + 1, np.zeros(8), None
            assert n == 0 or first_col is None     # a tile's last entry closes its piece
        assert np.all(written == 1)
        live = tile_class.reshape(-1, 4)
        for wg in live:
            assert len(set(wg[wg >= 0].tolist())) <= 1
        y = np.zeros((N, 8))
        for k, r in enumerate(p['seg_rows']):
            sl = p['row_slots'][p['row_slot_ptr'][k]:p['row_slot_ptr'][k + 1]]
            assert len(sl) and np.all(np.diff(slot_first_col[sl]) > 0)
            y[r] = ws[sl].sum(axis=0)
        seg = np.zeros(N, dtype=bool)
        seg[p['seg_rows']] = True
        assert sorted(np.concatenate([p['seg_rows'], p['direct_rows']]).tolist()) == list(range(N))
        assert np.allclose(y[seg], ref[seg], rtol=1e-12, atol=1e-12)
        if not nb_users:
            assert np.all(p['direct_rows'] < U) and len(p['direct_rows']) == U


def test_segment_heuristic_picks_item_rows_of_config_2_only():
    """propagate.segment_blocks_auto: on BASELINE config 2 the item rows (uniformly used 25.6 MB user table, 100
    entries per row) are segmented, the user rows (Zipf-popular 12.8 MB item table, 50 entries per row) are not; a
    small graph and a width-128 table beyond 8 L2 shares are left alone."""
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import segment_blocks_auto
    n_u, n_i, nnz, d, _ = synth.CONFIGS['c2']
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    specs = [(0, n_u, n_u, g.n), (n_u, g.n, 0, n_u)]
    assert [segment_blocks_auto(g.rowptr, g.colidx, sp, 64) for sp in specs] == [0, 8]
    assert [segment_blocks_auto(g.rowptr, g.colidx, sp, 128) for sp in specs] == [0, 0]     # 51 MB user table at d = 128
    assert [segment_blocks_auto(g.rowptr, g.colidx, sp, 48) for sp in specs] == [0, 0]      # no kernel for this width
    u, i = synth.interactions(600, 300, 9000, seed=1)
    gs = NormGraph.from_pairs(u, i, 600, 300)
    assert [segment_blocks_auto(gs.rowptr, gs.colidx, sp, 64) for sp in [(0, 600, 600, 900), (600, 900, 0, 600)]] == [0, 0]


def test_hot_row_rule_for_tables_beyond_the_infinity_cache():
    """propagate.segment_blocks_auto, round 4: over a gather table larger than the Infinity Cache (config 4's 1.28 GB user table) only
    the LONG rows are cut, at 8 MB windows -- (blocks, classes, min_row_len) -- and only where such rows hold enough entries; the
    rule needs row lengths only (no column scan)."""
    from textgcn_amd.propagate import segment_blocks_auto
    n_cols, d = 5_000_000, 64
    lens = np.concatenate([np.full(100, 50_000), np.full(5000, 20)])
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    got = segment_blocks_auto(rowptr, None, (0, len(lens), 0, n_cols), d)
    assert got == (160, 8, 640)
    assert segment_blocks_auto(rowptr, None, (0, len(lens), 0, n_cols), 128) == (312, 8, 1248)
    short = np.concatenate([[0], np.cumsum(np.full(200_000, 30))]).astype(np.int64)
    assert segment_blocks_auto(short, None, (0, 200_000, 0, n_cols), d) == 0            # nothing long enough to cut
    few = np.concatenate([[0], np.cumsum(np.concatenate([np.full(3, 1000), np.full(400_000, 30)]))]).astype(np.int64)
    assert segment_blocks_auto(few, None, (0, 400_003, 0, n_cols), d) == 0              # long rows exist but hold < 10 % of the entries
    assert segment_blocks_auto(rowptr, None, (0, len(lens), 0, 600_000), d) == 0        # 154 MB table: inside the Infinity Cache
