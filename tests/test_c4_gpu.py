"""BASELINE config 4 (U=5M, I=2M, nnz=100M, d=64, K=3) at FULL size on one GPU -- the size the north-star target is stated
on.  A CPU run of the whole forward would take minutes, so the checks are the size-independent ones: sampled rows (the
1 M-entry hottest row included) against the oracle's fmaf chain, the default (split) path against the exact one, linearity
under power-of-two scaling, the layer-mean identity, and the row-sharded layout arithmetic at this size.  The graph is
built once per session (~35 s on the host)."""
import numpy as np
import pytest
import torch

from conftest import bits, normwise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def c4():
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    n_u, n_i, nnz, d, K = synth.CONFIGS['c4']
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    gr = NormGraph.from_pairs(u, i, n_u, n_i)
    del u, i
    e0 = synth.embeddings(gr.n, d, seed=0)
    return gr, e0, d, K


def test_full_size_c4_properties(cuda, oracle, c4):
    from textgcn_amd.propagate import Propagator
    gr, e0, d, K = c4
    assert gr.nnz >= 200_000_000 and gr.n == 7_000_000
    prop = Propagator(gr, cuda)
    assert prop.csr.row_order is None            # the branch config 4 takes: rows in natural order (propagate.py)
    assert prop.csr.n_chunks > 0                  # rows beyond the split threshold exist (hottest row: ~1 M entries)
    e0d = e0.to(cuda)
    out_exact, layers = prop.forward(e0d, K, exact=True, keep_layers=True)
    # (1) sampled rows of every layer against the oracle chain; the hottest rows included
    rng = np.random.default_rng(0)
    deg = gr.degrees()
    hot = np.argsort(-deg)[:3]
    assert deg[hot[0]] > 500_000
    rows = np.unique(np.concatenate([rng.integers(0, gr.n, 300), hot, [0, gr.n_users - 1, gr.n_users, gr.n - 1]]))
    rows_d = torch.from_numpy(rows).to(cuda)
    for k in range(1, K + 1):
        src = layers[k - 1].cpu().numpy()
        got = layers[k][rows_d].cpu().numpy()
        for j, r in enumerate(rows):
            a, b = gr.rowptr[r], gr.rowptr[r + 1]
            sub = oracle.spmm_csr(np.array([0, b - a]), gr.colidx[a:b], gr.vals[a:b], src)
            assert np.array_equal(bits(got[j]), bits(sub[0])), (k, r)
        del src
    # (2) layer mean == explicit sequential sum / (K+1), bit for bit
    s = layers[0].clone()
    for k in range(1, K + 1):
        s += layers[k]
    assert torch.equal(out_exact, s / float(K + 1))
    del s, layers
    # (3) the default path (long rows split in 1024-entry chunks) vs the exact chains: rounding only
    out_split = prop.forward(e0d, K, exact=False)
    ref = out_exact.cpu().numpy()
    assert normwise(out_split.cpu().numpy(), ref) <= 1e-5
    # the default path also cuts the HOT item rows (>= 640 entries here) at 8 MB windows of the user table (propagate.segment_blocks_auto)
    assert prop.csr.segment_blocks and prop.csr.segment_blocks[0] == 0 and prop.csr.segment_blocks[1][0] >= 104
    cut_from = min(1024, prop.csr.segment_blocks[1][2])
    short = torch.from_numpy(np.nonzero(deg < cut_from)[0][:100000]).to(cuda)
    # rows that were never cut differ only through their (cut) neighbours: still within the bar, and layer 1 of them is exact
    y1a, y1b = torch.empty_like(e0d), torch.empty_like(e0d)
    from textgcn_amd.propagate import spmm
    spmm(prop.csr, e0d, y=y1a, exact=True)
    spmm(prop.csr, e0d, y=y1b, exact=False)
    assert torch.equal(y1a[short], y1b[short])
    del y1a, y1b, out_split
    # (4) linearity: A(2x) == 2 A(x) exactly in fp32 (power-of-two scaling commutes with rounding)
    out2 = prop.forward(e0d * 2.0, K, exact=True)
    assert torch.equal(out2, out_exact * 2.0)


def test_c4_row_partition_arithmetic(c4):
    """The sharded layout at config-4 size, P = 8 (host arithmetic only -- the 8-GPU run is the driver's): blocks are
    nnz-balanced, every global id gets a distinct table row, padding stays small, int32 offsets hold."""
    from textgcn_amd.dist import BlockLayout
    gr, _, d, _ = c4
    ub, ib = gr.partition(8)
    for b, lo, hi in ((ub, 0, gr.n_users), (ib, gr.n_users, gr.n)):
        assert b[0] == lo and b[-1] == hi and np.all(np.diff(b) > 0)
        per = np.diff(gr.rowptr[b]).astype(np.float64)
        # cuts fall between rows (the weight is entries + 1 per row, so that empty rows spread too): a rank's share is off by
        # at most one row's entries plus its row-count difference; the hottest item row alone is 8 % of a share
        assert per.max() / per.mean() < 1.0 + max(1e-3, 1.1 * gr.degrees()[lo:hi].max() / per.mean())
        assert per.max() < 2 ** 31
    lay = BlockLayout(ib - gr.n_users, chunks=4)
    rows = lay.table_rows(np.arange(gr.n_items))
    assert rows.max() < lay.n_pad and len(np.unique(rows)) == gr.n_items
    assert lay.n_pad <= 1.1 * gr.n_items                          # padding to the largest block costs < 10 %
    assert (lay.n_pad + BlockLayout(ub, chunks=4).n_pad) * d * 4 < 2 ** 32   # 32-bit byte offsets of the segmented kernel not needed, but fit


def test_c4_two_rank_sharded_forward_equals_single_gpu(cuda, c4, tmp_path):
    """Config 4 itself through the sharded path: two ranks on the one GPU (gloo staged through the host -- the rehearsal
    form; the driver's 8-GPU run uses RCCL), nnz-balanced blocks, 4 row chunks per block as bench.py uses.  Sampled rows of
    rank 0's users and of the gathered item table must carry the single-GPU bits."""
    from test_dist_cpu import run_ranks
    from textgcn_amd.propagate import Propagator
    gr, e0, d, K = c4
    out = str(tmp_path / 'r0.npz')
    # the worker draws E0 with seed 2; the graph is synth seed 0 as in the fixture
    run_ranks(2, 'gpu', out, extra=('--n-users', str(gr.n_users), '--n-items', str(gr.n_items), '--nnz', '100000000', '--graph-seed', '0',
                                    '--balance', 'nnz', '--chunks', '4', '--sample', '2000', '--no-segment'))
    got = np.load(out)
    from textgcn_amd import synth
    e0w = synth.embeddings(gr.n, d, seed=2).to(cuda)
    ref = Propagator(gr, cuda, split_threshold=64, segment=None).forward(e0w, K)
    ur = torch.from_numpy(got['user_rows']).to(cuda)
    ir = torch.from_numpy(got['item_rows'] + gr.n_users).to(cuda)
    assert np.array_equal(bits(got['users']), bits(ref[ur].cpu().numpy()))
    assert np.array_equal(bits(got['items']), bits(ref[ir].cpu().numpy()))
    # and with the hot-row windows on (what bench.py runs): the same rows to rounding
    run_ranks(2, 'gpu', out, extra=('--n-users', str(gr.n_users), '--n-items', str(gr.n_items), '--nnz', '100000000', '--graph-seed', '0',
                                    '--balance', 'nnz', '--chunks', '4', '--sample', '2000', '--split-threshold', '1024'))
    got = np.load(out)
    assert normwise(got['users'], ref[ur].cpu().numpy()) <= 1e-5 and normwise(got['items'], ref[ir].cpu().numpy()) <= 1e-5
