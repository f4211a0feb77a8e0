"""N > 1 path on CPU: world_size 2 and 3 over gloo.  The exchange logic (equal padded blocks, in-place
all-gather per half-step, fused layer sum on local rows, item block of the mean gathered last) must reproduce
the single-process oracle bit for bit -- row ownership never changes a row's summation order."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, bits


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(world, mode, out, extra=()):
    port = free_port()
    env = dict(os.environ, OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), '--rank', str(r), '--world',
                               str(world), '--port', str(port), '--mode', mode, '--out', out, *extra], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]


def reference(oracle, n_users=203, n_items=97, nnz=2500, d=64, layers=3, single=False):
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    u, i = synth.interactions(n_users, n_items, nnz, seed=1)
    g = NormGraph.from_pairs(u, i, n_users, n_items)
    e0 = synth.embeddings(g.n, d, seed=2).numpy()
    idx, val = g.to_coo()
    out, _ = oracle.propagate(idx, val, e0, layers, single=single)
    return out[:n_users], out[n_users:]


@pytest.mark.parametrize('world,balance,chunks', [(2, 'nnz', 1), (3, 'nnz', 3), (2, 'rows', 2), (3, 'rows', 1)])
@pytest.mark.parametrize('single', [False, True])
def test_sharded_forward_matches_oracle_gloo(oracle, tmp_path, world, balance, chunks, single):
    """gloo gathers the CPU tensors asynchronously (async_op=True + work.wait(), in place in the layer table), so the
    wait placement of the overlap schedule is exercised, not only the partition arithmetic."""
    out = str(tmp_path / 'r0.npz')
    run_ranks(world, 'cpu', out, extra=('--balance', balance, '--chunks', str(chunks)) + (('--single',) if single else ()))
    got = np.load(out)
    ru, ri = reference(oracle, single=single)
    assert np.array_equal(bits(got['users']), bits(ru))
    assert np.array_equal(bits(got['items']), bits(ri))
    if balance == 'nnz':   # blocks differ in rows, not (much) in entries
        assert len(set(np.diff(got['item_bounds']))) > 1


@pytest.mark.parametrize('world,n_users,n_items,nnz,d,chunks,balance', [
    (2, 3, 2, 4, 8, 2, 'nnz'),          # fewer rows than ranks x chunks: empty chunks, a rank without item rows
    (3, 5, 97, 300, 16, 4, 'rows'),     # user blocks of one or two rows, four chunks each
    (3, 203, 2, 300, 64, 1, 'nnz'),     # two items for three ranks: one rank owns no item row at all
    (4, 61, 7, 200, 32, 3, 'nnz'),      # world 4
    (2, 1, 1, 1, 100, 1, 'rows'),       # one edge, a width that is no power of two
])
def test_sharded_forward_corner_shapes_gloo(oracle, tmp_path, world, n_users, n_items, nnz, d, chunks, balance):
    """Blocks, padding and chunking at shapes where some rank's block (or chunk) is empty: still the oracle's bits."""
    out = str(tmp_path / 'r0.npz')
    run_ranks(world, 'cpu', out, extra=('--balance', balance, '--chunks', str(chunks), '--n-users', str(n_users), '--n-items', str(n_items),
                                        '--nnz', str(nnz), '--d', str(d), '--layers', '2'))
    got = np.load(out)
    ru, ri = reference(oracle, n_users=n_users, n_items=n_items, nnz=nnz, d=d, layers=2)
    assert np.array_equal(bits(got['users']), bits(ru))
    assert np.array_equal(bits(got['items']), bits(ri))


def test_padded_layout_and_local_blocks():
    from textgcn_amd.dist import BlockLayout, equal_row_bounds, padded_layout
    assert padded_layout(203, 97, 3) == (68, 33, 204, 99)
    assert padded_layout(100, 50, 1) == (100, 50, 100, 50)
    assert padded_layout(8, 8, 8) == (1, 1, 8, 8)
    assert equal_row_bounds(10, 4).tolist() == [0, 3, 6, 9, 10]
    lay = BlockLayout([0, 5, 7, 12], chunks=2)          # blocks of 5, 2, 5 rows -> cb = 3, b = 6
    assert (lay.cb, lay.b, lay.n_pad) == (3, 6, 18)
    rows = lay.table_rows(np.arange(12))
    assert len(set(rows.tolist())) == 12 and rows.max() < lay.n_pad
    # chunk-major: chunk 0 of ranks 0, 1, 2 is the first slab, and a rank's chunk is contiguous inside it
    assert rows[:3].tolist() == [0, 1, 2] and rows[5:7].tolist() == [3, 4] and rows[7:10].tolist() == [6, 7, 8]
    assert rows[3:5].tolist() == [9, 10] and rows[10:12].tolist() == [15, 16]
    assert lay.chunk_slab(1) == slice(9, 18) and lay.my_slab(2, 1) == slice(15, 18)


def test_nnz_balanced_partition_evens_out_entries():
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    u, i = synth.interactions(3000, 1000, 60000, seed=4)
    g = NormGraph.from_pairs(u, i, 3000, 1000)
    ub, ib = g.partition(4)
    assert ub[0] == 0 and ub[-1] == 3000 and ib[0] == 3000 and ib[-1] == 4000
    per = np.diff(g.rowptr[ib])
    rows_equal = np.diff(g.rowptr[3000 + np.minimum(np.arange(5) * 250, 1000)])
    assert per.max() / per.mean() < 1.1 <= rows_equal.max() / rows_equal.mean() + 0.1
    assert per.max() <= rows_equal.max()


def test_forward_results_are_views_until_the_next_forward_and_copy_detaches_them(oracle):
    """ShardedPropagator.forward hands back views of its reusable buffers (documented lifetime: until the next
    forward on the object); copy=True gives private tensors that survive it."""
    import torch
    from dist_worker import oracle_spmm
    from textgcn_amd import synth
    from textgcn_amd.dist import ShardedPropagator
    from textgcn_amd.graph import NormGraph
    u, i = synth.interactions(203, 97, 2500, seed=1)
    g = NormGraph.from_pairs(u, i, 203, 97)
    sp = ShardedPropagator(g, 0, 1, 'cpu', local_spmm=oracle_spmm, split_threshold=None, chunks=2)
    e_a = synth.embeddings(g.n, 64, seed=2)
    e_b = synth.embeddings(g.n, 64, seed=3)
    for layers in (0, 3):
        ua, ia = sp.forward(*sp.local_e0(e_a), layers, copy=True)
        keep_u, keep_i = ua.clone(), ia.clone()
        va, vi = sp.forward(*sp.local_e0(e_a), layers)
        assert torch.equal(va, keep_u) and torch.equal(vi, keep_i)
        ub, ib = sp.forward(*sp.local_e0(e_b), layers)            # overwrites the buffers the views point into
        assert torch.equal(ua, keep_u) and torch.equal(ia, keep_i)          # the copies are untouched
        assert ub.data_ptr() == va.data_ptr() and torch.equal(va, ub)       # the view now shows the NEW forward (documented)
        assert not torch.equal(ub, keep_u)
        if layers:
            want, _ = oracle.propagate(*g.to_coo(), e_a.numpy(), layers)
            got_i = sp.items_in_order(keep_i).numpy()
            assert np.array_equal(bits(keep_u[:203].numpy()), bits(want[:203])) and np.array_equal(bits(got_i), bits(want[203:]))


@pytest.mark.parametrize('world,n_users,n_items,nnz,d,chunks,single', [
    (2, 203, 97, 2500, 64, 1, False), (3, 203, 97, 2500, 64, 3, False), (2, 203, 97, 2500, 64, 2, True),
    (3, 2, 5, 6, 16, 2, False),         # fewer users than ranks: a rank without users
    (4, 61, 7, 200, 32, 4, False),      # world 4, a chunk per item or two
    (2, 1, 1, 1, 100, 1, False),
])
def test_user_partition_matches_oracle_gloo(oracle, tmp_path, world, n_users, n_items, nnz, d, chunks, single):
    """The user partition (dist.UserShardedPropagator): users local, item rows as per-rank partial sums joined by ONE all-reduce of
    the item table per layer (gloo, asynchronous, in chunks under the user half-step).  An item row is a sum of per-rank chains, so
    the result equals the oracle to rounding (normwise 1e-6; the path's bar is 1e-4) -- and with ONE rank, or where every item's
    users live on one rank, bit for bit."""
    from conftest import normwise
    out = str(tmp_path / 'r0.npz')
    run_ranks(world, 'cpu', out, extra=('--shard', 'users', '--chunks', str(chunks), '--n-users', str(n_users), '--n-items', str(n_items),
                                        '--nnz', str(nnz), '--d', str(d), '--layers', '2') + (('--single',) if single else ()))
    got = np.load(out)
    ru, ri = reference(oracle, n_users=n_users, n_items=n_items, nnz=nnz, d=d, layers=2, single=single)
    assert got['users'].shape == ru.shape and got['items'].shape == ri.shape
    assert normwise(got['users'], ru) <= 1e-6 and normwise(got['items'], ri) <= 1e-6
    assert got['user_bounds'][0] == 0 and got['user_bounds'][-1] == n_users and len(got['user_bounds']) == world + 1


def test_user_partition_one_rank_is_bit_identical(oracle):
    """world = 1: the partial sums ARE the rows -- the oracle's bits, every chunk count"""
    from dist_worker import oracle_spmm
    from textgcn_amd import synth
    from textgcn_amd.dist import UserShardedPropagator
    from textgcn_amd.graph import NormGraph
    u, i = synth.interactions(203, 97, 2500, seed=1)
    g = NormGraph.from_pairs(u, i, 203, 97)
    e0 = synth.embeddings(g.n, 64, seed=2)
    want, _ = oracle.propagate(*g.to_coo(), e0.numpy(), 3)
    for chunks in (1, 4):
        up = UserShardedPropagator(g, 0, 1, 'cpu', local_spmm=oracle_spmm, split_threshold=None, chunks=chunks)
        assert up.nnz_local == g.nnz
        uu, ii = up.forward(*up.local_e0(e0), 3, exact=True)
        assert np.array_equal(bits(uu.numpy()), bits(want[:203])) and np.array_equal(bits(ii.numpy()), bits(want[203:]))
