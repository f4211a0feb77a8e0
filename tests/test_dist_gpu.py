"""Two ranks on the one-GPU box (both on cuda:0, gloo staged through the host) running the real HIP kernels:
the sharded result must be bit-identical to the single-GPU result."""
import numpy as np
import pytest

from conftest import bits
from test_dist_cpu import run_ranks

pytestmark = pytest.mark.gpu


def _single_gpu_reference(cuda, n_u, n_i, nnz, d=64, K=3):
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import Propagator
    u, i = synth.interactions(n_u, n_i, nnz, seed=1)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    e0 = synth.embeddings(g.n, d, seed=2).to(cuda)
    # rows > 64 entries are split in both runs; chunking depends only on the row -> identical bits
    return Propagator(g, cuda, split_threshold=64, segment=None).forward(e0, K).cpu().numpy()


@pytest.mark.parametrize('collective,chunks', [('torch', 1), ('torch', 4), ('capi', 1), ('capi', 3)])
def test_rccl_collective_path_runs_on_one_rank(cuda, tmp_path, collective, chunks):
    """The code the multi-GPU node runs -- backend nccl (= RCCL), in-place asynchronous all-gathers into the layer table,
    waits placed by the overlap schedule -- executed on the one-GPU box with a 1-rank communicator (the world == 1
    short-circuit is bypassed).  `capi`: libtgcn's tgcn_comm_init_rank / tgcn_allgather_rows on a side stream."""
    out = str(tmp_path / 'r0.npz')
    n_u, n_i, nnz = 2030, 970, 40000
    run_ranks(1, 'nccl', out, extra=('--n-users', str(n_u), '--n-items', str(n_i), '--nnz', str(nnz), '--collective', collective,
                                     '--chunks', str(chunks)))
    got = np.load(out)
    ref = _single_gpu_reference(cuda, n_u, n_i, nnz)
    assert np.array_equal(bits(got['users']), bits(ref[:n_u]))
    assert np.array_equal(bits(got['items']), bits(ref[n_u:]))


def _n_gpus():
    import torch
    return torch.cuda.device_count()      # does not initialise the GPU (the ranks are fresh child processes)


@pytest.mark.parametrize('collective,chunks', [('torch', 1), ('torch', 4), ('capi', 1), ('capi', 3)])
@pytest.mark.parametrize('world', [2, 4])
def test_rccl_multi_gpu_forward_is_bit_identical(tmp_path, world, collective, chunks):
    """The first thing to run on a multi-GPU box: `world` ranks, ONE GPU EACH, backend nccl (= RCCL over xGMI) -- the in-place
    asynchronous all_gather_into_tensor into the chunk-major layer table with real peers, the CapiComm id broadcast and
    tgcn_allgather_rows on its side stream -- must reproduce the one-GPU forward bit for bit.  Skipped where fewer GPUs are
    visible (this pool's one-GPU boxes); the ranks are fresh child processes started before anything here touches a GPU."""
    if _n_gpus() < world:
        pytest.skip(f'needs {world} GPUs, {_n_gpus()} visible')
    import torch
    out = str(tmp_path / 'r0.npz')
    n_u, n_i, nnz = 2030, 970, 40000
    run_ranks(world, 'nccl', out, extra=('--n-users', str(n_u), '--n-items', str(n_i), '--nnz', str(nnz), '--collective', collective,
                                         '--chunks', str(chunks), '--device-per-rank'))
    got = np.load(out)
    ref = _single_gpu_reference(torch.device('cuda:0'), n_u, n_i, nnz)
    assert np.array_equal(bits(got['users']), bits(ref[:n_u]))
    assert np.array_equal(bits(got['items']), bits(ref[n_u:]))


def test_rccl_multi_gpu_feature_partition(tmp_path):
    """the feature partition's one collective (all-gather + column interleave) over two real RCCL ranks"""
    if _n_gpus() < 2:
        pytest.skip(f'needs 2 GPUs, {_n_gpus()} visible')
    import torch
    out = str(tmp_path / 'r0.npz')
    n_u, n_i, nnz = 2030, 970, 40000
    run_ranks(2, 'nccl', out, extra=('--n-users', str(n_u), '--n-items', str(n_i), '--nnz', str(nnz), '--shard', 'features',
                                     '--device-per-rank'))
    got = np.load(out)
    ref = _single_gpu_reference(torch.device('cuda:0'), n_u, n_i, nnz)
    assert np.array_equal(bits(got['users']), bits(ref[:n_u])) and np.array_equal(bits(got['items']), bits(ref[n_u:]))


def _check_alone(r, K=3):
    lay = r['layers']
    assert len(lay['spmm_alone_ms']) == K and len(lay['allgather_alone_ms']) == K
    assert all(x > 0 for x in lay['spmm_alone_ms']) and all(x >= 0 for x in lay['allgather_alone_ms'])
    assert lay['overlap_efficiency'] is not None and lay['overlap_efficiency'] > 0


def test_bench_two_gpus_rccl(tmp_path):
    """`python bench.py --gpus 2` started PLAINLY (no torch.distributed.run: the form the driver's N = 1 command has -- bench.py
    starts the two ranks itself as child processes), one GPU per rank, RCCL, on the small workload: one JSON line whose config shows
    two ranks on two distinct devices, a per-layer compute / wait split and the two halves alone."""
    if _n_gpus() < 2:
        pytest.skip(f'needs 2 GPUs, {_n_gpus()} visible')
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('TGCN_BENCH_REHEARSAL', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--workload', 'small', '--score-batches', '1', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert r['n_gpus'] == 2 and r['scaling'] == 'strong' and r['config']['world_size'] == 2 and r['config']['distinct_devices'] == 2
    assert r['config']['backend'] == 'nccl' and len(r['layers']['max_over_ranks']) == 4
    _check_alone(r)
    assert r['user_partition']['vs_row_partition']['ok'] and r['user_partition']['value'] > 0


@pytest.mark.parametrize('mode,world,chunks', [('gpu', 2, 1), ('gpu', 3, 2), ('nccl', 1, 2)])
def test_user_partition_hip_forward(cuda, tmp_path, mode, world, chunks):
    """dist.UserShardedPropagator with the HIP kernels: two / three gloo ranks on the one GPU (item rows = sums of per-rank partial
    chains: the one-GPU result to rounding, normwise 1e-6) and the RCCL all-reduce itself on a 1-rank communicator (then every
    partial sum is the row: bit-identical)."""
    from conftest import normwise
    out = str(tmp_path / 'r0.npz')
    n_u, n_i, nnz = 2030, 970, 40000
    run_ranks(world, mode, out, extra=('--n-users', str(n_u), '--n-items', str(n_i), '--nnz', str(nnz), '--shard', 'users',
                                       '--chunks', str(chunks)))
    got = np.load(out)
    ref = _single_gpu_reference(cuda, n_u, n_i, nnz)
    if world == 1:
        assert np.array_equal(bits(got['users']), bits(ref[:n_u])) and np.array_equal(bits(got['items']), bits(ref[n_u:]))
    else:
        assert normwise(got['users'], ref[:n_u]) <= 1e-6 and normwise(got['items'], ref[n_u:]) <= 1e-6


@pytest.mark.parametrize('world,balance,chunks', [(2, 'nnz', 1), (3, 'nnz', 2), (3, 'rows', 1)])
def test_sharded_hip_forward_equals_single_gpu(cuda, tmp_path, world, balance, chunks):
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import Propagator
    out = str(tmp_path / 'r0.npz')
    n_u, n_i, nnz, d, K = 2030, 970, 40000, 64, 3
    run_ranks(world, 'gpu', out, extra=('--n-users', str(n_u), '--n-items', str(n_i), '--nnz', str(nnz), '--balance', balance,
                                        '--chunks', str(chunks)))
    got = np.load(out)
    u, i = synth.interactions(n_u, n_i, nnz, seed=1)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    e0 = synth.embeddings(g.n, d, seed=2).to(cuda)
    # rows > 64 entries are split in both runs; chunking depends only on the row -> identical bits
    ref = Propagator(g, cuda, split_threshold=64).forward(e0, K).cpu().numpy()
    assert np.array_equal(bits(got['users']), bits(ref[:n_u]))
    assert np.array_equal(bits(got['items']), bits(ref[n_u:]))


def test_sharded_chunks_use_the_segmented_kernels(cuda, tmp_path):
    """A shard whose gather table is a few L2 sizes (here: 750 item rows per rank of ~2000 entries over a 15 MB user table) runs the
    XCD-affine segmented kernels, by the rule a whole-graph Propagator uses.  Rows cut at column-block boundaries are summed
    piecewise: the 2-rank result equals the one-GPU one-chain result to rounding (normwise 1e-5; bar 1e-4), is identical run to
    run, and exact=True -- which ignores the plans -- stays bit-identical."""
    from conftest import normwise
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import Propagator
    n_u, n_i, nnz, d, K = 60000, 1500, 3_000_000, 64, 3
    args = ('--n-users', str(n_u), '--n-items', str(n_i), '--nnz', str(nnz), '--split-threshold', '8192')
    u, i = synth.interactions(n_u, n_i, nnz, seed=1)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    e0 = synth.embeddings(g.n, d, seed=2).to(cuda)
    ref = Propagator(g, cuda, split_threshold=None, segment=None).forward(e0, K, exact=True).cpu().numpy()
    outs = []
    for rep in range(2):
        out = str(tmp_path / f'r{rep}.npz')
        run_ranks(2, 'gpu', out, extra=args)
        got = np.load(out)
        assert 'row chunks on this rank' in str(got['segment_note']), got['segment_note']
        assert normwise(got['users'], ref[:n_u]) <= 1e-5 and normwise(got['items'], ref[n_u:]) <= 1e-5
        outs.append((got['users'], got['items']))
    assert np.array_equal(bits(outs[0][0]), bits(outs[1][0])) and np.array_equal(bits(outs[0][1]), bits(outs[1][1]))
    out = str(tmp_path / 'exact.npz')
    run_ranks(2, 'gpu', out, extra=args + ('--exact',))
    got = np.load(out)
    assert np.array_equal(bits(got['users']), bits(ref[:n_u])) and np.array_equal(bits(got['items']), bits(ref[n_u:]))


@pytest.mark.parametrize('launch', ['plain', 'torchrun'])
def test_bench_two_rank_rehearsal(cuda, tmp_path, launch):
    """bench.py's N > 1 code path (sharded propagation, per-rank scoring, max-over-ranks timing, one JSON line from
    rank 0) on the one-GPU box: two ranks on cuda:0 over gloo (TGCN_BENCH_REHEARSAL=1).  `plain`: `python bench.py --gpus 2`
    with no torch.distributed environment -- bench.py starts its ranks itself, as child processes; `torchrun`: as the driver
    launches N > 1.  The driver's real runs use RCCL with one GPU per rank; everything else is the same code."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    from test_dist_cpu import free_port
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(TGCN_BENCH_REHEARSAL='1', OMP_NUM_THREADS='4')
    tail = [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'small', '--score-batches', '1',
            '--no-cpu-baseline']
    if launch == 'plain':
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(free_port())] + tail
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['value'] > 0 and r['scaling'] == 'strong' and 'roofline' in r and r['scoring']['value'] > 0
    # two ranks, ONE device (the rehearsal): the line says so
    assert r['config']['world_size'] == 2 and r['config']['backend'] == 'gloo' and r['config']['distinct_devices'] == 1
    assert len(r['layers']['rank0']) == 4
    _check_alone(r)
    up = r['user_partition']
    assert up['value'] > 0 and up['vs_row_partition']['ok'] and len(up['layers_max_over_ranks']) == 3


def test_bench_sharded_path_on_one_rank_rccl(cuda):
    """bench.py --force-sharded: the N > 1 branch of the benchmark (process group on backend nccl, ShardedPropagator with the
    chunked in-place all-gathers, barrier + all-reduced timing, per-rank scoring) with a 1-rank RCCL communicator on the one
    GPU -- everything the driver's multi-GPU launch executes except a second rank."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-sharded', '--workload', 'small', '--steps', '2',
           '--warmup', '1', '--score-batches', '1', '--chunks', '3', '--no-cpu-baseline']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert r['n_gpus'] == 1 and 'scaling' not in r and 'row-sharded x1' in r['config']['sharding'] and '3 row chunk' in r['config']['sharding']
    assert r['value'] > 0 and r['scoring']['value'] > 0 and r['roofline']['frac'] > 0
    # the evidence fields of a multi-GPU line: what RCCL saw, on which devices, and where each layer's time went
    c = r['config']
    assert c['world_size'] == 1 and c['backend'] == 'nccl' and len(c['devices']) == 1 and c['distinct_devices'] == 1
    lay = r['layers']['rank0']
    assert [x['layer'] for x in lay] == [1, 2, 3, 'final gather'] and all(x['compute_ms'] > 0 for x in lay[:3])
    assert all(x['wait_on_gather_ms'] >= 0 for x in lay)
    _check_alone(r)


@pytest.mark.parametrize('d,world', [(64, 2), (64, 4), (64, 8), (128, 4), (256, 8)])
def test_column_partition_is_bit_identical(cuda, oracle, d, world):
    """The feature partition without any process group: every 'rank' computes its d / P columns of the K-layer forward on the
    whole graph (d / P in {8, 16, 32}: the narrow kernels); the concatenation must carry the bits of the full-width forward and
    of the oracle -- the product is independent per column."""
    import torch
    from textgcn_amd import synth
    from textgcn_amd.dist import ColumnShardedPropagator
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.propagate import Propagator
    u, i = synth.interactions(1500, 600, 30000, seed=6, zipf=1.0)
    g = NormGraph.from_pairs(u, i, 1500, 600)
    e0 = synth.embeddings(g.n, d, seed=3)
    K = 3
    for exact in (True, False):
        thr = None if exact else 48
        ref = Propagator(g, cuda, split_threshold=thr, segment=None).forward(e0.to(cuda), K, exact=exact)
        parts = []
        for r in range(world):
            cp = ColumnShardedPropagator(g, d, r, world, cuda, split_threshold=thr)
            parts.append(cp.forward(cp.local_e0(e0), K, exact=exact))
        got = torch.cat(parts, dim=1)
        assert torch.equal(got, ref), (exact, (got != ref).sum().item())
    idx, val = g.to_coo()
    want, _ = oracle.propagate(idx, val, e0.numpy(), K)
    cp = ColumnShardedPropagator(g, d, world - 1, world, cuda, split_threshold=None)
    out = cp.forward(cp.local_e0(e0), K, exact=True).cpu().numpy()
    assert np.array_equal(bits(out), bits(np.ascontiguousarray(want[:, cp.cols])))


@pytest.mark.parametrize('mode,world', [('gpu', 2), ('nccl', 1)])
def test_column_partition_assemble_collective(cuda, tmp_path, mode, world):
    """the one collective of the feature partition (all-gather + column interleave): two gloo ranks on the GPU, and the RCCL
    call itself on a 1-rank communicator"""
    out = str(tmp_path / 'r0.npz')
    n_u, n_i, nnz = 2030, 970, 40000
    run_ranks(world, mode, out, extra=('--n-users', str(n_u), '--n-items', str(n_i), '--nnz', str(nnz), '--shard', 'features'))
    got = np.load(out)
    ref = _single_gpu_reference(cuda, n_u, n_i, nnz)
    assert np.array_equal(bits(got['users']), bits(ref[:n_u]))
    assert np.array_equal(bits(got['items']), bits(ref[n_u:]))
