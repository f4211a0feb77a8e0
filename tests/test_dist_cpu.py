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


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('single', [False, True])
def test_sharded_forward_matches_oracle_gloo(oracle, tmp_path, world, single):
    out = str(tmp_path / 'r0.npz')
    run_ranks(world, 'cpu', out, extra=('--single',) if single else ())
    got = np.load(out)
    ru, ri = reference(oracle, single=single)
    assert np.array_equal(bits(got['users']), bits(ru))
    assert np.array_equal(bits(got['items']), bits(ri))


def test_padded_layout_and_local_blocks():
    from textgcn_amd.dist import padded_layout
    assert padded_layout(203, 97, 3) == (68, 33, 204, 99)
    assert padded_layout(100, 50, 1) == (100, 50, 100, 50)
    assert padded_layout(8, 8, 8) == (1, 1, 8, 8)
