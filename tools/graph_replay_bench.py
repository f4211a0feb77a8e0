#!/usr/bin/env python3
"""Eager launches vs HIP-graph replay of the K-layer forward on small graphs (launch-bound regime)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import synth  # noqa: E402
from textgcn_amd.graph import NormGraph  # noqa: E402
from textgcn_amd.propagate import Propagator  # noqa: E402

dev = torch.device('cuda:0')
res = {}
for name in ('tiny', 'small', 'c2'):
    n_u, n_i, nnz, d, K = synth.CONFIGS[name]
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    prop = Propagator(g, dev)
    e0 = synth.embeddings(g.n, d).to(dev)
    out = torch.empty_like(e0)
    for fn, label in ((lambda: prop.forward(e0, K, out=out), 'eager'), (lambda: prop.forward_graphed(e0, K, out=out), 'graph')):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[f'{name}_{label}_us'] = (time.perf_counter() - t0) / n * 1e6
print(json.dumps(res))
