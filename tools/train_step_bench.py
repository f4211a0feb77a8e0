#!/usr/bin/env python3
"""Time one BPR training step (forward + backward through the HIP SpMM + Adam) on BASELINE config 2's graph."""
import json
import os
import sys
import types

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import synth  # noqa: E402
from textgcn_amd.graph import NormGraph, train_mask_csr  # noqa: E402
from textgcn_amd.model import LightGCN  # noqa: E402


def main():
    n_u, n_i, nnz, d, K = synth.CONFIGS['c2']
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    rp, items = train_mask_csr(u, i, n_u)
    ds = types.SimpleNamespace(n_users=n_u, n_items=n_i, graph=g, norm_matrix=None, mask_rowptr=rp, mask_items=items,
                               true_test_lil=[[0]], train_user_dict=None, test_df=pd.DataFrame({'user_id': [0], 'asin': [0]}),
                               user_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['u']}),
                               item_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['i']}))
    out = {}
    modes = sys.argv[1:] or ['device', 'device-generic', 'cpu']
    for rng_mode in modes:
        p = types.SimpleNamespace(k=[20, 40], emb_size=d, n_layers=K, device='cuda:0', load=None, batch_size=2048, quiet=True,
                                  dropout=0.4, dropout_rng=rng_mode.split('-')[0], lr=1e-3)
        m = LightGCN(p, ds)
        if rng_mode.endswith('generic'):      # the torch composition subclasses with their own scoring keep
            m._native_loss = lambda: False
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)     # as LightGCN.fit does on the GPU
        m.training = True
        from collections import defaultdict
        m._loss_values = defaultdict(float)
        rng = np.random.default_rng(0)
        batch = torch.from_numpy(np.stack([rng.integers(0, n_u, 2048), rng.integers(0, n_i, 2048), rng.integers(0, n_i, 2048)], axis=1))

        def step():
            opt.zero_grad()
            loss = m.get_loss(batch)
            loss.backward()
            opt.step()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        out[f'train_step_ms_dropout_rng_{rng_mode}'] = (time.perf_counter() - t0) / n * 1e3
        del m, opt
        torch.cuda.empty_cache()
    print(json.dumps({'config': 'c2 training step (batch 2048, K=3 forward + backward, Adam, dropout 0.4)', **out}))


if __name__ == '__main__':
    main()
